#!/usr/bin/env python3
"""Every launch of one UNet_256 x64 fp16 forward in order (the library's HIP-event launch timer): name, us, algorithmic GFLOP / MB."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("p2p_256", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).train()
x = torch.from_numpy(synth.uniform((64, 3, 256, 256), 3, -1, 1)).to(dev).half()
for _ in range(5): net(x)
torch.cuda.synchronize()
runs = [L.timed_launches(lambda: net(x)) for _ in range(5)]
n = len(runs[0])
tot = 0.0
for i in range(n):
    ms = sorted(r[i][1] for r in runs)[2]
    name, _, fl, by = runs[0][i]
    tot += ms
    print(f"{i:3d} {name[:70]:70s} {ms * 1e3:8.1f} us  {fl / 1e9:8.2f} GFLOP  {by / 1e6:8.1f} MB  {fl / ms / 1e9 if ms else 0:7.1f} TFLOP/s")
print(f"sum {tot * 1e3:.1f} us")
