#!/usr/bin/env python3
"""SRResNet / SRGAN 4x on a 1080p frame and a 16-tile chop batch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("srgan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
for shape in ((1, 3, 1080, 1920), (16, 3, 200, 200)):
    x = torch.from_numpy(synth.uniform(shape, 3)).to(dev).half()
    for _ in range(2): y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): y = net(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = net.flops(*shape[:1], *shape[2:])
    print(f"SRResNet 4x {shape}: {ms:8.3f} ms  {16 * shape[0] * shape[2] * shape[3] / ms / 1e3:8.1f} out MPix/s  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
