#!/usr/bin/env python3
"""PPON 4x (nf 64, 24 + 4 residual-in-residual blocks) on 200x200 chop tiles and a 540x960 frame."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("ppon", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
# MACs per input pixel: first conv + 84 residual blocks x (c1 + 8 dilated + c2) + LR conv + 3 heads
mac = 3 * 64 * 9 + 84 * (64 * 64 * 9 + 8 * 64 * 32 * 9 + 256 * 64) + 64 * 64 * 9 + 3 * (4 * 64 * 64 * 9 + 16 * 64 * 64 * 9 + 16 * 64 * 64 * 9 + 16 * 64 * 3 * 9)
for (N, H, W) in ((1, 200, 200), (8, 200, 200), (1, 540, 960)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3)).to(dev).half()
    for _ in range(2): y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): y = net(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"PPON 4x N={N:2d} {H}x{W}: {ms:9.3f} ms  {N * H * W * 16 / ms / 1e3:8.2f} output MPix/s  {2 * mac * N * H * W / ms / 1e9:7.2f} TFLOP/s", flush=True)
