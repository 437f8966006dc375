#!/usr/bin/env python3
"""PAN 4x (nf 40, unf 24, 16 SCPA blocks, FSA) on a 200x200 chop tile batch and on a 540x960 frame."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
def timed(x, reps=20, windows=5):
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    win = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(x)
        e1.record(); torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / reps)
    return sorted(win)[len(win) // 2]


for (N, H, W) in [(16, 200, 200)] if os.environ.get("PAN_ONLY_TILES") else ((1, 200, 200), (16, 200, 200), (1, 540, 960)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3)).to(dev).half()
    for fused in (True, False, True, False):          # interleaved A/B: an SCPA block as one launch (default) / as five launches
        net.fused_scpa = fused
        ms = timed(x)
        print(f"PAN 4x N={N:2d} {H}x{W} {'one launch per SCPA block' if fused else 'five launches per block  '}: {ms:9.3f} ms  {N * H * W * 16 / ms / 1e3:8.2f} output MPix/s  "
              f"{2 * 488952 * N * H * W / ms / 1e9:7.2f} TFLOP/s", flush=True)
    net.fused_scpa = True
