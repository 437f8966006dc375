#!/usr/bin/env python3
"""Per-kernel table of one PAN 4x forward (library launch timer): 1x540x960 and 16x200x200, SCPA block as one launch / five launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
import innfer_amd.lib as L
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
for (N, H, W) in ((1, 540, 960), (16, 200, 200)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3)).to(dev).half()
    for fused in (True, False):
        net.fused_scpa = fused
        for _ in range(5):
            net(x)
        launches = L.timed_launches(lambda: net(x))
        agg = {}
        for i, (name, ms, fl, by) in enumerate(launches):
            a = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
            a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
        print(f"--- PAN 4x N={N} {H}x{W}, {'one launch per SCPA block' if fused else 'five launches per block'}: {len(launches)} timed launches, sum {sum(l[1] for l in launches):.3f} ms")
        for name, (ms, fl, by, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print(f"  {name[:64]:64s} n={n:3d} {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s  {by / ms / 1e6 if ms else 0:8.1f} GB/s (algorithmic)")
        if fused:
            print("  in order:", ", ".join(f"{l[0][:18]}:{l[1] * 1e3:.0f}us" for l in launches))
net.fused_scpa = True
