#!/usr/bin/env python3
"""PAN fp16: the SCPA block on two 4-wave workgroups per CU (8 x 32 tiles, pan_scpa_duo; fused_scpa = 6) against one 8-wave workgroup (16 x 32, fused_scpa = 7) and the default (by the frame):
bit equality on ragged / batched / border-only frames, then ms per forward and per-launch times at 540 x 960 and 16 x 200^2, interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
for shape in [(1, 3, 8, 32), (1, 3, 9, 33), (2, 3, 37, 45), (1, 3, 6, 40), (3, 3, 16, 64), (1, 3, 70, 130), (1, 3, 200, 200), (1, 3, 270, 480)]:
    x = torch.from_numpy(synth.uniform(shape, 60 + shape[2], 0, 1)).to(dev).half()
    net.fused_scpa = 7; a = net(x)
    net.fused_scpa = 6; b = net(x)
    print(f"{shape}: duo == mono: {torch.equal(a, b)}   max |diff| {(a.float() - b.float()).abs().max().item():.2e}", flush=True)

def ms(x, reps=20):
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): net(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for shape in [(1, 3, 540, 960), (16, 3, 200, 200), (1, 3, 200, 200)]:
    x = torch.from_numpy(synth.uniform(shape, 3, 0, 1)).to(dev).half()
    t = {7: [], 6: [], 1: []}
    for f in (7, 6, 1, 7, 6, 1):
        net.fused_scpa = f
        t[f].append(round(ms(x), 4))
    per = {}
    for f in (7, 6, 1):
        net.fused_scpa = f
        per[f] = round(sum(m for n, m, fl, by in L.timed_launches(lambda: net(x)) if "pan_scpa" in n), 4)
    print(f"== {shape}: mono {t[7]} ms (16 launches {per[7]} ms)   duo {t[6]} ms (16 launches {per[6]} ms)   default {t[1]} ms ({per[1]} ms)", flush=True)
net.fused_scpa = 1
