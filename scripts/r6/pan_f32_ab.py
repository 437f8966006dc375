#!/usr/bin/env python3
"""PAN in the fp32 mode (float32 tensors / -no_fp16): the SCPA trunk as one split-operand launch per block (csrc/pan_scpa_split.hip, fused_scpa = 1) against the
six generic fp32 launches per block (fused_scpa = 0): ms per forward at 540 x 960 and 16 x 200^2, max |difference| of the two, per-launch breakdown of the new form,
and both against the oracle on ragged sizes (border tiles, batches)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
import oracle
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()

def ms(x, reps=10):
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        net(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for shape in [(2, 3, 37, 45), (1, 3, 50, 70), (3, 3, 8, 32), (1, 3, 9, 33), (1, 3, 64, 96)]:
    x = torch.from_numpy(synth.uniform(shape, 11, 0, 1))
    with torch.no_grad():
        ref = oracle.pan_forward(sd, x, nb=16, scale=4)
    out = {}
    for f in (1, 0):
        net.fused_scpa = f
        out[f] = net(x.to(dev)).cpu()
    print(f"{shape}: split-fused vs oracle {(out[1] - ref).abs().max().item():.2e}   generic vs oracle {(out[0] - ref).abs().max().item():.2e}   |ref| max {ref.abs().max().item():.2f}", flush=True)

for shape in [(1, 3, 540, 960), (16, 3, 200, 200)]:
    x = torch.from_numpy(synth.uniform(shape, 3, 0, 1)).to(dev)
    t, y = {}, {}
    for f in (1, 0, 1, 0):
        net.fused_scpa = f
        t.setdefault(f, []).append(ms(x))
        y[f] = net(x)
    print(f"== {shape} fp32 mode: split-fused {t[1][0]:.3f} / {t[1][1]:.3f} ms   generic {t[0][0]:.3f} / {t[0][1]:.3f} ms   max |diff| {(y[1] - y[0]).abs().max().item():.2e}", flush=True)
    net.fused_scpa = 5
    print(f"   (the PA blocks as their own split 1x1 launches: {ms(x):.3f} ms; max |diff| to the fused form {(net(x) - y[1]).abs().max().item():.2e})", flush=True)
    net.fused_scpa = 1
    launches = L.timed_launches(lambda: net(x))
    agg = {}
    for name, msl, fl, by in launches:
        a = agg.setdefault(name, [0.0, 0])
        a[0] += msl; a[1] += 1
    for k, (m, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:70s} n={n:4d} {m:9.4f} ms")
