#!/usr/bin/env python3
"""Fuzz of PAN's fp32 mode on split operands (csrc/pan_scpa_split.hip, conv3x3_pc SPLIT + self gate, pan_attention_mfma<true>): random variants (scale 1..4, 1..5 SCPA blocks,
self attention on / off, double trunk, bilinear stages, 1..4 input channels), ragged frames around the 8 x 32 tile and the 4 x 4 pooling, batches -- against the CPU oracle
(<= 1e-4 of the output range) and against the generic fp32 kernels of the same module (innfer_pan_set_fused_scpa(pan, 0); <= 1e-5).
Usage: fuzz_pan_f32.py [seconds] [seed]; prints BAD lines, exit code 1 if any."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle
from innfer_amd import synth
from innfer_amd.architectures.PAN_arch import PAN
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
dev = torch.device("cuda:0")
torch.set_num_threads(max(1, os.cpu_count() or 1))
bad = done = 0
t0 = time.time()
while time.time() - t0 < budget:
    scale = int(rng.choice([1, 2, 3, 4, 4, 4]))
    nb = int(rng.randint(1, 6))
    sa, dbl = bool(rng.randint(0, 2)), bool(rng.randint(0, 4) == 0)
    mode = "bilinear" if rng.randint(0, 5) == 0 else "nearest"
    in_nc = int(rng.choice([1, 3, 3, 4]))
    n = int(rng.randint(1, 4))
    h = int(rng.choice([4, 5, 7, 8, 9, 15, 16, 17, 24, 31, 33, int(rng.randint(4, 90))]))
    w = int(rng.choice([4, 5, 31, 32, 33, 63, 64, 65, 96, int(rng.randint(4, 130))]))
    net = PAN(in_nc, in_nc, 40, 24, nb, scale=scale, self_attention=sa, double_scpa=dbl, ups_inter_mode=mode)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict({k: tuple(t.shape) for k, t in net.state_dict().items()}, int(rng.randint(1, 1 << 20))).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.uniform((n, in_nc, h, w), int(rng.randint(1, 1 << 20)), 0, 1))
    with torch.no_grad():
        ref = oracle.pan_forward(sd, x, nb=nb, scale=scale, ups_inter_mode=mode, self_attention=sa, double_scpa=dbl)
    net.fused_scpa = 1
    a = net(x.to(dev)).cpu()
    net.fused_scpa = 0
    b = net(x.to(dev)).cpu()
    lim = max(1.0, ref.abs().max().item())
    e, d = (a - ref).abs().max().item(), (a - b).abs().max().item()
    ok = e < 1e-4 * lim and d < 1e-5 * lim and bool(torch.isfinite(a).all())
    done += 1
    if not ok:
        bad += 1
        print(f"BAD scale {scale} nb {nb} sa {sa} dbl {dbl} {mode} in_nc {in_nc} shape {(n, in_nc, h, w)}: vs oracle {e:.2e} vs generic {d:.2e} (range {lim:.2f})", flush=True)
    del net
print(f"fuzz_pan_f32: {done} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
