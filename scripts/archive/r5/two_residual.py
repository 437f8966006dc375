#!/usr/bin/env python3
"""what does the RRDB-end launch's second (memory) residual cost in isolation?  The dense block's last conv (192 -> 64, plane row order, x from the conv's own
LDS stages = the shipped RLDS form) at 1080 x 1920 with and without the RRDB-level residual, interleaved; the difference against the residual's bytes."""
import os, sys, statistics
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import innfer_amd.lib as L
dev = torch.device("cuda:0")
H, W, Cc, K = 1080, 1920, 192, 64
g = H * W * 32
slab = (torch.rand((Cc // 32) * g, device=dev) - 0.5).half()
res2 = (torch.rand(2 * g, device=dev) - 0.5).half()
out = torch.empty(2 * g, dtype=torch.float16, device=dev)
w = ((np.random.rand(K, Cc, 3, 3) - 0.5) / np.sqrt(9 * Cc)).astype(np.float32)
packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
L.check(L.lib.innfer_pack_conv3x3_rows(w.ctypes.data, K, Cc, 1, packed.ctypes.data))
d_packed = torch.from_numpy(packed).to(dev)
d_bias = torch.zeros(64, device=dev)


def args(two, lds):
    a = L.ConvArgs()
    a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g, Cc
    a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
    a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g, 0, K
    a.N, a.H, a.W, a.act, a.plane_rows = 1, H, W, 0, 1
    a.d_res1, a.res1_group_stride, a.res1_scale = slab.data_ptr(), g, 0.2
    a.res1_from_input = int(lds)
    if two:
        a.d_res2, a.res2_group_stride, a.res2_scale = res2.data_ptr(), g, 0.2
    return a


def timed(a, reps=40):
    for _ in range(3):
        L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lib.innfer_conv3x3_f16(C.byref(a), None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cases = {"x from memory, one residual": args(False, False), "x from LDS, one residual": args(False, True),
         "x from memory, two residuals": args(True, False), "x from LDS + RRDB residual from memory (shipped RRDB-end form)": args(True, True)}
res = {k: [] for k in cases}
for _ in range(5):
    for k, a in cases.items():
        res[k].append(timed(a))
med = {k: statistics.median(v) for k, v in res.items()}
for k, v in med.items():
    print(f"192 -> 64 @1080x1920, {k:70s} {v:7.1f} us")
rb = H * W * 64 * 2 / 1e6
d = med["x from LDS + RRDB residual from memory (shipped RRDB-end form)"] - med["x from LDS, one residual"]
print(f"the RRDB-level residual: {rb:.0f} MB read per launch, +{d:.1f} us = {rb / d / 1e3 * 1e3:.2f} TB/s -- it moves at the chip's streaming rate and nothing hides it: the epilogue's loads are")
print("issued when the tile's MFMAs are done (no register room to issue them earlier: 168-register budget; no free LDS stage: 2 x 78.8 of 160 KB), see DESIGN section 5")
