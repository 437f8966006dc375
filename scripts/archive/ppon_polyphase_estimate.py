#!/usr/bin/env python3
"""Estimate for DESIGN.md section 7: PPON's eight dilated 3x3 convs (64 -> 32, rates 1..8) as ordinary 3x3 convs on the polyphase
components of the tiles (rate r: N*r*r sub-images of ceil(200/r)^2), timed with the halo-tile kernel on contiguous sub-images --
a LOWER bound for the strided-addressing version (it ignores the half-used 128-byte lines).  Compare with the grouped gather GEMM."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import innfer_amd.lib as L
dev = torch.device("cuda:0")
N, S = 8, 200
tot = 0.0
for r in range(1, 9):
    n, h = N * r * r, (S + r - 1) // r
    g = n * h * h * 32
    slab = (torch.rand(2 * g, device=dev) - 0.5).half()
    out = torch.empty(g, dtype=torch.float16, device=dev)
    w = ((np.random.rand(32, 64, 3, 3).astype(np.float32) - 0.5) / 24)
    packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(32, 64), dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, 32, 64, packed.ctypes.data))
    dp = torch.from_numpy(packed).to(dev); db = torch.zeros(64, device=dev)
    a = L.ConvArgs()
    a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g, 64
    a.d_packed, a.d_bias = dp.data_ptr(), db.data_ptr()
    a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g, 0, 32
    a.N, a.H, a.W, a.act = n, h, h, 0
    for _ in range(3): L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.lib.innfer_conv3x3_f16(C.byref(a), None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"rate {r}: {n:4d} sub-images of {h:3d}^2  {us:7.1f} us", flush=True)
print(f"eight rates: {tot:.1f} us per residual block at {N} tiles of {S}^2")
