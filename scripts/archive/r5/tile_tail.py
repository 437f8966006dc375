#!/usr/bin/env python3
"""what does the tile-round tail of the 32-output layers cost?
One persistent workgroup per CU walks ceil(tiles / 256) tiles; at 1080 x 1920 a 24 x 32 tiling has 2700 tiles = 10.55 rounds, so the
launch lasts 11 tile times and a CU idles 4 % of it on average -- if tiles cost a fixed time.  Measure us/Mpx of each trunk layer at
frame heights whose tile counts end a round differently (interleaved, three passes, median)."""
import os
import sys
import statistics
import io
import contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
import numpy as np
import innfer_amd.lib as L

dev = torch.device("cuda:0")
W = 1920


def make(Cc, K, H, res):
    g = H * W * 32
    slab = (torch.rand((Cc // 32) * g, device=dev) - 0.5).half()
    out = torch.empty((max(K, 32) // 32) * g, dtype=torch.float16, device=dev)
    w = ((np.random.rand(K, Cc, 3, 3) - 0.5) / np.sqrt(9 * Cc)).astype(np.float32)
    packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, K, Cc, packed.ctypes.data))
    d_packed = torch.from_numpy(packed).to(dev)
    d_bias = torch.zeros(64, device=dev)
    a = L.ConvArgs()
    a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g, Cc
    a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
    a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g, 0, K
    a.N, a.H, a.W, a.act = 1, H, W, 1
    if res:
        a.d_res1, a.res1_group_stride, a.res1_scale = slab.data_ptr(), g, 0.2
    return a, (slab, out, d_packed, d_bias)


def timed(a, reps=40):
    for _ in range(3):
        L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lib.innfer_conv3x3_f16(C.byref(a), None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if __name__ == "__main__":
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    print(f"# CUs {ncu}; us/Mpx per layer and frame height (median of 3 interleaved passes of 40 launches)")
    for (Cc, K, th) in [(64, 32, 24), (96, 32, 24), (128, 32, 24), (160, 32, 24), (192, 64, 16)]:
        hs = (1008, 1032, 1080, 1104, 1224) if th == 24 else (1024, 1040, 1080, 1088, 1092)
        sets = {H: make(Cc, K, H, K == 64) for H in hs}
        res = {H: [] for H in hs}
        for _ in range(3):
            for H in hs:
                res[H].append(timed(sets[H][0]))
        for H in hs:
            tiles = ((H + th - 1) // th) * (W // 32)
            us = statistics.median(res[H])
            r = tiles / ncu
            print(f"C={Cc:3d} K={K:2d} H={H:4d} tiles={tiles:5d} rounds={r:6.2f} ceil/rounds={-(-tiles // ncu) / r:5.3f}  {us:7.1f} us  us/Mpx={us / (H * W / 1e6):6.2f}  us/tile-round={us / -(-tiles // ncu):6.2f}", flush=True)
        del sets
        torch.cuda.empty_cache()
