#!/usr/bin/env python3
"""Single-conv microbenchmark through the C ABI: repeated launches of one 3x3 conv on a
blocked-NHWC slab.  Small H keeps input+output resident in the 256 MiB Infinity Cache,
large H streams from HBM.  Prints us/launch, TFLOP/s and algorithmic GB/s."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import innfer_amd.lib as L

dev = torch.device("cuda:0")


def run(Cc, K, H, W, reps=20, res=False):
    N = 1
    pad = int(os.environ.get("INNFER_GPAD", "0"))          # extra elements between channel groups
    g = N * H * W * 32 + pad
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
    a.N, a.H, a.W, a.act = N, H, W, 1
    if res:
        a.d_res1, a.res1_group_stride, a.res1_scale = slab.data_ptr(), g, 0.2
    for _ in range(3):
        L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lib.innfer_conv3x3_f16(C.byref(a), None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * 9 * Cc * K * H * W
    by = (Cc + K) * 2.0 * H * W
    print(f"C={Cc:3d} K={K:2d} H={H:4d} W={W:4d} in+out={by / 1e6:7.1f} MB  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  "
          f"{by / us / 1e3:6.2f} TB/s(alg)  us/Mpx={us / (H * W / 1e6):6.1f}", flush=True)


if __name__ == "__main__":
    W = 1920
    for (Cc, K) in [(64, 32), (128, 32), (160, 32), (192, 64), (64, 64)]:
        for H in (128, 272, 544, 1088, 2176):
            run(Cc, K, H, W)
