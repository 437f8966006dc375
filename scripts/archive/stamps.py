#!/usr/bin/env python3
"""Diagnostic: WG timeline of the v1 conv kernel from in-kernel s_memtime stamps
(needs `make stamps`; run with INNFER_LIB=innfer_amd/lib/libinnfer_amd_stamps.so INNFER_PERSIST=0: one workgroup per tile)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("INNFER_PC", "0")      # these hooks live in the two-workgroup kernel (conv3x3_mfma)
import innfer_amd.lib as L
from scripts.bench_conv import run  # noqa

L.lib.innfer_debug_read_stamps.restype = C.c_int
L.lib.innfer_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
names = ["start", "prolog_done", "c0_issued", "c0_landed", "c0_done", "c1_issued", "c1_landed", "c1_done",
         "c2_issued", "c2_landed", "c2_done", "loop_done", "stores_done", "stores_issued"]
for (Cc, K) in [(64, 32), (160, 32), (192, 64)][:int(os.environ.get('STAMPS_CASES', '3'))]:
    run(Cc, K, 1080, 1920, reps=1)
    torch.cuda.synchronize()
    nwg = 512          # persistent: two workgroups per CU; stamps are those of each workgroup's LAST tile
    nwg = min(nwg, 8192)
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    ns = L.lib.innfer_debug_read_stamps(buf.ctypes.data, buf.size)
    assert ns == 16, ns
    st = buf.reshape(8192, 16)[:nwg].astype(np.float64)
    t0 = st[:, 0].min()
    print(f"--- C={Cc} K={K}: kernel span {(st[:, 12].max() - t0) / 100:.1f} us (100 MHz s_memtime ticks?)")
    valid = [0, 1, 2, 3, 4, 5, 6, 7] + ([8, 9, 10] if Cc >= 96 else []) + [11, 13, 12]
    rel = st - st[:, :1]
    med = np.median(rel, axis=0)
    for i in valid:
        print(f"  {names[i]:12s} median {med[i]:9.0f} ticks   p10 {np.percentile(rel[:, i], 10):9.0f}  p90 {np.percentile(rel[:, i], 90):9.0f}")
    wall = (st[:, 15] - st[:, 14]) / 100.0          # us (s_memrealtime, 100 MHz)
    print(f"  shader clock while this tile ran: median {np.median(rel[:, 12] / wall) / 1e3:.3f} GHz  (tile wall time {np.median(wall):.2f} us)")
    life = rel[:, 12]
    print(f"  WG lifetime median {np.median(life):.0f} ticks; start times: first {0:.0f}, median {np.median(st[:, 0] - t0):.0f}, last {(st[:, 0] - t0).max():.0f}")
