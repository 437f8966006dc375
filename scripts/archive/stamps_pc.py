#!/usr/bin/env python3
"""Diagnostic: where the consumer / loader waves of conv3x3_pc spend their cycles (needs `make stamps`;
run with INNFER_LIB=innfer_amd/lib/libinnfer_amd_stamps.so INNFER_PC=1).  Sums over all chunks of a launch of
wave 0 (consumer) and wave 8 (loader) of every workgroup."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import innfer_amd.lib as L
from scripts.bench_conv import run
L.lib.innfer_debug_read_stamps.restype = C.c_int
L.lib.innfer_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
L.lib.innfer_debug_clear_stamps.restype = C.c_int
names = ["consumer compute", "consumer epilogue", "consumer barrier wait", "loader issue", "loader vmcnt wait", "loader barrier wait"]
for (Cc, K) in [(64, 32), (160, 32), (192, 64)]:
    L.lib.innfer_debug_clear_stamps()
    run(Cc, K, 1080, 1920, reps=1)          # 3 warm-up launches + 1 timed = 4 launches accumulated
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert L.lib.innfer_debug_read_stamps(buf.ctypes.data, buf.size) == 16
    st = buf.reshape(8192, 16)[:256].astype(np.float64)
    chunks = st[:, 6].mean()
    print(f"--- C={Cc} K={K}: {chunks:.0f} chunks per workgroup (4 launches); cycles per chunk, mean over workgroups:")
    print(f"  shader clock over the chunk loop: {st[:, 7].sum() / (st[:, 8].sum() / 100.0) / 1e3:.3f} GHz; loop wall time per launch {st[:, 8].mean() / 100.0 / 4:.1f} us")
    for i, nm in enumerate(names):
        print(f"  {nm:24s} {st[:, i].mean() / chunks:9.0f}")
