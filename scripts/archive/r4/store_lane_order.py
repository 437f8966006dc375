#!/usr/bin/env python3
"""Lane order of the slab stores (diagnostic library, INNFER_ABL 256: lane L writes piece L of its wave's 1 KB run -- same bytes, same lines, values permuted):
interleaved rounds of (shipped order, lane order) per layer shape, median over the rounds."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("INNFER_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "innfer_amd", "lib", "libinnfer_amd_ablate.so"))
import io, contextlib, re
from scripts.bench_conv import run
for (Cc, K) in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]:
    t = {0: [], 256: [], 1: []}
    for r in range(9):
        for abl in (0, 256, 1):
            os.environ["INNFER_ABL"] = str(abl)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                run(Cc, K, 1080, 1920, reps=30)
            t[abl].append(float(re.search(r"([0-9.]+) us ", buf.getvalue()).group(1)))
    m = {a: statistics.median(v) for a, v in t.items()}
    print(f"C={Cc:3d} K={K:2d}: shipped {m[0]:7.1f} us   lane-ordered stores {m[256]:7.1f} us ({100 * (m[256] / m[0] - 1):+.1f} %)   no stores {m[1]:7.1f} us", flush=True)
