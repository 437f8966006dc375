#!/usr/bin/env python3
"""Ablation of the conv kernel (diagnostic library `make ablate`, INNFER_LIB must point at it):
INNFER_ABL bits 1 no stores, 2 no weight DMA, 4 no input DMA, 8 no MFMA phase, 16 stores land in a
cache-resident window.  Results are wrong
by construction; only the launch times mean anything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# INNFER_PC=0: the two-workgroup kernel (conv3x3_mfma); default: the producer / consumer kernels (bits 1, 2, 4, 8)
os.environ.setdefault("INNFER_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                 "innfer_amd", "lib", "libinnfer_amd_ablate.so"))
from scripts.bench_conv import run
for (Cc, K) in [(64, 32), (160, 32), (192, 64)]:
    for abl in [int(v) for v in os.environ.get('ABLS', '0,1,8,9,6,7,14').split(',')]:
        os.environ["INNFER_ABL"] = str(abl)
        print(f"abl={abl:2d} ", end="")
        run(Cc, K, 1080, 1920, reps=20)
