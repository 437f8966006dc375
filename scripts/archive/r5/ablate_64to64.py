#!/usr/bin/env python3
"""Ablation of the 64 -> 64 conv (SRResNet's trunk layers, 2 chunks per tile: VERDICT r4 weak 6) on the diagnostic library (`make ablate`):
INNFER_ABL bits 1 no stores, 2 no weight DMA, 4 no input DMA, 8 no MFMA phase, 32 free-running loaders.  Wrong results by construction; only times mean anything."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("INNFER_LIB", os.path.join(ROOT, "innfer_amd", "lib", "libinnfer_amd_ablate.so"))
from scripts.bench_conv import run
for (Cc, K, res) in [(64, 64, False), (64, 64, True), (192, 64, True)]:
    for abl in [0, 1, 2, 4, 6, 7, 8, 9, 14, 15, 32]:
        os.environ["INNFER_ABL"] = str(abl)
        print(f"res={int(res)} abl={abl:2d} ", end="")
        run(Cc, K, 1080, 1920, reps=30, res=res)
