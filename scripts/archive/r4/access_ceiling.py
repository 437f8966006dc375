#!/usr/bin/env python3
"""The ceiling of conv3x3_pc's own access pattern: the kernel's loader -- its LDS-DMA piece shapes, 26 x 34-pixel halo tiles,
weight panels, persistent tile walk -- and its slab stores, with the MFMA phase and the consumers' LDS reads removed (diagnostic library `make ablate`):

  abl  0   the shipped schedule (MFMA + data movement)
  abl  8   data movement only, loaders still COUPLED to the consumers: issue chunk g + 1, wait for all of it, one barrier per chunk
  abl 40   data movement only, loaders FREE-RUNNING: wait only for the chunk before the one just issued, one barrier per tile
  abl  9 / 41   the same two without the stores (the read stream alone)
  abl 12 / 44   the same two without the input pieces (weights + stores)

Prints us per launch and TB/s of algorithmic bytes ((C + K) * 2 B per pixel) at 1080 x 1920; run it under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
for the counter bytes (scripts/r4/access_ceiling.sh).  Results are wrong by construction; only times and bytes mean anything."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ.setdefault("INNFER_LIB", os.path.join(REPO, "innfer_amd", "lib", "libinnfer_amd_ablate.so"))
from scripts.bench_conv import run  # noqa: E402

layers = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]
abls = [int(v) for v in os.environ.get("ABLS", "0,8,40,9,41,12,44").split(",")]
reps = int(os.environ.get("REPS", "20"))
for (Cc, K) in layers:
    for abl in abls:
        os.environ["INNFER_ABL"] = str(abl)
        print(f"abl={abl:2d} ", end="")
        run(Cc, K, 1080, 1920, reps=reps)
