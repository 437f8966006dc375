#!/usr/bin/env python3
"""Run ONE conv config a few times (target program of scripts/pmc_conv.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_conv import run
Cc, K, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
run(Cc, K, H, 1920, reps=int(sys.argv[4]) if len(sys.argv) > 4 else 5)
