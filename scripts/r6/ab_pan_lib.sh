#!/bin/bash
# same-box A/B of the fp16 PAN engine: previous library vs current, interleaved
for i in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_prev.so; else unset INNFER_LIB; fi
    echo "== $lib (round $i)"
    python3 scripts/bench_pan.py 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
