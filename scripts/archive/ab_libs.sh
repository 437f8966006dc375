#!/bin/bash
# A/B of shared-library builds on ONE box: scripts/ab_libs.sh lib1.so lib2.so ...  (two bench runs each, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for L in "$@"; do
    INNFER_LIB=$PWD/innfer_amd/lib/$L python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-extras --sharded-steps 0 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['value'], {k.replace('conv3x3_pc',''):round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
  done
done
