#!/bin/bash
# A/B of bench.py flag sets on ONE box (two interleaved runs each): scripts/r3/ab_flags.sh "" "--no-upconv-phases" "--no-fused-tail --no-upconv-phases"
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for F in "$@"; do
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-extras --sharded-steps 0 $F 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$F]', d['ms_per_step'], d['value'], {k.replace('conv3x3_pc',''):round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
  done
done
