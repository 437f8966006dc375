#!/bin/bash
# A/B of shared-library builds on ONE box, chop workload: scripts/ab_chop.sh lib1.so lib2.so ...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for L in "$@"; do
    INNFER_LIB=$PWD/innfer_amd/lib/$L python bench.py --workload ${WL:-chop4k} --steps 2 --warmup 1 --no-cpu-baseline --sharded-steps 0 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['value'])"
  done
done
