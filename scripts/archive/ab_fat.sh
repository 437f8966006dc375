#!/bin/bash
# A/B of the fat-consumer-wave instantiations on ONE box (diagnostic library reads INNFER_FAT: 1 = 32-output layers, 2 = 64-output layers)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for F in 0 1 2 3; do
    INNFER_FAT=$F INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --sharded-steps 0 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FAT=$F', d['ms_per_step'], d['value'], {k.replace('conv3x3_pc',''):round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
  done
done
