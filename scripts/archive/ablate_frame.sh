#!/bin/bash
# Sustained (power-settled) ablation of the producer / consumer kernels: the WHOLE bench frame on the diagnostic library with parts of the
# kernels removed (INNFER_ABL bits: 1 no stores, 2 no weight DMA, 4 no input DMA, 8 no MFMA phase).  Results are wrong by construction; only
# the times mean anything.  A single-conv loop (scripts/ablate.py) finishes before the power manager has settled and reads 20 % fast.
cd ${GRAFT_REPO_ROOT:-.}
for abl in ${ABLS:-0 1 8 9 6 7 14}; do
  INNFER_ABL=$abl INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline --sharded-steps 0 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('abl=$abl', d['ms_per_step'], {k.replace('conv3x3_pc',''):round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()}, d['roofline'].get('power'))"
done
