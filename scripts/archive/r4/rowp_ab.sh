#!/bin/bash
# A/B of the plane row order of the 64-output layers (shipped) against the lane-contiguous order (make ab_rowp0), interleaved on one box
for i in 1 2 3; do
  for lib in "" innfer_amd/lib/libinnfer_amd_rowp0.so; do
    INNFER_LIB=$lib python3 bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline --sharded-steps 0 --no-power-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
pk=d.get('per_kernel') or d['roofline'].get('per_kernel')
print('%-28s frame %.3f ms  %.4f of the MFMA peak | ' % ('${lib:+lane-contiguous rows (A/B)}' or 'plane rows (shipped)', d['ms_per_step'], d['frac_of_mfma_peak']) + '  '.join('%s %.3f' % (k.replace('conv3x3_pc',''), v['ms_total']) for k, v in pk.items()))
"
  done
done
