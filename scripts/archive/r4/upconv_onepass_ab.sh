#!/bin/bash
# A/B of the frame with the phase up-convs as one visit per tile (default) and as one phase per visit (round 3's form), interleaved, shipped library
for i in 1 2 3; do
  for f in "" "--upconv-phase-visits"; do
    python3 bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline --sharded-steps 0 --no-power-probe $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
pk=d.get('per_kernel') or d['roofline'].get('per_kernel')
r=pk['conv3x3_pc<2,4,4,0>+upconv_phases']
print('%-24s frame %.3f ms  frac %.4f   up-conv launches %.4f ms' % ('$f' or 'one visit (default)', d['ms_per_step'], d['roofline']['frac'] if 'frac' in d['roofline'] else 0, r['ms_total']))
"
  done
done
