#!/bin/bash
# What bounds the two phase up-conv launches of the frame: the same bench with the diagnostic library's ablations (results wrong by construction, only times count)
# INNFER_ABL 0 shipped work / 4 no input DMA / 1 no stores / 5 neither / 8 no MFMA phase / 16 stores into a cache-resident window / 64 phase stores as whole lines
export INNFER_LIB=innfer_amd/lib/libinnfer_amd_ablate.so
for form in "" "--upconv-phase-visits"; do
echo "== ${form:-four phases in one visit of a tile (default)}"
for abl in 0 512 0 512; do
  INNFER_ABL=$abl python3 bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline --sharded-steps 0 --no-power-probe $form 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
pk=d.get('per_kernel') or d['roofline'].get('per_kernel')
r=pk['conv3x3_pc<2,4,4,0>+upconv_phases']
print('abl=%-3s up-conv launches: %.4f ms total (%d launches)   frame %.2f ms' % ('$abl', r['ms_total'], r['launches'], d['ms_per_step']))
"
done
done
