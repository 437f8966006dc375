#!/bin/bash
# Phases of f32conv_tiled removed one at a time (diagnostic library `make ablate`, INNFER_F32_ABL bits: 1 MFMA steps, 2 epilogue, 4 patch loads, 8 weight DMA):
# per conv class of PAN 540x960 / UNet x64 in the fp32 mode.  Results are wrong by construction; only times mean anything.
cd ${GRAFT_REPO_ROOT:-.}
export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so
for abl in 0 1 2 4 8 3 7 15; do
  echo "=== INNFER_F32_ABL=$abl"
  INNFER_F32_ABL=$abl DETAIL=1 python scripts/r5/fp32_breakdown.py ${1:-pan} 2>&1 | grep -v amdgpu.ids | head -8
done
