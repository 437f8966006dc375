#!/bin/bash
# Upper bound of what an LDS-resident-input gather GEMM could gain on the UNet's <= 8x8 levels: the diagnostic library stages the pixel operand for the FIRST tap of a
# chunk only (INNFER_GG_ABL=1: 1/16 resp. 1/4 of the gathers; results wrong by construction), per-kernel medians of the x64 forward
ROOT=$(pwd); OUT=$ROOT/gpurun_out
export INNFER_LIB=$ROOT/innfer_amd/lib/libinnfer_amd_ablate.so UNET_N=64 UNET_REPS=60
for abl in 0 3; do
  ( cd /tmp && export TMPDIR=/tmp && INNFER_GG_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_gg$abl -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/gg$abl.err )
  echo "== INNFER_GG_ABL=$abl"; python3 scripts/r4/kernel_medians.py gpurun_out/prof_gg$abl unet_first_mfma 20 | grep -E "forwards in the trace|gemm_gather"
  rm -rf $OUT/prof_gg$abl
done
