#!/bin/bash
# What bounds each halo-tile launch of the UNet x64 forward: per-kernel medians under the conv kernel's ablations (diagnostic library; results wrong by construction)
# INNFER_ABL 0 shipped / 1 no stores / 2 no weight DMA / 4 no input DMA / 8 no MFMA phase
ROOT=$(pwd); OUT=$ROOT/gpurun_out
export INNFER_LIB=$ROOT/innfer_amd/lib/libinnfer_amd_ablate.so UNET_N=64 UNET_REPS=60
for abl in 0 1 2 4 8; do
  ( cd /tmp && export TMPDIR=/tmp && INNFER_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ua$abl -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/ua$abl.err )
  echo "== INNFER_ABL=$abl"; python3 scripts/r4/kernel_medians.py gpurun_out/prof_ua$abl unet_first_mfma 20 | grep -E "forwards in the trace|conv3x3_pc" | cut -c1-140
  rm -rf $OUT/prof_ua$abl
done
