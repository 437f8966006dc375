#!/bin/bash
# per-kernel medians of the UNet x64 forward with one stored skip form and with two (diagnostic library)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
export INNFER_LIB=$ROOT/innfer_amd/lib/libinnfer_amd_ablate.so UNET_N=64 UNET_REPS=60
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ov1 -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/ov1.err )
( cd /tmp && export TMPDIR=/tmp && INNFER_UNET_TWO_VIEWS=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ov2 -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/ov2.err )
echo "== one view"; python3 scripts/r4/kernel_medians.py gpurun_out/prof_ov1 unet_first_mfma 20
echo "== two views"; python3 scripts/r4/kernel_medians.py gpurun_out/prof_ov2 unet_first_mfma 20
