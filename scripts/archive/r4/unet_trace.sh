#!/bin/bash
# per-kernel medians of the UNet x64 forward (shipped library)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
export UNET_N=64 UNET_REPS=60
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_unet -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/unet_trace.err )
python3 scripts/r4/kernel_medians.py gpurun_out/prof_unet unet_first_mfma 20
rm -rf $OUT/prof_unet
