#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root: scripts/evidence_r4.sh.  Everything lands under gpurun_out/r4/evidence/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/evidence
mkdir -p $OUT
cd $ROOT
# 1. the driver's command
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_frame1080.json 2> $OUT/bench_frame1080_per_layer.txt
# 2. rocprofv3 kernel stats + FETCH / WRITE passes of the headline workload
bash scripts/profile.sh r4_frame --no-extras > $OUT/profile_frame.txt 2>&1
cp gpurun_out/prof_r4_frame/traffic.json $OUT/traffic.json 2>/dev/null
# 3. config 5 and PAN kernel statistics as MEDIANS over forwards, warm-up dropped (scripts/r4/kernel_medians.py)
( cd /tmp && export TMPDIR=/tmp && UNET_N=64 UNET_REPS=60 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_r4_unet -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/prof_unet.err )
python3 scripts/r4/kernel_medians.py gpurun_out/prof_r4_unet unet_first_mfma 20 > $OUT/kernel_medians_unet64.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_r4_pan -- python3 $ROOT/scripts/r4/pan_trace.py > /dev/null 2> $OUT/prof_pan.err )
python3 scripts/r4/kernel_medians.py gpurun_out/prof_r4_pan pan_pre 10 > $OUT/kernel_medians_pan540.txt 2>&1
# 4. the fp32 modes
python3 bench.py --fp32 --steps 5 --warmup 2 --no-extras --sharded-steps 0 --no-cpu-baseline > $OUT/bench_frame1080_fp32.json 2> $OUT/bench_frame1080_fp32.err
python3 scripts/r4/fp32_modes.py > $OUT/fp32_modes.txt 2>&1
ls -la $OUT
