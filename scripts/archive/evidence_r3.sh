#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root: scripts/evidence_r3.sh
# Everything lands under gpurun_out/r3/evidence/; the summaries are copied into profiles/r3/evidence/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/evidence
mkdir -p $OUT
cd $ROOT
# 1. the driver's command: headline (BASELINE config 2) with roofline + power probe, tile_sharded (config 4), chop8k (config 3), unet64 (config 5, per-kernel), cpu baseline
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_frame1080.json 2> $OUT/bench_frame1080_per_layer.txt
# 2. rocprofv3 kernel stats + FETCH / WRITE passes of the same workload
bash scripts/profile.sh r3_frame --no-extras > $OUT/profile_frame.txt 2>&1
cp gpurun_out/prof_r3_frame/traffic.json $OUT/traffic.json 2>/dev/null
# 3. the fp32-accurate engine on the same frame (-no_fp16)
python3 bench.py --fp32 --steps 5 --warmup 2 --no-extras --sharded-steps 0 --no-cpu-baseline > $OUT/bench_frame1080_fp32.json 2> $OUT/bench_frame1080_fp32.err
# 4. config 5 kernel statistics at two forward counts: what scales with the forwards and what is set-up (the copyBuffer calls = the one-off weight upload)
for REPS in 5 25; do
( cd /tmp && export TMPDIR=/tmp && UNET_N=64 UNET_REPS=$REPS rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r3_unet_$REPS -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/prof_unet_$REPS.err )
python3 - <<PY > $OUT/kernel_stats_unet64_reps$REPS.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r3_unet_$REPS/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- UNET_N=64 UNET_REPS=$REPS python3 scripts/bench_unet.py (2 warm-up + $REPS timed forwards of 64x3x256x256)")
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", "")[:84].ljust(84), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(14), r["AverageNs"][:12].rjust(13), r["Percentage"])
PY
done
# 5. SQ / TCC counters of the shipped trunk kernels (separate --pmc passes)
bash scripts/pmc_conv.sh r3_c160k32 160 32 1080 > $OUT/pmc_pc_160to32.txt 2>&1
bash scripts/pmc_conv.sh r3_c192k64 192 64 1080 > $OUT/pmc_pc_192to64.txt 2>&1
ls -la $OUT
