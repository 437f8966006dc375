#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root: scripts/evidence_r2.sh
# Everything lands under gpurun_out/r2/evidence/; the summaries are copied into profiles/r2/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/evidence
mkdir -p $OUT
cd $ROOT
# 1. headline bench (BASELINE config 2) with roofline, power probe, tile_sharded (config 4) and cpu baseline
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_frame1080.json 2> $OUT/bench_frame1080.err
# 2. rocprofv3 kernel stats + FETCH / WRITE passes of the same workload
bash scripts/profile.sh r2_frame > $OUT/profile_frame.txt 2>&1
cp gpurun_out/prof_r2_frame/traffic.json $OUT/traffic.json 2>/dev/null
# 3. configs 3 / 4 on one GPU: bench lines + kernel stats
for w in chop8k chain4k; do
  python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --sharded-steps 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r2_$w -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --sharded-steps 0 > /dev/null 2> $OUT/prof_$w.err )
  python3 - <<PY > $OUT/kernel_stats_$w.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r2_$w/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 1 --warmup 1 (two passes of the workload)")
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", "")[:84].ljust(84), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(14), r["AverageNs"][:12].rjust(13), r["Percentage"])
PY
done
# 4. config 5: pix2pix UNet_256, batch 64
UNET_N=64 python3 scripts/bench_unet.py > $OUT/bench_unet64.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && UNET_N=64 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r2_unet -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $OUT/prof_unet.err )
python3 - <<PY > $OUT/kernel_stats_unet64.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r2_unet/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- UNET_N=64 python3 scripts/bench_unet.py (7 forwards of 64x3x256x256)")
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", "")[:84].ljust(84), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(14), r["AverageNs"][:12].rjust(13), r["Percentage"])
PY
# 5. SQ / TCC counters of the SHIPPED producer / consumer kernels on the two trunk shapes (separate --pmc passes)
bash scripts/pmc_conv.sh r2_c160k32 160 32 1080 > $OUT/pmc_pc_160to32.txt 2>&1
bash scripts/pmc_conv.sh r2_c192k64 192 64 1080 > $OUT/pmc_pc_192to64.txt 2>&1
ls -la $OUT
