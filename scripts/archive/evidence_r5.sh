#!/bin/bash
# Round-5 evidence, run on the GPU box from the repo root: scripts/evidence_r5.sh.  Everything lands under gpurun_out/r5/evidence/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5/evidence
mkdir -p $OUT
cd $ROOT
# 1. the driver's command
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_frame1080.json 2> $OUT/bench_frame1080_per_layer.txt
# 2. cumulative same-box A/B: the round-4 library (build/libinnfer_amd_r4.so, kept from the round's first minute) against this round's, interleaved, driver-form frame
#    (VERDICT r4 weak 8).  The round-4 Python shells do not exist any more: both libraries run under this round's host code (the RRDB path's host code is unchanged).
if [ -f build/libinnfer_amd_r4.so ]; then
  cp build/libinnfer_amd_r4.so innfer_amd/lib/libinnfer_amd_r4.so
  for rep in 1 2 3; do
    for LIB in libinnfer_amd_r4.so libinnfer_amd.so; do
      INNFER_LIB=$PWD/innfer_amd/lib/$LIB INNFER_ABI_ANY=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-power-probe --no-extras --sharded-steps 0 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LIB', d['ms_per_step'], d['value'], {k.replace('conv3x3_pc',''):round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
    done
  done > $OUT/ab_r4_vs_r5.txt 2>&1
  rm -f innfer_amd/lib/libinnfer_amd_r4.so
fi
# 3. rocprofv3 kernel stats + FETCH / WRITE passes of the headline workload
bash scripts/profile.sh r5_frame --no-extras > $OUT/profile_frame.txt 2>&1
cp gpurun_out/prof_r5_frame/traffic.json $OUT/traffic.json 2>/dev/null
# 4. kernel statistics of the side workloads (rocprofv3 --kernel-trace --stats of one program each)
for ARCH in srgan pan p2p_256; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r5_$ARCH -- python3 $ROOT/scripts/r5/fp32_once.py $ARCH fp16 > /dev/null 2> $OUT/prof_$ARCH.err )
  python3 - <<PY > $OUT/kernel_stats_$ARCH.txt 2>&1
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r5_$ARCH/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- python3 scripts/r5/fp32_once.py $ARCH fp16   (6 forwards, the first includes weight packing)")
for r in list(csv.DictReader(open(f)))[:18]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "")[:100].ljust(100), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:10].rjust(12), r["Percentage"])
PY
done
# 5. the fp32 modes
python3 bench.py --fp32 --steps 5 --warmup 2 --no-extras --sharded-steps 0 --no-cpu-baseline > $OUT/bench_frame1080_fp32.json 2> $OUT/bench_frame1080_fp32.err
python3 scripts/r4/fp32_modes.py > $OUT/fp32_modes.txt 2>&1
ls -la $OUT
