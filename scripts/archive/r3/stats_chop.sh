#!/bin/bash
# rocprofv3 kernel statistics of BASELINE configs 3 / 4 on one GPU (bench.py --workload chop8k | chain4k), round 3 -> gpurun_out/r3/evidence/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/evidence
mkdir -p $OUT
cd $ROOT
for w in chop8k chain4k; do
  python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --sharded-steps 0 --no-extras > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r3_$w -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --sharded-steps 0 --no-extras > /dev/null 2> $OUT/prof_$w.err )
  python3 - <<PY > $OUT/kernel_stats_$w.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r3_$w/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 1 --warmup 1 (two passes of the workload)")
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", "")[:84].ljust(84), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(14), r["AverageNs"][:12].rjust(13), r["Percentage"])
PY
done
tail -n 3 $OUT/kernel_stats_*.txt
