#!/bin/bash
# rocprofv3 kernel statistics of the CycleGAN ResNet-9 and PPON benches (whole process, warm-up included: for the distribution over kernels only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out
for g in cyclegan ppon; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$g -- python3 $ROOT/scripts/bench_$g.py > $OUT/bench_$g.txt 2> $OUT/$g.err )
  echo "== $g"; grep -v amdgpu $OUT/bench_$g.txt | tail -3
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/prof_$g/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-110s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -rf $OUT/prof_$g
done
