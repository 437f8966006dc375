#!/bin/bash
# rocprofv3 kernel statistics of the smaller generators at their bench sizes (three forwards each; the first includes the weight upload copies)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/other_generators_kernel_stats.txt
mkdir -p $ROOT/gpurun_out/r2; : > $OUT
for s in "unet_up_vs_deconv.py deconv" cyclegan_once.py wbc_once.py ppon_once.py pan_once.py; do
  set -- $s
  tag=${1%.py}
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/og_$tag -- python3 $ROOT/scripts/$1 $2 > /dev/null 2>&1 )
  python3 - "$ROOT/gpurun_out/og_$tag" "$s" >> $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats -- python3 scripts/" + sys.argv[2])
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", "")[:96].ljust(96), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:10].rjust(11), r["Percentage"][:5])
print()
PY
done
cat $OUT
