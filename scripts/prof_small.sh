#!/bin/bash
# rocprofv3 --kernel-trace --stats of a small script: scripts/prof_small.sh <tag> <script.py>
TAG=$1; SCRIPT=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/$SCRIPT 2>&1 | grep -E "UNet|PAN"
cd $ROOT
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_$TAG/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:95].ljust(95), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:10].rjust(12), r["Percentage"])
PY
