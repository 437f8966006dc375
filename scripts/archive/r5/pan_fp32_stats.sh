#!/bin/bash
# rocprofv3 per-kernel statistics of PAN 540 x 960 in fp32 mode (6 forwards)
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_panf32 -- python3 $ROOT/scripts/r5/fp32_once.py pan > /dev/null 2> $OUT/panf32.err )
f=$(find $OUT/prof_panf32 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:16]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/6e3:9.1f} us/fwd  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
rm -rf $OUT/prof_panf32
