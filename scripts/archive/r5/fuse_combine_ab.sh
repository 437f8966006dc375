#!/bin/bash
# average duration of fuse_combine_kernel inside SRResNet / RRDBNet 1080p forwards (rocprofv3 kernel stats) for two library builds + the forwards' output checksums
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
for L in ${BASE:-build/libinnfer_amd_base7.so} innfer_amd/lib/libinnfer_amd.so; do
  for A in srgan esrgan; do
    export INNFER_LIB=$ROOT/$L
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_fc -- python3 $ROOT/scripts/r5/fp32_once.py $A fp16 > /dev/null 2>&1 )
    f=$(find gpurun_out/prof_fc -name '*kernel_stats.csv' | head -1)
    echo "$L $A: $(python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'fuse_combine' in r['Name']: print('fuse_combine calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
")"
    rm -rf gpurun_out/prof_fc
  done
done
