#!/bin/bash
# average duration of first_conv_mfma inside SRResNet / RRDBNet 1080p forwards (rocprofv3 kernel stats) for two library builds
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
for L in ${BASE:-build/libinnfer_amd_base2.so} innfer_amd/lib/libinnfer_amd.so; do
  for A in srgan esrgan; do
    export INNFER_LIB=$ROOT/$L
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_fc -- python3 $ROOT/scripts/r5/fp32_once.py $A fp16 > /dev/null 2>&1 )
    f=$(find gpurun_out/prof_fc -name '*kernel_stats.csv' | head -1)
    echo "$L $A: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'first_conv' in r['Name']: print(r['Name'][-40:], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
")"
    rm -rf gpurun_out/prof_fc
  done
done
