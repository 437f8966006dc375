#!/bin/bash
# UNet_256 x64 with phases of the halo-tile conv kernel removed one at a time (diagnostic build: `make ablate`), kernel statistics per setting.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for A in ${ABLS:-0 1 2 4 8 16}; do
  UNET_N=64 INNFER_ABL=$A INNFER_LIB=$ROOT/innfer_amd/lib/libinnfer_amd_ablate.so rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abl_unet_$A -- python3 $ROOT/scripts/bench_unet.py > /dev/null 2> $ROOT/gpurun_out/abl_unet_$A.err
  echo "ABL=$A"; python3 - $ROOT/gpurun_out/abl_unet_$A <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"))[-1]
import re
for r in sorted(csv.DictReader(open(f)), key=lambda r: r["Name"]):
    if "conv3x3_pc" in r["Name"]:
        m = re.search(r"conv3x3_pc<([^>]*)>", r["Name"])
        # (the first launch of an instantiation carries the code-object load: the minimum is the steady state)
        print("   conv3x3_pc<%s>" % m.group(1).replace(" ", ""), r["Calls"].rjust(4), "avg", r["AverageNs"][:9].rjust(10), "min", r["MinNs"].rjust(8))
PY
done
