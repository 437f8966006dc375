#!/bin/bash
# Hardware counters of the row-Winograd experiment against the direct conv on the same tiles and the shipped kernel (160 -> 32 at 1080p): one rocprofv3 --pmc pass per
# counter set and mode.  scripts/r3/wino_pmc.sh   -> gpurun_out/wino_pmc.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wino_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in 0 2 1; do
  i=0
  for SET in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/m${MODE}_p$i -- python3 $ROOT/scripts/r3/wino_one.py 160 32 $MODE > $OUT/m${MODE}_p$i.out 2> $OUT/m${MODE}_p$i.err
  done
done
cd $ROOT
python3 - <<PY > $ROOT/gpurun_out/wino_pmc.txt
import csv, glob, collections
names = {0: "shipped conv3x3_pc<3,2,4> (24 x 32 tiles)", 2: "direct conv on 16 x 32 tiles", 1: "Winograd F(2,3) rows on 16 x 32 tiles"}
for mode in (0, 2, 1):
    agg = collections.defaultdict(lambda: [0.0, 0]); dur = []
    for f in glob.glob("$OUT/m%d_p*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3" not in r["Kernel_Name"]: continue
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("$OUT/m%d_p1/**/*kernel_trace.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3x3" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("160 -> 32, 1x1080x1920,", names[mode], ": dispatches", len(dur), " avg duration us %.1f" % (sum(dur) / max(1, len(dur)) / 1e3))
    for k in sorted(agg): print("  %-28s avg/dispatch %16.1f" % (k, agg[k][0] / agg[k][1]))
PY
cat $ROOT/gpurun_out/wino_pmc.txt
