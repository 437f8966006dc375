#!/bin/bash
# SQ counters of the fp32-mode conv kernel per instantiation: scripts/r5/pmc_f32.sh <tag> [arch]   (separate --pmc passes, kernel trace only beside them)
set -u
TAG=$1; ARCH=${2:-p2p_256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/r5/fp32_once.py $ARCH > $OUT/p$i.out 2> $OUT/p$i.err
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "f32conv" not in r["Kernel_Name"]: continue
        key = r["Kernel_Name"].split("(")[0][-28:] + " grid " + r.get("Grid_Size", "?") + " lds " + r.get("LDS_Block_Size", "?")
        a = agg[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(agg):
    print(key)
    for k in sorted(agg[key]): print(f"    {k:32s} avg/dispatch {agg[key][k][0] / agg[key][k][1]:16.1f}  (n={agg[key][k][1]})")
PY
