#!/bin/bash
# SQ counters of hr_chain_kernel alone (scripts/micro/hr_chain_micro, shipped form: INNFER_ABL unset), one --pmc pass per counter group, --kernel-trace only beside it.
set -u -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_chain
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- "$ROOT/scripts/micro/hr_chain_micro" 4320 7680 6 > "$OUT/p$i.txt" 2> "$OUT/p$i.err" || echo "pass $i failed" >> "$OUT/failed"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hr_chain_kernel" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("hr_chain_kernel<1>, 4320 x 7680, per dispatch (mean over the dispatches of each pass):")
for k in sorted(agg):
    v, n = agg[k]
    print(f"  {k:28s} {v / n:18.0f}   ({n} dispatches)")
g = lambda k: agg[k][0] / max(1, agg[k][1])
if g("SQ_LDS_IDX_ACTIVE"): print(f"  LDS bank-conflict share of LDS-array cycles: {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")
if g("SQ_BUSY_CYCLES"): print(f"  MFMA busy / SQ busy cycles (per-SE sums): {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):.3f}")
if g("SQ_INSTS_MFMA"): print(f"  VALU (incl. MFMA) per MFMA instruction: {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.2f};  LDS instructions per MFMA: {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):.2f}")
PY
