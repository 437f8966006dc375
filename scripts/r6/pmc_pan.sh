#!/bin/bash
# SQ counters of PAN's kernels at 540 x 960 (pan_scpa_fused, pan_attention_mfma), one --pmc pass per counter group, --kernel-trace only beside it.
set -u -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pan
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/scripts/r6/pan540_once.py" > "$OUT/p$i.txt" 2> "$OUT/p$i.err" || echo "pass $i failed" >> "$OUT/failed"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("pan_scpa_fused", "pan_attention_mfma"):
            if key in r["Kernel_Name"]:
                a = agg[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key, d in agg.items():
    print(f"{key}, PAN 4x 540 x 960, per dispatch:")
    g = lambda k: d[k][0] / max(1, d[k][1])
    for k in sorted(d): print(f"  {k:28s} {g(k):18.0f}   ({d[k][1]} dispatches)")
    if g("SQ_INSTS_MFMA"): print(f"  other VALU per MFMA {(g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'):.2f}; LDS per MFMA {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):.2f}; SALU per MFMA {g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA'):.2f}")
    if g("GRBM_GUI_ACTIVE"): print(f"  matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (g('GRBM_GUI_ACTIVE') / 8):.3f} of the launch; launch = {g('GRBM_GUI_ACTIVE') / 8:.0f} cycles")
    if g("SQ_WAVE_CYCLES"): print(f"  waves: parked {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.2f}, issue-stalled {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f}, issuing {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f}")
    if g("SQ_LDS_IDX_ACTIVE"): print(f"  LDS bank conflicts {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f} of the LDS-array cycles")
PY
