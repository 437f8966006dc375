#!/bin/bash
# SQ counters per kernel of ANY python program (one --pmc pass per counter group beside --kernel-trace):  scripts/r6/pmc_generic.sh <tag> <script.py> [env assignments are the caller's]
set -u -o pipefail
TAG=$1; PROG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/$PROG" > "$OUT/p$i.txt" 2> "$OUT/p$i.err" || echo "pass $i failed" >> "$OUT/failed"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("innfer::(anonymous namespace)::", "").replace("void ", ""))[:60]
        a = agg[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for key, d in agg.items():
    g = lambda k, d=d: d[k][0] / max(1, d[k][1])
    if not g("GRBM_GUI_ACTIVE") or "rocclr" in key or "at::" in key: continue
    tot = d["GRBM_GUI_ACTIVE"][0]
    rows.append((tot, key, d["GRBM_GUI_ACTIVE"][1], g))
for tot, key, n, g in sorted(rows, reverse=True)[:14]:
    mf = g("SQ_INSTS_MFMA")
    print(f"{key:60s} n={n:4d} cycles/launch={g('GRBM_GUI_ACTIVE') / 8:10.0f}  mfma_busy={g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / max(1.0, g('GRBM_GUI_ACTIVE') / 8):5.3f} "
          f"valu/mfma={(g('SQ_INSTS_VALU') - mf) / mf if mf else float('nan'):6.2f} lds/mfma={g('SQ_INSTS_LDS') / mf if mf else float('nan'):5.2f} "
          f"parked={g('SQ_WAIT_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):4.2f} stalled={g('SQ_WAIT_INST_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):4.2f} issuing={g('SQ_ACTIVE_INST_ANY') / max(1.0, g('SQ_WAVE_CYCLES')):4.2f} "
          f"lds_conflict={g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):5.3f}")
PY
