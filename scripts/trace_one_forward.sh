#!/bin/bash
# Per-launch timeline of the LAST forward of a small script: scripts/trace_one_forward.sh <tag> <script.py> <launches per forward, 0 = guess>
TAG=$1; SCRIPT=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_$TAG -- python3 $ROOT/$SCRIPT 2>&1 | grep -E "UNet|PAN|PPON|ms"
cd $ROOT
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/trace_$TAG/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last forward = from the last occurrence of the first kernel of a forward
first = next(n for n in names if "unet_pre" in n or "pre" in n.lower())
i0 = max(i for i, n in enumerate(names) if n == first)
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
tot = {}
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("innfer::", "").replace("(anonymous namespace)::", "")[:48]
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?")); w = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {n:48s} grid {g} wg {w}")
    tot[n] = tot.get(n, 0) + (e - s); prev_end = e
print("total", (prev_end - t0) / 1e3, "us")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"  {v / 1e3:9.1f} us  {n}")
PY
