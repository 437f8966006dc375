#!/usr/bin/env python3
"""Per-launch timeline of the LAST forward in a rocprofv3 --kernel-trace csv: timeline.py <dir> <name of the forward's first kernel>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    n = r["Kernel_Name"].replace("innfer::", "").replace("(anonymous namespace)::", "")[:64]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {n:64s} {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
