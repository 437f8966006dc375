#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace CSV WITHOUT the warm-up: dispatches are grouped into forwards by a marker kernel (the first kernel of a
forward), the first `skip` forwards are dropped, and per kernel name the report gives calls per forward, MEDIAN / mean / max duration and the share of the
median forward (the round-3 summaries averaged a one-off 25 ms dispatch into a 12 us kernel).
usage: kernel_medians.py <dir with *_kernel_trace.csv> <marker substring> [skip forwards]"""
import csv, glob, os, statistics, sys
from collections import defaultdict

root, marker = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 10
files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("innfer::(anonymous namespace)::", "").replace("innfer::gg::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:92]
fwd, cur = [], None
for r in rows:
    name = r["Kernel_Name"]
    if marker in name:
        cur = []
        fwd.append(cur)
    if cur is not None:
        cur.append((short(name), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
if len(fwd) <= skip + 1:
    sys.exit(f"only {len(fwd)} forwards found for marker '{marker}'")
kept = fwd[skip:-1] if len(fwd[-1]) != len(fwd[skip]) else fwd[skip:]
per = defaultdict(list)
tot = []
for f in kept:
    tot.append(sum(d for _, d in f))
    for n, d in f:
        per[n].append(d)
med_total = statistics.median(tot)
print(f"# {len(fwd)} forwards in the trace, the first {skip} dropped as warm-up, {len(kept)} kept; kernel time per forward: median {med_total:.1f} us, "
      f"min {min(tot):.1f}, max {max(tot):.1f}; longest single dispatch among the kept forwards {max(max(v) for v in per.values()):.1f} us")
print(f"{'kernel':92s} {'calls/fwd':>9s} {'median us':>10s} {'mean us':>9s} {'max us':>9s} {'% of fwd':>8s}")
for n, v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    cpf = len(v) / len(kept)
    print(f"{n:92s} {cpf:9.1f} {statistics.median(v):10.2f} {statistics.mean(v):9.2f} {max(v):9.2f} {100 * statistics.median(v) * cpf / med_total:8.1f}")
dropped = [d for f in fwd[:skip] for _, d in f]
print(f"# warm-up forwards (dropped): longest dispatch {max(dropped):.1f} us")
