#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of scripts/profile.sh: per-kernel time stats and
per-kernel HBM bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of the
bytes of wide coalesced reads; WRITE_SIZE as is; both counters are in KiB... see below)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


def short(name):
    name = name.replace("innfer::(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


print("== kernel stats (--kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"  {short(r['Name']):70s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} "
              f"avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")

# kernel-trace: per-dispatch durations (to match the counter passes by dispatch order)
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = find(f"{tag}/**/*counter_collection.csv")
    if not files:
        print(f"== {counter}: no counter_collection.csv ==")
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"== {counter} per kernel (raw counter units as reported by rocprofv3) ==")
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f"  {k:70s} dispatches={n:6d} sum={v:16.0f} avg={v / n:14.1f}")

# machine-readable per-kernel HBM traffic for bench.py's roofline.traffic:
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB  (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half of
# the bytes of wide coalesced reads, WRITE_SIZE is exact; both counters are in KiB)
import json, re
traffic = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{tag}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            # conv3x3_pc<RPW, NT, NLW, OUT, S9, POLY, TM, CV, NSI>: the key keeps the first four numbers (bench.py's kernel names); the 7x7 / polyphase /
            # canvas variants get a suffix
            if "hr_chain_kernel" in r["Kernel_Name"]:          # csrc/hr_chain.hip (round 6): bench.py's kernel_key of launch kind 5000 + ..
                t = traffic.setdefault("hr_chain_kernel<upconv+HR_conv0+conv_last>", {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                t[counter][0] += float(r["Counter_Value"]); t[counter][1] += 1
                continue
            m = re.search(r"(conv3x3_mfma|conv3x3_pc)<([^>]*)>", r["Kernel_Name"])
            if not m:
                continue
            a = [v.strip() for v in m.group(2).split(",")]
            nnum = 4 if m.group(1) == "conv3x3_pc" else 3
            key = m.group(1) + "<" + ",".join(a[:nnum]) + ">"
            if m.group(1) == "conv3x3_pc":
                key += ("+s9" if len(a) > 4 and a[4] == "true" else "") + ("+poly" if len(a) > 5 and a[5] == "true" else "") + \
                       ("+tm" + a[6] if len(a) > 6 and a[6] != "511" else "") + ("+cv" if len(a) > 7 and a[7] == "true" else "")
            t = traffic.setdefault(key, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            t[counter][0] += float(r["Counter_Value"]); t[counter][1] += 1
out = {}
for k, t in traffic.items():
    if t["FETCH_SIZE"][1] and t["WRITE_SIZE"][1]:
        f = t["FETCH_SIZE"][0] / t["FETCH_SIZE"][1]; w = t["WRITE_SIZE"][0] / t["WRITE_SIZE"][1]
        out[k] = {"fetch_kib_avg": f, "write_kib_avg": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                  "dispatches": t["FETCH_SIZE"][1]}
json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print("== traffic.json ==", json.dumps(out))
