#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only beside them) of the kernels beyond the RRDB convs (VERDICT r4 missing 3):
# PAN 540x960, UNet_256 x64, SRResNet 1080p in their fp16 engines.  scripts/r5/traffic_other.sh  ->  gpurun_out/traffic_other/{summary.txt,traffic_other.json}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ARCH in pan p2p_256 srgan; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/${ARCH}_$CNT -- python3 $ROOT/scripts/r5/fp32_once.py $ARCH fp16 > $OUT/${ARCH}_$CNT.out 2> $OUT/${ARCH}_$CNT.err
  done
done
cd $ROOT
python3 - <<PY
import csv, glob, json, re, collections
OUT = "$OUT"
res = {}
for arch in ("pan", "p2p_256", "srgan"):
    t = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0], "ns": [0.0, 0]})
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{OUT}/{arch}_{cnt}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != cnt: continue
                name = r["Kernel_Name"].replace("innfer::(anonymous namespace)::", "").replace("innfer::", "").replace("void ", "")
                name = re.sub(r"\(.*", "", name)[:90]
                a = t[name][cnt]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for f in glob.glob(f"{OUT}/{arch}_{cnt}/**/*kernel_trace.csv", recursive=True):
            if cnt != "FETCH_SIZE": continue
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("innfer::(anonymous namespace)::", "").replace("innfer::", "").replace("void ", "")
                name = re.sub(r"\(.*", "", name)[:90]
                a = t[name]["ns"]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
    out = {}
    for k, v in t.items():
        if v["FETCH_SIZE"][1] and v["WRITE_SIZE"][1]:
            fk = v["FETCH_SIZE"][0] / v["FETCH_SIZE"][1]; wk = v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1]
            ns = v["ns"][0] / max(1, v["ns"][1])
            out[k] = {"fetch_kib_avg": fk, "write_kib_avg": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "dispatches": v["FETCH_SIZE"][1], "avg_us_under_pmc": ns / 1e3}
    res[arch] = out
json.dump(res, open(f"{OUT}/traffic_other.json", "w"), indent=1)
with open(f"{OUT}/summary.txt", "w") as fo:
    for arch, out in res.items():
        fo.write(f"== {arch}: HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md), per kernel, 6 forwards\\n")
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"]):
            fo.write(f"  {k:92s} n={v['dispatches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  {v['avg_us_under_pmc']:9.1f} us  {v['hbm_bytes_per_launch'] / max(1e-9, v['avg_us_under_pmc']) / 1e6:7.2f} TB/s\\n")
print(open(f"{OUT}/summary.txt").read())
PY
