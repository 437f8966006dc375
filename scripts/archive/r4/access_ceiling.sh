#!/bin/bash
# times, then FETCH_SIZE / WRITE_SIZE per dispatch for the coupled (8) and free-running (40) data-movement-only forms and the shipped schedule (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4_ceiling; mkdir -p $OUT
python3 $ROOT/scripts/r4/access_ceiling.py > $OUT/times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  ABLS=0,8,40 REPS=5 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 $ROOT/scripts/r4/access_ceiling.py > $OUT/pmc_$ctr.txt 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
# dispatch order: per layer (5) x abl (0, 8, 40) x (3 warm-up + 5 timed) launches of conv3x3_pc; the slab fill / pack kernels are skipped by name
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{out}/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == ctr and "conv3x3_pc" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    res[ctr] = [float(r["Counter_Value"]) for r in rows]
layers = ["64->32", "96->32", "128->32", "160->32", "192->64"]
n = 8
print("layer    abl   FETCH_SIZE x2 (MB)   WRITE_SIZE (MB)   HBM bytes (MB)   [KiB counters; FETCH doubled per MI355X_MICROARCH.md]")
for li, name in enumerate(layers):
    for ai, abl in enumerate((0, 8, 40)):
        k = (li * 3 + ai) * n
        f = res["FETCH_SIZE"][k + 3:k + n]; w = res["WRITE_SIZE"][k + 3:k + n]
        if len(f) == 5 and len(w) == 5:
            fb, wb = 2 * sum(f) / 5 * 1024 / 1e6, sum(w) / 5 * 1024 / 1e6
            print(f"{name:8s} {abl:3d}   {fb:12.1f}        {wb:12.1f}     {fb + wb:12.1f}")
PY
cat $OUT/times.txt
