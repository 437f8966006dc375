#!/bin/bash
# A/B of one bench.py flag on ONE box: scripts/r4/ab_flags.sh <tag> "<flag for variant B>" [reps]
# interleaved runs (A B A B ..), frame workload only; prints ms/frame per run and the per-kernel table of the last pair
TAG=$1; FLAG=$2; REPS=${3:-2}
OUT=gpurun_out/r4_$TAG; mkdir -p $OUT
for i in $(seq 1 $REPS); do
  python3 bench.py --steps 20 --warmup 5 --no-extras --sharded-steps 0 --no-cpu-baseline --no-power-probe > $OUT/a$i.json 2> $OUT/a$i.err
  python3 bench.py --steps 20 --warmup 5 --no-extras --sharded-steps 0 --no-cpu-baseline --no-power-probe $FLAG > $OUT/b$i.json 2> $OUT/b$i.err
done
python3 - "$OUT" "$FLAG" $REPS <<'PY'
import json, sys
out, flag, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
for i in range(1, reps + 1):
    for v, name in (("a", "default"), ("b", flag)):
        d = json.loads(open(f"{out}/{v}{i}.json").read().strip().splitlines()[-1])
        pk = d["roofline"]["per_kernel"]
        print(f"run {i} {name:24s} {d['ms_per_step']:8.3f} ms/frame  " + "  ".join(f"{k}: n={e['launches']} {e['ms_total']:.3f}" for k, e in pk.items()))
PY
grep -h "conv3x3_pc<2,4,4,0> " $OUT/a$REPS.err $OUT/b$REPS.err
