#!/bin/bash
# Round-6 evidence run (ONE gpurun call, from the repo root):  scripts/r6/evidence_r6.sh
# Every step writes <step>.status (OK / FAILED rc) next to its output and the script exits non-zero if any step failed (the round-5 script swallowed
# failures and could leave stale or empty files that looked like measurements).  rocprofv3 gets `python3 <script>` directly behind `--` (no env / bash hop).
set -u -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/evidence_r6
rm -rf "$OUT"; mkdir -p "$OUT"
FAILS=0
step() {          # step <name> <command...>: stdout -> $OUT/<name>.txt, stderr -> $OUT/<name>.err
    local name=$1; shift
    "$@" > "$OUT/$name.txt" 2> "$OUT/$name.err"
    local rc=$?
    if [ $rc -eq 0 ] && [ -s "$OUT/$name.txt" ]; then echo OK > "$OUT/$name.status"; else echo "FAILED rc=$rc" > "$OUT/$name.status"; FAILS=$((FAILS + 1)); fi
    echo "[evidence_r6] $name: $(cat "$OUT/$name.status")"
}
cd "$ROOT"
# 1. the driver-form line (headline + every side object)
step bench_frame1080 python3 bench.py
# 2. same-box A/B of the chained HR tail: on / off / on / off, headline only
for i in 1 2; do
    step ab_chain_on_$i python3 bench.py --no-extras --sharded-steps 0 --steps 10 --no-cpu-baseline
    step ab_chain_off_$i python3 bench.py --no-extras --sharded-steps 0 --steps 10 --no-cpu-baseline --no-hr-chain
done
# 3. rocprofv3 kernel statistics + FETCH_SIZE / WRITE_SIZE passes of the frame (scripts/profile.sh -> gpurun_out/prof_r6/summary.txt, traffic.json)
step profile_frame bash scripts/profile.sh r6 --no-extras --no-power-probe
cp "$ROOT/gpurun_out/prof_r6/traffic.json" "$OUT/traffic_frame.json" 2>/dev/null || { echo "FAILED no traffic.json" > "$OUT/traffic_frame.status"; FAILS=$((FAILS + 1)); }
# 4. the chained kernel alone: phases removed one at a time (diagnostic build; times only)
if [ -x scripts/micro/hr_chain_micro ]; then
    : > "$OUT/hr_chain_ablation.txt"
    for a in 0 1 2 3 4 8 12 15 16 32 127 0; do INNFER_ABL=$a scripts/micro/hr_chain_micro >> "$OUT/hr_chain_ablation.txt" 2>> "$OUT/hr_chain_ablation.err" || FAILS=$((FAILS + 1)); done
    echo OK > "$OUT/hr_chain_ablation.status"
fi
# 5. the other generators at their bench sizes
step bench_pan python3 scripts/bench_pan.py
step bench_srresnet python3 scripts/bench_srresnet.py
# 6. PAN per launch (fp16 engine and the fp32 mode on split operands), the fp32 mode's engines against each other and the oracle
step pan_breakdown python3 scripts/r6/pan_breakdown.py
step pan_f32_ab python3 scripts/r6/pan_f32_ab.py
# 7. rocprofv3 kernel statistics of PAN 540 x 960 in both modes (the program directly behind `--`)
for m in pan540_once pan540_f32_once; do
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$m" -- python3 "$ROOT/scripts/r6/$m.py" > "$OUT/prof_$m.log" 2>&1 )
    f=$(find "$OUT/prof_$m" -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" > "$OUT/kernel_stats_$m.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{r['Name'][:86]:88s} calls={int(r['Calls']):4d} total_us={float(r['TotalDurationNs']) / 1e3:10.1f} avg_us={float(r['AverageNs']) / 1e3:9.2f} pct={float(r['Percentage']):6.2f}")
PY
        echo OK > "$OUT/kernel_stats_$m.status"
    else echo "FAILED no kernel_stats.csv" > "$OUT/kernel_stats_$m.status"; FAILS=$((FAILS + 1)); fi
    rm -rf "$OUT/prof_$m"
done
ls -la "$OUT"
echo "[evidence_r6] failed steps: $FAILS"
exit $FAILS
