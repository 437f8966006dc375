#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   scripts/profile.sh <tag> [bench args...]
# 1. --kernel-trace --stats  -> per-kernel time summary
# 2. --pmc FETCH_SIZE / --pmc WRITE_SIZE in their own passes -> HBM bytes per launch
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --sharded-steps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
cd $ROOT
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
