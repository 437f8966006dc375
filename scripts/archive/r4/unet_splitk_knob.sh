#!/bin/bash
# Split-K segment counts of the gather GEMM (diagnostic library knobs): 256-step / 128-step / 64-step layers; x64 and x1 forwards, each setting twice, interleaved
export INNFER_LIB=innfer_amd/lib/libinnfer_amd_ablate.so UNET_REPS=300
for rep in 1 2; do
for w in "8 4 2" "8 2 2" "4 2 2" "4 4 2" "8 2 1" "8 4 1"; do set -- $w
  echo -n "WANT256=$1 WANT128=$2 WANT64=$3:  "
  INNFER_GG_WANT256=$1 INNFER_GG_WANT128=$2 INNFER_GG_WANT64=$3 UNET_N=64 python3 scripts/bench_unet.py 2>&1 | grep "N=64" | cut -c1-40 | tr '\n' ' '
  INNFER_GG_WANT256=$1 INNFER_GG_WANT128=$2 INNFER_GG_WANT64=$3 UNET_N=1 python3 scripts/bench_unet.py 2>&1 | grep "N= 1" | cut -c1-40
done; done
