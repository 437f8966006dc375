#!/bin/bash
# same-box A/B of pan_scpa_fused: the previous library, this one with 8 waves per workgroup, this one with 16 (INNFER_SCPA_NW, experiment build only)
for i in 1 2; do
  for cfg in prev nw8 nw16; do
    unset INNFER_LIB INNFER_SCPA_NW
    case $cfg in
      prev) export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_prev.so;;
      nw8) export INNFER_SCPA_NW=8;;
      nw16) export INNFER_SCPA_NW=16;;
    esac
    echo "== $cfg (round $i) INNFER_LIB=${INNFER_LIB:-} NW=${INNFER_SCPA_NW:-}"
    python3 scripts/bench_pan.py 2>&1 | grep "one launch" | grep -v "N= 1 200"
  done
done
