#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for L in ${BASE:-build/libinnfer_amd_base.so} innfer_amd/lib/libinnfer_amd.so; do
    echo "$L: $(INNFER_LIB=$PWD/$L python scripts/r5/pan_tail_ab.py 2>&1 | tail -1)"
  done
done
