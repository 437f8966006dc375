#!/bin/bash
# same-box A/B of two library builds on a script that prints timings: scripts/r5/lib_ab.sh <base.so> <script.py> [args]   (three interleaved passes)
cd ${GRAFT_REPO_ROOT:-.}
B=$1; shift
for rep in 1 2 3; do
  for L in $B innfer_amd/lib/libinnfer_amd.so; do
    echo "== $L"; INNFER_LIB=$PWD/$L python "$@" 2>&1 | grep -v amdgpu.ids
  done
done
