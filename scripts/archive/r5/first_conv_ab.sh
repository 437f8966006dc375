#!/bin/bash
# row-walking first conv (round 5) against the group-striding one, same box, interleaved: SRResNet / RRDBNet 1080p forwards, UNet-less
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  for L in ${BASE:-build/libinnfer_amd_base.so} innfer_amd/lib/libinnfer_amd.so; do
    echo "$L: $(INNFER_LIB=$PWD/$L python scripts/r5/first_conv_time.py 2>&1 | grep 'per forward' | sed 's/; first.*//' | tr '\n' ' ')"
  done
done
