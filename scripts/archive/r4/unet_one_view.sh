#!/bin/bash
# A/B of the UNet's one stored skip form (ReLU applied by the up conv as it reads) against the two stored views, on the diagnostic library; interleaved.
export INNFER_LIB=innfer_amd/lib/libinnfer_amd_ablate.so UNET_N=64 UNET_REPS=200
for i in 1 2 3; do
  echo "one view:"; python3 scripts/bench_unet.py 2>&1 | grep "N=64"
  echo "two views:"; INNFER_UNET_TWO_VIEWS=1 python3 scripts/bench_unet.py 2>&1 | grep "N=64"
done
