export UNET_N=64 UNET_REPS=300
for i in 1 2 3; do
  echo -n "plane order (shipped): "; python3 scripts/bench_unet.py 2>&1 | grep "N=64"
  echo -n "lane-contiguous (A/B): "; INNFER_LIB=innfer_amd/lib/libinnfer_amd_unetold.so python3 scripts/bench_unet.py 2>&1 | grep "N=64"
done
