cd /tmp && export TMPDIR=/tmp
for A in 0 1 2 4 8; do
  UNET_N=64 INNFER_ABL=$A INNFER_LIB=$GRAFT_REPO_ROOT/innfer_amd/lib/libinnfer_amd_ablate.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_unet_$A -- python3 $GRAFT_REPO_ROOT/scripts/bench_unet.py > /dev/null 2>&1
  echo "ABL=$A"; python3 - $GRAFT_REPO_ROOT/gpurun_out/abl_unet_$A <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:30]:
    if "conv3x3_pc" in r["Name"] or "first" in r["Name"]:
        print("  ", r["Name"][:70].ljust(70), r["Calls"].rjust(4), r["AverageNs"][:9].rjust(10))
PY
done
