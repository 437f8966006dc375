cd ${GRAFT_REPO_ROOT:-.}
export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so
for abl in 0 1 2 4 8 3 7 15; do
  echo "=== INNFER_F32_ABL=$abl"
  INNFER_F32_ABL=$abl DETAIL=1 python scripts/r5/fp32_breakdown.py p2p_256 2>&1 | grep -E "\(537, 655\)|\(134, 655\)|2147, 2621|8590, 10486|1074, 4719|4295, 18874|17180, 75497|34360, 41943" | cut -c1-140
done
