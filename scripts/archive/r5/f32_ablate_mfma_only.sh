cd ${GRAFT_REPO_ROOT:-.}
export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so
for abl in 0 14 15; do
  echo "=== INNFER_F32_ABL=$abl"
  INNFER_F32_ABL=$abl DETAIL=1 python scripts/r5/fp32_breakdown.py p2p_256 2>&1 | grep -E "fp32 mode|34360, 570425|68719, 100663|34360, 1140851"
done
