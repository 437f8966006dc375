cd ${GRAFT_REPO_ROOT:-.}
export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_ablate.so
for rep in 1 2; do
for cfg in "1200 80" "0 76" "1200 76" "0 80" "2500 80"; do
  set -- $cfg
  echo "== unhidden $1 ldscap $2: $(INNFER_F32_UNHIDDEN=$1 INNFER_F32_LDSCAP=$2 python scripts/r5/fp32_breakdown.py pan p2p_256 resnet_9blocks wbcunet 2>&1 | grep 'fp32 mode' | sed 's/.*fp32 mode: //' | tr '\n' ' ')"
done
done
