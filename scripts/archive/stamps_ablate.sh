export INNFER_PERSIST=0 INNFER_LIB=innfer_amd/lib/libinnfer_amd_stamps.so STAMPS_CASES=2
for cfg in "0 1" "6 0" "6 1" "0 0"; do set -- $cfg; echo "=== ABL=$1 PREFETCH=$2"; INNFER_ABL=$1 INNFER_PREFETCH=$2 timeout 200 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids | grep -E "^C=|prolog_done|c0_issued|c0_landed|c0_done|c1_issued|c1_done|loop_done|stores_"; done
