# experiment driver: specialised epilogue
mkdir -p gpurun_out
for st in 0 2; do
  echo "PERSIST=$st" >> gpurun_out/exp6.log
  INNFER_PERSIST=$st timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/exp6.log
done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/exp6.log
INNFER_PERSIST=0 INNFER_LIB=innfer_amd/lib/libinnfer_amd_stamps.so timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/exp6.log
cat gpurun_out/exp6.log
