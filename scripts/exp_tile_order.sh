# experiment driver: store cache policy (INNFER_STORE) on the frame1080 bench
mkdir -p gpurun_out
for st in 0 1 2 3 0; do
  echo "STORE=$st" >> gpurun_out/exp3.log
  INNFER_STORE=$st timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/exp3.log
done
INNFER_STORE=2 timeout 600 python -m pytest tests -m gpu -x -q -k "conv or rrdb or golden" 2>&1 | tail -3 >> gpurun_out/exp3.log
cat gpurun_out/exp3.log
