#!/bin/bash
# Sample sclk / power with rocm-smi while bench.py runs (is the trunk clock- / power-limited?).  Output: gpurun_out/clk_samples.txt
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 400 --warmup 5 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
for i in $(seq 1 45); do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed 's/GPU\[0\]//; s/[[:space:]]\+/ /g' | tr '\n' ';'
  echo
  sleep 0.7
done > gpurun_out/clk_samples.txt
wait $BP
rocm-smi --showmaxpower 2>&1 | grep -i power > gpurun_out/clk_caps.txt
tail -1 gpurun_out/clk_bench.log | cut -c1-200
cat gpurun_out/clk_caps.txt
cat gpurun_out/clk_samples.txt | cut -c1-200
