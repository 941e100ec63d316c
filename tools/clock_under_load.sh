#!/bin/bash
# shader clock and power while the bench workload runs (is the fp64 work power-limited?)
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 300 --warmup 2 --cpu-sample 0 --no-latency --no-traffic > gpurun_out/clk_bench.json 2>/dev/null &
pid=$!
for i in $(seq 1 60); do
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket Graphics' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
  kill -0 $pid 2>/dev/null || break
done
wait $pid
python -c "import json; r=json.load(open('gpurun_out/clk_bench.json')); print(r['value'], r['ms_per_step'])"
