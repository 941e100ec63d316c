#!/bin/bash
# A/B of the LDS-staged cost kernel: CILQR_STAGED = 0 (registers), 1 (staged, 3 waves), 2 (staged, 2 waves, all cell words up front)
mkdir -p gpurun_out
for v in 0 1 2; do
  CILQR_STAGED=$v python bench.py --steps 4 --warmup 1 --cpu-sample 0 --in-flight 1 > gpurun_out/st_seq_$v.json 2>/dev/null
  CILQR_STAGED=$v python bench.py --steps 6 --warmup 2 --cpu-sample 0 > gpurun_out/st_pipe_$v.json 2>/dev/null
done
