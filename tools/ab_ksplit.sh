#!/bin/bash
# knot split A/B: CILQR_KBLOCKS = 0 (one knot per block, as in round 2), 1536, 3072, 6144, 12288; staged variants with the split
mkdir -p gpurun_out
for v in 0 768 1536 3072 6144 12288; do
  CILQR_KBLOCKS=$v python bench.py --steps 4 --warmup 1 --cpu-sample 0 --in-flight 1 > gpurun_out/ks_seq_$v.json 2>/dev/null
done
for v in 1536 3072; do
  CILQR_KBLOCKS=$v python bench.py --steps 6 --warmup 2 --cpu-sample 0 > gpurun_out/ks_pipe_$v.json 2>/dev/null
done
for st in 1 2; do
  CILQR_STAGED=$st CILQR_KBLOCKS=3072 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --in-flight 1 > gpurun_out/ks_seq_3072_staged$st.json 2>/dev/null
done
