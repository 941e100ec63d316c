#!/bin/bash
# quick check of the backward kernels' HBM traffic and duration on the default workload (one solve alone)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcq; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -- python bench.py --steps 1 --warmup 0 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > $out/$c.json 2> $out/$c.err
  python tools/pmc_kernel.py $out/$c k_backward > $out/pmc_$c.txt 2>&1
done
python tools/make_traffic_json.py $out/traffic.json mix11_n50 $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt $out/FETCH_SIZE.json
head -14 $out/pmc_FETCH_SIZE.txt
python bench.py --steps 4 --warmup 1 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > $out/seq.json 2>/dev/null
python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-latency > $out/pipe.json 2>/dev/null
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
