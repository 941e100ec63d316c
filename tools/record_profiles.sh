#!/bin/bash
# Records the measurement set of one round on the GPU box into gpurun_out/<tag>/ :
#   bench JSON, the same command under rocprofv3 --kernel-trace --stats (+ summaries; --pipeline 1
#   drops the 3-batches-in-flight extra AFTER the timed region, whose overlapping streams would
#   inflate every kernel's duration in the per-kernel averages), the two PMC
#   passes for HBM traffic of k_backward (separate runs, counters only), the PCIe-inclusive rate.
#   usage (through gpurun): bash tools/record_profiles.sh r01
# then copy gpurun_out/<tag>/*.{json,txt} into profiles/ (see DESIGN.md "Measurement").
set -u
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
python bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -d "$out/kt" -- python bench.py --pipeline 1 > "$out/${tag}_bench_under_rocprofv3.json" 2> "$out/kt.err"
{ python tools/prof_summary.py "$out/kt" --iters 1,2,5,10,20,40,80; python tools/phase_summary.py "$out/kt"; } > "$out/${tag}_kernel_stats.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" -- python bench.py --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
  python tools/pmc_kernel.py "$out/pmc_$c" k_backward > "$out/${tag}_pmc_backward_$c.txt" 2>&1
done
python tools/make_traffic_json.py "$out/${tag}_pmc_backward_FETCH_SIZE.txt" "$out/${tag}_pmc_backward_WRITE_SIZE.txt" "$out/${tag}_bench.json" "$out/backward_traffic.json"
python tools/pcie_rate.py > "$out/${tag}_pcie_inclusive.json" 2> "$out/pcie.err"
ls -la "$out"
