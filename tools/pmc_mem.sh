#!/bin/bash
# Memory-path counters (TA / TCP / TCC) of every kernel of one solve: gpurun_out/<tag>_pmc_mem_{1,2,3}.txt
tag=${1:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$root"
i=0
for set in "TA_TA_BUSY TA_TOTAL_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ GRBM_GUI_ACTIVE" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA_RDREQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$out/pmcm_$i" -- python bench.py --steps 1 --warmup 0 --in-flight 1 --pipeline 1 --cpu-sample 0 > "$out/pmcm_$i.json" 2> "$out/pmcm_$i.err"
  python tools/pmc_summary.py "$out/pmcm_$i" > "$out/${tag}_pmc_mem_$i.txt" 2>&1
  rm -rf "$out/pmcm_$i"
done
head -12 "$out"/${tag}_pmc_mem_*.txt
