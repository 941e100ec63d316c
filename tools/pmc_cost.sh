#!/bin/bash
# PMC passes over tools/cost_microbench.py (k_cost_knots at full batch); counters only (no kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_cost_$i
  rocprofv3 --pmc $set -d gpurun_out/pmc_cost_$i -- python tools/cost_microbench.py > gpurun_out/pmc_cost_$i.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_cost_$i 2>&1 | grep -E "kernel|k_cost_knots"
done
