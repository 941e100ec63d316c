#!/bin/bash
# kernel time of k_cost_knots (full batch, init-guess trajectories) for the library variants given as arguments
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  rm -rf gpurun_out/abl_$name
  CILQR_LIB=$lib rocprofv3 --kernel-trace --stats -d gpurun_out/abl_$name -- python tools/cost_microbench.py > gpurun_out/abl_$name.log 2>&1
  echo "== $name"; tail -1 gpurun_out/abl_$name.log
  python tools/prof_summary.py gpurun_out/abl_$name 2>/dev/null | grep -E "k_cost_knots|k_reduce_only"
done
