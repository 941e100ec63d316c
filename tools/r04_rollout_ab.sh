#!/bin/bash
# per-kernel durations of one solve alone (rocprofv3 --kernel-trace --stats) for library variants: bash tools/r04_rollout_ab.sh <lib>...
root=$(cd "$(dirname "$0")/.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for v in "$@"; do
  case $v in base) L=cilqr_amd/lib/libcilqr_hip.so;; *) L=cilqr_amd/lib/variants/libcilqr_hip_$v.so;; esac
  out=gpurun_out/rab_$v; rm -rf $out; mkdir -p $out
  CILQR_LIB=$L rocprofv3 --kernel-trace --stats -d $out/kt -- python bench.py --steps 6 --warmup 2 --in-flight 1 --pipeline 1 --cpu-sample 0 --no-latency > $out/bench.json 2> $out/err.txt
  echo "== $v"; python tools/prof_summary.py $out/kt 2>/dev/null | grep -E "forward|k_init_guess|TOTAL" | cut -c1-110
  rm -rf $out/kt
done
