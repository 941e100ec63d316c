#!/bin/bash
# Runs bench.py for a list of "name|CILQR_LIB path or -|extra bench args" variants (tuning experiments on the GPU box).
# usage: tools/bench_variants.sh <out-prefix> "name|lib|args" ...
out=$1; shift
mkdir -p "$(dirname "$out")"
for v in "$@"; do
  IFS='|' read -r name lib args <<< "$v"
  if [ "$lib" = "-" ]; then unset CILQR_LIB; else export CILQR_LIB="$lib"; fi
  python bench.py --steps 4 --warmup 1 --in-flight 1 --cpu-sample 0 --no-latency --no-traffic $args > "${out}_${name}.json" 2> "${out}_${name}.err"
  python - "$name" "${out}_${name}.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", r["value"], "ms", r["ms_per_step"], "breakdown", r["breakdown_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
