#!/bin/bash
# handles x solves in flight (and option sweeps beside them): what the GPU gives when more than one first stage is resident
# usage (gpurun): bash tools/bench_r03_handles.sh "<handles> <in-flight> [extra bench flags]" ...   results under gpurun_out/handles/
mkdir -p gpurun_out/handles
i=0
for cfg in "$@"; do
  set -- $cfg
  h=$1; f=$2; shift 2
  i=$((i+1))
  python bench.py --steps 30 --warmup 8 --pipeline $h --in-flight $f --no-traffic --no-latency --cpu-sample 0 --cpu-configs 0 "$@" \
     > gpurun_out/handles/run$i.json 2> gpurun_out/handles/run$i.err
  python - "$h" "$f" "$*" gpurun_out/handles/run$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(f"handles {sys.argv[1]} in-flight {sys.argv[2]} {sys.argv[3]}: {d['value']:.0f} solves/s {d['ms_per_step']} ms {d['device_bytes']/1e9:.1f} GB identical={d.get('results_identical_across_solves_in_flight')} one_handle={(d.get('one_handle') or {}).get('value')} single={(d.get('single_batch') or {}).get('value')} tail_ms={(d.get('breakdown_ms_per_step') or {}).get('tail_ms')}")
PY
done
