#!/bin/bash
# does re-packing the survivors EARLIER (before the active list of a big backward launch gets sparse)
# pay?  CILQR_OPT_COMPACTION re-packs when the survivors fill at most P % of the occupied slots (default 75).  For each P:
# the HBM bytes per problem-step of the backward launches (PMC passes inside bench.py), the backward roofline of one solve
# alone, the sequential solve and the pooled throughput.   usage (through gpurun): bash tools/r04_compaction_sweep.sh
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/${OUTDIR:-r04_compaction}
mkdir -p "$out"
cd "$root"
for p in ${PERCENTS:-75 85 92 100}; do
  python bench.py --compact-percent $p --cpu-sample 0 --no-latency > "$out/compact_$p.json" 2> "$out/compact_$p.err"
done
python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
print("percent  value  single_batch  one_handle  bytes/problem-step  frac  frac_alg  frac_full  avg_launch_ms  bwd_ms/solve  other_ms/solve")
for p in [int(x) for x in os.environ.get("PERCENTS", "75 85 92 100").split()]:
    try:
        d = json.load(open(os.path.join(out, f"compact_{p}.json")))
    except Exception as e:
        print(p, "failed", e); continue
    r = d["roofline"]; b = d["breakdown_ms_per_step"]
    print(p, d["value"], d["single_batch"]["value"], d["one_handle"]["value"], r["bytes_per_problem_step"], r["frac"], r["frac_algorithmic"],
          r["frac_full_batch"], round(r["avg_launch_ms"], 4), b["bwd_ms"], b["other_ms"])
PY
