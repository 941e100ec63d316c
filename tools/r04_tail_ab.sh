#!/bin/bash
# Round 4, tail-kernel work: GPU suite, the default bench line (with the drop-in latency), the per-phase cycle counts of k_tail
# (tuning build -DCILQR_TAIL_PROFILE) with the split quadratisation on and off.   usage (through gpurun): bash tools/r04_tail_ab.sh <tag>
tag=${1:-r04t}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
(time timeout 1500 python -m pytest tests -m gpu -x -q) > "$out/pytest.log" 2>&1
tail -4 "$out/pytest.log"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
python tools/tail_phase_profile.py 256 > "$out/tailprof_split.txt" 2>&1
CILQR_TAIL_QUAD_SPLIT=0 python tools/tail_phase_profile.py 256 > "$out/tailprof_unsplit.txt" 2>&1
tail -6 "$out/tailprof_split.txt" "$out/tailprof_unsplit.txt"
python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "one_handle", d["one_handle"]["value"], "single", d["single_batch"]["value"])
print("breakdown", d["breakdown_ms_per_step"])
for f in ("ped6", "mix11"):
    l = d["latency"][f]
    print(f, "plan_b1", l["plan_b1"], "phases", {k: l["plan_b1_phases"][k] for k in ("tail_kernel_ms", "load_initguess_export_ms", "iterations", "tail_us_per_iteration")}, "b64", l["solve_batch"]["mean_ms"])
r = d["roofline"]
print({k: r[k] for k in ("frac", "frac_full_batch", "frac_contended", "bytes_per_problem_step", "avg_launch_ms")})
PY
