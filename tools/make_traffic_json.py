#!/usr/bin/env python
"""profiles/backward_traffic.json from the two PMC passes (tools/pmc_kernel.py outputs) and a bench JSON."""
import json
import sys

fetch_txt, write_txt, bench_json, out = sys.argv[1:5]


def parse(path):
    lines = open(path).read().strip().split("\n")
    first = lines[1].split()
    tot = lines[-1].split()
    return float(first[2]), float(tot[2]), int(tot[1])


ff, fs, n = parse(fetch_txt)
wf, ws, _ = parse(write_txt)
b = json.load(open(bench_json))
N = b["config"]["n_steps"]
B = b["config"]["batch_per_gpu"]
al = b["roofline"].get("all_launches", b["roofline"])
n_act_sum = al["mean_problems_per_launch"] * al["launches"] / b["steps"]
hbm = fs * 1024 * 2 + ws * 1024
o = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile",
    "kernel": "cilqr::k_backward", "launches": n,
    "correction": "FETCH_SIZE counts 64 B per 128 B request for 16 B/lane coalesced loads on gfx950 -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; both in KiB",
    "fetch_size_kib_sum": fs, "write_size_kib_sum": ws, "hbm_bytes_sum": hbm,
    "problem_steps_sum": n_act_sum * N,
    "hbm_bytes_per_problem_step_all_launches": hbm / (n_act_sum * N),
    "full_batch_launch": {
        "problems": B, "fetch_kib": ff, "write_kib": wf, "hbm_bytes": ff * 2048 + wf * 1024,
        "expected_real_bytes": B * ((N * 18 + 9) * 16 + (N * 7) * 16 + 24),
        "algorithmic_bytes": B * (N * 110 + 44) * 8,
        "hbm_bytes_per_problem_step": (ff * 2048 + wf * 1024) / (B * N)},
}
json.dump(o, open(out, "w"), indent=1)
print(json.dumps(o["full_batch_launch"]), o["hbm_bytes_per_problem_step_all_launches"])
