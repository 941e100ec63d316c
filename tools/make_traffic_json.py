#!/usr/bin/env python
"""profiles/backward_traffic.json from PMC passes (tools/pmc_kernel.py outputs) and the bench JSON of the same command.
    python tools/make_traffic_json.py <out.json> <workload> <fetch.txt> <write.txt> <bench.json> [<workload> <fetch> <write> <bench> ...]
Every workload gets an entry under "by_workload" ("<scene>_n<N>"); bench.py looks its own workload up there."""
import json
import sys


def parse(path, kernel=None):
    lines = [l for l in open(path).read().strip().split("\n") if l.strip()]
    first = lines[1].split()
    tot = [l for l in lines if l.startswith("SUM")][0].split()
    solves = [int(l.split()[1]) for l in lines if l.startswith("SOLVES")]
    return float(first[2]), float(tot[2]), int(tot[1]), (solves[0] if solves else 1)


def entry(fetch_txt, write_txt, bench_json):
    ff, fs, n, solves = parse(fetch_txt)
    wf, ws, _, _ = parse(write_txt)
    b = json.loads(open(bench_json).read().strip().splitlines()[-1])
    N = b["config"]["n_steps"]
    B = b["config"]["batch_per_gpu"]
    rf = b["roofline"]
    # the capture holds `solves` identical solves of the same batch (calibration, single-batch timing, the timed step)
    # roofline.launches / mean_problems_per_launch describe ONE solve (round 4 on; earlier records: the whole timed region)
    per_solve = rf["mean_problems_per_launch"] * rf["launches"] / (1 if "launches_contended" in rf else b["steps"])
    n_act_sum = solves * per_solve
    hbm = fs * 1024 * 2 + ws * 1024
    return {
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python bench.py --steps 1 --warmup 0 --in-flight 1 --cpu-sample 0 [--scene ...]",
        "kernel": "cilqr::k_backward + cilqr::k_backward_team", "launches": n, "solves_in_capture": solves,
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


out = sys.argv[1]
args = sys.argv[2:]
doc = {"by_workload": {}}
for i in range(0, len(args), 4):
    wl, f, w, bj = args[i:i + 4]
    doc["by_workload"][wl] = entry(f, w, bj)
    print(wl, json.dumps(doc["by_workload"][wl]["full_batch_launch"]), doc["by_workload"][wl]["hbm_bytes_per_problem_step_all_launches"])
json.dump(doc, open(out, "w"), indent=1)
