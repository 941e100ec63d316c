#!/usr/bin/env python
"""Prints the status table of README.md from the committed profiles of a round:  python tools/readme_status.py r06"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def J(name):
    return json.load(open(os.path.join(P, f"{tag}_{name}.json")))


def M(x):
    return f"{x / 1e6:.2f} M"


def k(x):
    return f"{x / 1e3:.0f} k"


b = J("bench")
r = b["roofline"]
lat = b["latency"]
c1, c4, c4x = J("bench_config1_ped6_b4096"), J("bench_config4_dyn20_n100"), J("bench_config4_dyn20x_n100")
dp, ft, h3 = J("bench_dp_coarse"), J("bench_fast_lane_ties"), J("bench_three_handles")
pc, e2e = b["pcie_inclusive"], b["end_to_end"]
pr = J("parity_report_8192")
stable = sum(v["oracle_stable"] for v in pr["families"].values())
differ = sum(len(v["stable_but_different"]) for v in pr["families"].values())
confirmed = sum(len(v.get("stable_but_different_confirmed", v["stable_but_different"])) for v in pr["families"].values())
looks = [l for v in pr["families"].values() for l in v.get("stable_but_different_second_look", [])]
steps = sum(v["steps"]["replayed"] for v in pr["families"].values())
within = sum(v["steps"]["within_1e-8"] for v in pr["families"].values())
exc = sum(v["steps"]["excused_discontinuous_in_oracle"] for v in pr["families"].values())
failed = sum(v["steps"]["n_failed"] for v in pr["families"].values())
rows = [
    ("CILQR solves/s, 1 GPU (target ≥ 100 k): a pool of two handles on the GPU (`cilqr_pool_*`), two solves in flight on each, "
     f"{b['device_bytes'] / 1e9:.1f} GB, the reference's exact lane-tie rule",
     f"**{M(b['value'])}** ({b['ms_per_step']:.1f} ms per 65536-problem solve; three handles: {M(h3['value'])}; round 5: 2.06 M)"),
    ("the same through ONE handle with two solves in flight (`cilqr_submit` / `cilqr_wait`), "
     f"{b['one_handle']['device_bytes'] / 1e9:.1f} GB",
     f"{M(b['one_handle']['value'])} ({b['one_handle']['ms_per_step']:.1f} ms)"),
    ("one batch at a time (`cilqr_solve_batch`)", f"{M(b['single_batch']['value'])} ({b['single_batch']['ms_per_step']:.1f} ms)"),
    ("drop-in call: `planning::IlqrOptimizer::Plan` through the C++ adapter, batch of ONE, 256 scenes per family",
     "; ".join(f"{f}: mean {lat[f]['plan_b1']['mean_ms']:.2f} ms, median {lat[f]['plan_b1']['median_ms']:.2f}, p95 {lat[f]['plan_b1']['p95_ms']:.2f} "
               f"(CPU restatement on the same scenes: {lat[f]['cpu_restatement']['mean_ms']:.1f} / {lat[f]['cpu_restatement']['median_ms']:.1f} / {lat[f]['cpu_restatement']['p95_ms']:.1f})"
               for f in ("mix11", "ped6"))
     + (f"; dyn20 (N = 100, 20 obstacles, Cmax = 24): {lat['dyn20']['plan_b1']['mean_ms']:.2f} / {lat['dyn20']['plan_b1']['median_ms']:.2f} / "
        f"{lat['dyn20']['plan_b1']['p95_ms']:.2f} ms against {lat['dyn20']['cpu_restatement']['mean_ms']:.1f} / "
        f"{lat['dyn20']['cpu_restatement']['median_ms']:.1f} / {lat['dyn20']['cpu_restatement']['p95_ms']:.1f}" if "dyn20" in lat else "")),
    ("the same stream of batches with every array in pageable HOST memory, in and out (`pcie_inclusive`: upload ahead on a transfer "
     "thread, live cost rows packed; bit-identical to the device-resident result)",
     f"**{M(pc['value'])}** ({pc['ms_per_step']:.1f} ms per step, {pc['host_cores_busy']:.1f} host cores; round 5: 454 k)"),
    ("obstacle points → `cilqr_build_corridors` → solve, pooled (`end_to_end`)",
     f"**{M(e2e['value'])}** ({e2e['ms_per_step']:.1f} ms per step; corridor kernel alone {e2e['corridor_ms_alone']:.1f} ms, one batch at a time "
     f"{M(e2e['sequential_value'])}; round 5: 1.04 M, 23.4 ms)"),
    ("CPU oracle, 1 thread, same scenes (all cores of the box's quota)", f"{b['cpu_baseline']['value']:.0f} solves/s ({b['cpu_baseline']['all_cores']['value'] / 1e3:.1f} k)"),
    ("`k_backward`, launch over the whole batch, alone on the GPU",
     f"{r['full_batch_avg_launch_ms']:.3f} ms: {r['frac_full_batch']:.2f} of the 8 TB/s peak on its HBM traffic as counted in the run "
     f"({r['bytes_per_problem_step_full_batch_launch']:.0f} B per problem-step, `rocprofv3 --pmc`); {r['frac_full_batch_algorithmic']:.2f} × peak on the dense SURVEY §8(d) bytes (34 of 96 scalars are stored)"),
    ("backward pass, all launches of a solve (65536 → 256 problems), alone on the GPU (`roofline.frac`)",
     f"{r['frac']:.2f} of peak on counted bytes ({r['bytes_per_problem_step']:.0f} B per problem-step), {r['frac_algorithmic']:.2f} on dense bytes; round 5: 0.35 / 0.75 (next step's operands now prefetched into LDS)"),
    ("configs[1] (B = 4096) / configs[4] (B = 65536, N = 100; barriers active at the init guess)",
     f"{k(c1['value'])} / {M(c4['value'])} ({k(c4x['value'])}) solves/s"),
    ("DP coarse planner → corridor producer → solver (`bench.py --coarse dp`)", f"{M(dp['value'])} solves/s"),
    ("the opt-in fast lane-tie rule (`CILQR_OPT_EXACT_LANE_TIES` = 0)", f"{M(ft['value'])} solves/s"),
    ("host CPU of one rank in the timed region (16-core quota on the GPU box; eight ranks share it)",
     f"{b['host']['cores_busy_all_ranks']:.2f} cores ({b['host']['cpu_s_per_step_max_rank'] * 1e3:.0f} CPU-ms per step); with spinning waits: "
     f"{J('bench_host_wait_spin')['host']['cores_busy_all_ranks']:.2f} cores"),
    (f"parity vs oracle (32768 scenes, `profiles/{tag}_parity_report_8192.json`, exact lane ties, no `lane_tie` excuse)",
     f"{confirmed} of {stable} oracle-stable problems differ at 1e-4"
     + (f" ({differ} differ among those the 8-run mask called stable: " + ", ".join(f"the oracle itself ends elsewhere in {l['ended_elsewhere']} of {l['oracle_reruns']} further re-runs and {l.get('library_result_equals_a_perturbed_oracle_ending')} of them end where the library does" for l in looks) + ")" if differ else "")
     + f"; {within} of {steps} iteration steps replay in the oracle at 1e-8, {exc} are shown "
     f"discontinuous there, {failed} fail; 4–9 % of scenes are chaotic in the oracle itself (DESIGN.md §5)"),
]
print("| quantity | value |\n|---|---|")
for a, v in rows:
    print(f"| {a} | {v} |")
