#!/usr/bin/env python
"""bench.py -- CILQR solves/sec on MI355X (BASELINE.json metric).

One "step" = one cilqr_solve_batch over the whole per-GPU batch (inputs already resident in HBM):
load/prepare -> init guess -> lockstep iLQR iterations until every problem terminated -> export.
Workload (config.workload): BASELINE.json configs[2] -- batch 65536 per GPU, 50-step horizon,
6 pedestrians + 3 moving + 2 static vehicles ("mix11" scenes of cilqr_amd.scenario).  With
--gpus N every rank solves its own 65536 scenes (weak scaling, configs[3] at N=8) and the
results are gathered to rank 0 with one RCCL gather inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     backward-pass kernel: algorithmic bytes ((N*110+44)*8 B per problem
               per launch) / HIP-event time of the launches, against the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (single thread) on a bounded sample of the same scenes
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {("ped6", 4096): "configs[1]", ("mix11", 65536): "configs[2] (configs[3] when sharded over 8 GPUs)",
             ("dyn20", 65536): "configs[4]"}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU")
    ap.add_argument("--scene", default="mix11")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample", type=int, default=3072, help="problems timed on the CPU oracle (0 = skip); 3072 scenes = about 15 s on one core")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--compact-percent", type=int, default=-1, help="CILQR_OPT_COMPACTION value (tuning experiments)")
    ap.add_argument("--spec-threshold", type=int, default=-1, help="CILQR_OPT_SPEC_THRESHOLD value (tuning experiments)")
    ap.add_argument("--seq-rounds", type=int, default=-1, help="CILQR_OPT_SEQ_ROUNDS value (tuning experiments)")
    ap.add_argument("--team-threshold", type=int, default=-1, help="CILQR_OPT_TEAM_THRESHOLD value (tuning experiments)")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="after the timed region, also measure throughput with this many batches in flight "
                         "(one handle + stream + host thread each; 0/1 = skip; single-GPU runs only)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="extra (never `value`): obstacle points -> cilqr_build_corridors -> solve on the device")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "backward_traffic.json"))
    args = ap.parse_args()

    # stdout carries exactly one line (rank 0's JSON); everything else any library prints on fd 1
    # (RCCL's version banner, for one) is sent to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CILQR_BENCH_FORCE_DIST=1 exercises the RCCL path (process group, all-reduce, gather) even
    # with a single rank -- the only way to smoke-test it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("CILQR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # keep RCCL's log lines off stdout (rank 0 prints exactly one JSON line there, last)
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from cilqr_amd import api, scenario
    from cilqr_amd.distributed import gather_results

    spec = scenario.SPECS[args.scene]
    B, N, K, cmax = args.batch, spec.n_steps, spec.n_steps + 1, spec.cmax
    t0 = time.time()
    workers = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    sc = scenario.generate(spec, B, seed=args.seed, first_problem=rank * B, workers=workers)
    t_gen = time.time() - t0

    cfg = api.default_config(N)
    M = cfg.max_iter
    opt = api.BatchIlqrOptimizer(cfg, device=local_rank, batch_capacity=B, cmax=cmax,
                                 max_lane_segments=max(sc["left"].shape[0], sc["right"].shape[0]))
    opt.set_stream(torch.cuda.current_stream().cuda_stream)
    opt.set_profiling(not args.no_profile)
    if args.compact_percent >= 0:
        opt.set_option(api.OPT_COMPACTION, args.compact_percent)
    if args.seq_rounds >= 1:
        opt.set_option(api.OPT_SEQ_ROUNDS, args.seq_rounds)
    if args.spec_threshold >= 0:
        opt.set_option(api.OPT_SPEC_THRESHOLD, args.spec_threshold)
    if args.team_threshold >= 0:
        opt.set_option(api.OPT_TEAM_THRESHOLD, args.team_threshold)

    d_start = torch.from_numpy(sc["start"]).to(dev)
    d_coarse = torch.from_numpy(sc["coarse"]).to(dev)
    d_cor = torch.from_numpy(sc["corridor"]).to(dev)
    d_cnt = torch.from_numpy(sc["ccount"]).to(dev)
    left = np.ascontiguousarray(sc["left"])
    right = np.ascontiguousarray(sc["right"])
    prob = opt.make_problem(B, d_start.data_ptr(), d_coarse.data_ptr(), d_cor.data_ptr(), d_cnt.data_ptr(),
                            cmax, left.ctypes.data, right.ctypes.data, left.shape[0], right.shape[0],
                            api.MEM_DEVICE)
    o_traj = torch.zeros((B, K, 10), dtype=torch.float64, device=dev)
    o_hist = torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev)
    o_nc = torch.zeros(B, dtype=torch.int32, device=dev)
    o_st = torch.zeros(B, dtype=torch.int32, device=dev)
    o_ni = torch.zeros(B, dtype=torch.int32, device=dev)
    sol = api.SolutionBatch(api.MEM_DEVICE, 0, o_traj.data_ptr(), o_hist.data_ptr(), o_nc.data_ptr(),
                            o_st.data_ptr(), o_ni.data_ptr(), None, None)

    torch.cuda.synchronize()   # inputs uploaded and outputs zero-filled before the solver's own stream starts

    def step():
        rc = opt.solve_raw(prob, sol)
        if rc != api.OK:
            raise api.CilqrError(rc, "in bench step")
        if use_dist:
            # 8 of the 10 trajectory columns travel (time and kappa are functions of the others)
            return gather_results(o_traj, o_hist, o_nc, o_st, dst=0, densify=False, derive=(cfg.dt, cfg.wheel_base))
        return None

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up steps carry HIP events around every phase (the per-phase breakdown below; ~4 % slower);
    # the timed steps only around the backward launches, which is what the roofline needs (< 1 %).
    warm = dict(quad_ms=0.0, bwd_ms=0.0, ls_ms=0.0, other_ms=0.0, total_ms=0.0, steps=0)
    for _ in range(args.warmup):
        step()
        if not args.no_profile:
            p = opt.profile()
            warm["quad_ms"] += p.quadratize_ms
            warm["bwd_ms"] += p.backward_ms
            warm["ls_ms"] += p.linesearch_ms
            warm["other_ms"] += p.other_ms
            warm["total_ms"] += p.total_ms
            warm["steps"] += 1
    fence()
    if not args.no_profile:
        opt.set_profiling(2)
    prof_acc = dict(bwd_ms=0.0, bwd_launches=0, bwd_steps=0, iters=0, full_ms=0.0, full_launches=0)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        p = opt.profile()
        prof_acc["bwd_ms"] += p.backward_ms
        prof_acc["bwd_launches"] += p.backward_launches
        prof_acc["bwd_steps"] += p.backward_problem_steps
        prof_acc["iters"] += p.iterations
        prof_acc["full_ms"] += p.backward_full_ms
        prof_acc["full_launches"] += p.backward_full_launches
    fence()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Extra (never `value`): several batches in flight.  The tail of a solve (a few hundred straggler
    # problems for ~60 iterations) leaves the GPU mostly idle; a second batch on its own stream fills it.
    pipelined = None
    if world == 1 and args.pipeline > 1:
        P = args.pipeline
        ctx = []
        for i in range(P):
            o_i = opt if i == 0 else api.BatchIlqrOptimizer(
                cfg, device=local_rank, batch_capacity=B, cmax=cmax,
                max_lane_segments=max(sc["left"].shape[0], sc["right"].shape[0]))
            st_i = torch.cuda.Stream()
            o_i.set_stream(st_i.cuda_stream)
            o_i.set_profiling(False)
            bufs = (torch.zeros((B, K, 10), dtype=torch.float64, device=dev),
                    torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
            sol_i = api.SolutionBatch(api.MEM_DEVICE, 0, bufs[0].data_ptr(), bufs[1].data_ptr(),
                                      bufs[2].data_ptr(), bufs[3].data_ptr(), None, None, None)
            ctx.append((o_i, st_i, bufs, sol_i))
        torch.cuda.synchronize()
        per_handle = max(2, args.steps)
        errs = []
        for c in ctx:                      # warm-up (staging buffers, clocks)
            if c[0].solve_raw(prob, c[3]) != api.OK:
                errs.append("warm-up")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        inflight = [False] * P
        for s_ in range(P * per_handle):   # round-robin over the handles, each keeps one batch in flight
            i = s_ % P
            if inflight[i]:
                rc_ = ctx[i][0].wait()
                if rc_ != api.OK:
                    errs.append(rc_)
            rc_ = ctx[i][0].submit_raw(prob, ctx[i][3])
            if rc_ != api.OK:
                errs.append(rc_)
            inflight[i] = True
        for i in range(P):
            if inflight[i] and ctx[i][0].wait() != api.OK:
                errs.append("wait")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        per_thread = per_handle
        if not errs:
            same = all(bool(torch.equal(c[2][0], o_traj)) and bool(torch.equal(c[2][2], o_nc)) for c in ctx)
            pipelined = {"batches_in_flight": P, "steps": P * per_thread, "value": round(P * per_thread * B / dt, 1),
                         "unit": "solves/s", "results_identical_to_timed_region": same}
        for c in ctx[1:]:
            c[0].close()
        opt.set_stream(torch.cuda.current_stream().cuda_stream)

    # Extra (never `value`): the producer in front of the solve.  Obstacle corner
    # points per knot -> cilqr_build_corridors -> cilqr_solve_batch, everything resident in HBM.
    end_to_end = None
    if world == 1 and args.end_to_end:
        sc_p = scenario.generate(spec, B, seed=args.seed, first_problem=rank * B, workers=workers, obstacle_points=True)
        P_ = sc_p["obstacle_points"].shape[2]
        d_knots = torch.from_numpy(np.ascontiguousarray(sc_p["coarse"][:, :, :3])).to(dev)
        d_pts = torch.from_numpy(sc_p["obstacle_points"]).to(dev)
        d_pcnt = torch.from_numpy(sc_p["obstacle_count"]).to(dev)
        e_cor = torch.zeros((B, K, cmax, 3), dtype=torch.float64, device=dev)
        e_cnt = torch.zeros((B, K), dtype=torch.int32, device=dev)
        ccfg = api.default_corridor_config()
        prob_e = opt.make_problem(B, d_start.data_ptr(), d_coarse.data_ptr(), e_cor.data_ptr(), e_cnt.data_ptr(),
                                  cmax, left.ctypes.data, right.ctypes.data, left.shape[0], right.shape[0],
                                  api.MEM_DEVICE)
        opt.set_profiling(False)
        torch.cuda.synchronize()
        times_c, times_s, failed = [], [], 0
        for it_ in range(1 + max(2, args.steps // 2)):
            t1 = time.perf_counter()
            rc_, nf_ = opt.build_corridors_raw(ccfg, B, K, d_knots.data_ptr(), d_pts.data_ptr(), d_pcnt.data_ptr(), P_,
                                               e_cor.data_ptr(), e_cnt.data_ptr(), cmax, api.MEM_DEVICE)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rc_ != api.OK or opt.solve_raw(prob_e, sol) != api.OK:
                raise api.CilqrError(rc_, "in end-to-end step")
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if it_ > 0:
                times_c.append(t2 - t1)
                times_s.append(t3 - t2)
            failed = nf_
        tc, ts = sum(times_c) / len(times_c), sum(times_s) / len(times_s)
        end_to_end = {"value": round(B / (tc + ts), 1), "unit": "solves/s", "corridor_ms": round(tc * 1e3, 3),
                      "solve_ms": round(ts * 1e3, 3), "steps": len(times_c), "corridors_failed": failed,
                      "mean_obstacle_points": round(float(sc_p["obstacle_count"].mean()), 2),
                      "mean_half_planes": round(float(e_cnt.double().mean().item()), 2),
                      "note": "corridors built by k_build_corridors (sphere-flip construction) instead of the "
                              "generator's simplified ones: a different, larger feasible set, hence another "
                              "iteration count than the timed region"}
        opt.set_profiling(not args.no_profile)

    # sanity: every problem must have terminated with a valid status
    st = o_st.cpu().numpy()
    nc = o_nc.cpu().numpy()
    assert ((st >= 1) & (st <= 5)).all() and (nc >= 1).all(), "unterminated problems in the batch"

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        roof = None
        if not args.no_profile and prof_acc["bwd_ms"] > 0:
            # ALGORITHMIC bytes: a launch over n problems moves n*(N*110+44)*8 B dense.
            # One solve launches k_backward once per lockstep iteration, over 65536 problems at first
            # and over a few stragglers at the end; `achieved` aggregates ALL launches of the timed
            # region (sum of bytes / sum of HIP-event durations = bytes per launch / avg duration, the
            # figure `rocprofv3 --stats` reproduces); `full_batch` is the same ratio over the launches
            # that covered the whole batch (the metric's "batch=65536" case).
            per_problem = (N * api.DENSE_DOUBLES_PER_STEP + api.DENSE_DOUBLES_TERMINAL) * 8.0
            n_act_sum = prof_acc["bwd_steps"] / N
            alg_bytes = n_act_sum * per_problem
            achieved = alg_bytes / (prof_acc["bwd_ms"] * 1e-3) / 1e9
            traffic = None
            tf = {}
            if os.path.exists(args.traffic_file):
                try:
                    with open(args.traffic_file) as f:
                        tf = json.load(f)
                    # PMC-measured HBM bytes per problem-step of the same workload (rocprofv3
                    # FETCH_SIZE x2 + WRITE_SIZE, separate passes), scaled to this run's launches
                    traffic = tf["hbm_bytes_per_problem_step_all_launches"] * prof_acc["bwd_steps"] / prof_acc["bwd_launches"]
                except Exception:
                    traffic = None
            agg = {
                "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": alg_bytes / prof_acc["bwd_launches"],
                "avg_launch_ms": prof_acc["bwd_ms"] / prof_acc["bwd_launches"],
                "launches": prof_acc["bwd_launches"],
                "mean_problems_per_launch": n_act_sum / prof_acc["bwd_launches"],
                "traffic": traffic,
                "note": "every k_backward launch of the timed region, 65536 problems down to a handful; "
                        "launches under ~2000 problems sit on the latency floor of N dependent steps",
            }
            if prof_acc["full_launches"] > 0:
                # the metric's case: one backward pass over the whole batch
                t_full = prof_acc["full_ms"] / prof_acc["full_launches"] * 1e-3
                fb = B * per_problem / t_full / 1e9
                tr_full = None
                try:
                    tr_full = tf["full_batch_launch"]["hbm_bytes_per_problem_step"] * B * N
                except Exception:
                    pass
                roof = {
                    "bound": "hbm", "kernel": "cilqr::k_backward", "launch": f"all {B} problems of the batch",
                    "achieved": round(fb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fb / HBM_PEAK_GBS, 4),
                    "traffic": tr_full,
                    "algorithmic_bytes_per_launch": B * per_problem, "avg_launch_ms": t_full * 1e3,
                    "launches": prof_acc["full_launches"],
                    "real_bytes_gbs": round(tr_full / t_full / 1e9, 1) if tr_full else None,
                    "all_launches": agg,
                }
            else:
                roof = dict(agg, bound="hbm", kernel="cilqr::k_backward", peak=HBM_PEAK_GBS, unit="GB/s")
        cpu = None
        if args.cpu_sample > 0 and world == 1:   # rank 0 at N = 1 only
            from oracle import oracle as orc
            ns = min(args.cpu_sample, B)
            sub = {k: (v[:ns] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
            ocfg = orc.OracleConfig()
            for name, _ in orc.OracleConfig._fields_:
                setattr(ocfg, name, getattr(cfg, name))
            r = orc.solve_batch(sub, ocfg, want_margin=False)
            cpu = {"value": round(ns / r["seconds"], 2), "unit": "solves/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} scenes of rank 0's batch, single thread, g++ -O2 restatement "
                             f"(oracle/cilqr_oracle.cc), {r['seconds']:.1f} s"}
        out = {
            "metric": f"CILQR solves/sec ({N}-step horizon, batch={B} per GPU)",
            "value": round(value, 1), "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE {WORKLOADS.get((args.scene, B), 'custom')}: batch={B}/GPU x {world} GPU, "
                                   f"{N}-step horizon, scene family {args.scene} ({spec.n_pedestrians} pedestrians + "
                                   f"{spec.n_dynamic} moving + {spec.n_static} static vehicles), reference road, "
                                   f"seed {args.seed}",
                       "batch_per_gpu": B, "n_steps": N, "cmax": cmax, "results_gather": "rccl" if use_dist else "none"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "pipelined": pipelined,
            "end_to_end": end_to_end,
            # per-phase HIP-event times of the WARM-UP step(s), which run with events around every phase
            "breakdown_ms_per_step": ({k: round(warm[k] / warm["steps"], 3)
                                       for k in ("quad_ms", "bwd_ms", "ls_ms", "other_ms", "total_ms")}
                                      if warm["steps"] else None),
            "lockstep_iterations_per_step": prof_acc["iters"] / args.steps,
            "mean_cost_rows": float(nc.mean()),
            "status_histogram": np.bincount(st, minlength=6).tolist(),
            "scene_generation_s": round(t_gen, 1),
            "device_bytes": opt.device_bytes(),
        }
    opt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:  # flush what native libraries left in the C stdio buffer (to stderr), then the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
