"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, config
defaults, argument errors that need no GPU, the scene generator, sharding + gather (gloo, world 2)."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from cilqr_amd import api, scenario
from cilqr_amd.distributed import shard_range, shard_scene
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(api.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"\b(cilqr_[a-z_0-9]+)\s*\(", hdr)))
    assert set(declared) == set(api.EXPORTS), (declared, api.EXPORTS)
    L = api.lib()
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/cilqr.h but not exported"
    assert L.cilqr_abi_version() == int(re.search(r"CILQR_ABI_VERSION (\d+)", hdr).group(1))
    out = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cilqr_[a-z_0-9]+)", out))
    assert exported == set(declared)     # nothing else leaks through the C-ABI prefix


def _check_library_fingerprint():
    """cilqr_build_id: the first 32 hex digits of the SHA-256 over the library's sources, compiled in by the
    Makefile.  The .so is a prebuilt artefact that travels to the GPU box; this pins it to the sources beside it."""
    import ctypes
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "cilqr_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")), key=os.path.basename)
    files = sorted([os.path.basename(f) for f in files if os.path.basename(f) != "build_id.h"])
    h = hashlib.sha256()   # same order as the Makefile's $(sort ...): the ../../include paths sort first
    for f in ("cilqr.h", os.path.join("cilqr", "dp_planner.hpp")):
        h.update(open(os.path.join(ROOT, "include", f), "rb").read())
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    L = api.lib()
    L.cilqr_build_id.restype = ctypes.c_char_p
    assert L.cilqr_build_id().decode() == h.hexdigest()[:32]


def test_library_was_built_from_the_sources_in_the_tree(built):
    _check_library_fingerprint()


@pytest.mark.gpu
def test_library_on_the_gpu_box_was_built_from_the_sources_in_the_tree():
    """The same pin inside the -m gpu set: the .so that travelled to the GPU box is the one these sources build
    (no `built` fixture: a rebuild on the box would make the test vacuous)."""
    _check_library_fingerprint()


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cilqr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "cilqr_oracle" not in txt and "from oracle" not in txt, f
    assert "oracle" not in open(api.HEADER_PATH).read().lower()


def test_default_config_matches_reference_defaults(built):
    c = api.default_config(50)
    o = orc.default_config(50)
    for name, _ in orc.OracleConfig._fields_:
        assert getattr(c, name) == getattr(o, name), name
    assert (c.n_steps, c.num_of_disc, c.max_iter) == (50, 5, 200)          # planner_config.h:58,63
    assert (c.w_x, c.w_y, c.w_theta, c.w_jerk, c.w_delta_rate) == (0.5, 0.5, 1e-3, 1.0, 1.0)
    assert c.delta_max == 40.0 / 180 * np.pi and c.delta_rate_max == c.delta_max / 3.0
    assert (c.barrier_t, c.barrier_eps, c.safe_margin) == (5.0, 0.01, 0.2)
    assert C.sizeof(api.Config) == 4 * 4 + 27 * 8


def test_option_and_status_constants_match_the_header(built):
    """cilqr_amd/api.py mirrors the #defines of include/cilqr.h by hand: every CILQR_OPT_* must agree."""
    hdr = open(api.HEADER_PATH).read()
    opts = dict(re.findall(r"#define (CILQR_OPT_[A-Z_]+) (\d+)", hdr))
    assert len(opts) >= 9 and len(set(opts.values())) == len(opts)            # distinct codes
    for name, val in opts.items():
        assert getattr(api, name[len("CILQR_"):]) == int(val), name


def test_multi_device_entry_points_reject_bad_arguments_without_a_gpu(built):
    L = api.lib()
    cfg = api.default_config(50)
    h = C.c_void_p()
    dev = np.zeros(2, np.int32)
    assert L.cilqr_multi_create(None, dev.ctypes.data, 2, 8, 16, 64, C.byref(h)) == api.ERR_NULL
    assert L.cilqr_multi_create(C.byref(cfg), None, 2, 8, 16, 64, C.byref(h)) == api.ERR_NULL
    assert L.cilqr_multi_create(C.byref(cfg), dev.ctypes.data, 0, 8, 16, 64, C.byref(h)) == api.ERR_ARG
    assert L.cilqr_multi_create(C.byref(cfg), dev.ctypes.data, 2, 1, 16, 64, C.byref(h)) == api.ERR_ARG   # fewer problems than shards
    assert L.cilqr_multi_solve(None, None, None) == api.ERR_NULL
    assert L.cilqr_multi_destroy(None) == api.ERR_NULL
    assert L.cilqr_multi_device_bytes(None) == 0


def test_pool_entry_points_reject_bad_arguments_without_a_gpu(built):
    L = api.lib()
    cfg = api.default_config(50)
    h = C.c_void_p()
    assert L.cilqr_pool_create(None, 0, 2, 8, 16, 64, C.byref(h)) == api.ERR_NULL
    assert L.cilqr_pool_create(C.byref(cfg), 0, 2, 8, 16, 64, None) == api.ERR_NULL
    assert L.cilqr_pool_create(C.byref(cfg), 0, 0, 8, 16, 64, C.byref(h)) == api.ERR_ARG     # no handles
    assert L.cilqr_pool_create(C.byref(cfg), 0, 17, 8, 16, 64, C.byref(h)) == api.ERR_ARG    # more than 16
    assert L.cilqr_pool_submit(None, None, None) == api.ERR_NULL
    assert L.cilqr_pool_wait(None) == api.ERR_NULL
    assert L.cilqr_pool_destroy(None) == api.ERR_NULL
    assert L.cilqr_pool_depth(None) == 0 and L.cilqr_pool_device_bytes(None) == 0


def test_errors_without_gpu(built):
    L = api.lib()
    assert L.cilqr_default_config(None, 50) == api.ERR_NULL
    assert L.cilqr_destroy(None) == api.ERR_NULL
    assert L.cilqr_solve_batch(None, None, None) == api.ERR_NULL
    assert L.cilqr_stage_init_guess(None) == api.ERR_NULL
    h = C.c_void_p()
    cfg = api.default_config(50)
    assert L.cilqr_create(C.byref(cfg), 0, 0, 16, 64, C.byref(h)) == api.ERR_ARG      # zero capacity
    bad = api.default_config(50, num_of_disc=99)
    assert L.cilqr_create(C.byref(bad), 0, 8, 16, 64, C.byref(h)) == api.ERR_ARG
    assert L.cilqr_create(C.byref(cfg), 0, 8, 16, 64, None) == api.ERR_NULL
    assert b"coarse_traj" in L.cilqr_error_string(api.ERR_KNOTS)
    assert b"constraints" in L.cilqr_error_string(api.ERR_CONSTRAINTS)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        # the product path must fail loudly without a GPU: no CPU fallback
        with pytest.raises(api.CilqrError) as e:
            api.BatchIlqrOptimizer(n_steps=50, batch_capacity=4)
        assert e.value.code == api.ERR_DEVICE


def test_scene_generator_is_deterministic_and_well_formed():
    a = scenario.generate("mix11", 12, seed=5, chunk=5)
    b = scenario.generate("mix11", 4, seed=5, first_problem=8)
    for k in ("start", "coarse", "corridor", "ccount"):
        assert np.array_equal(a[k][8:12], b[k])
    K = a["n_steps"] + 1
    assert a["coarse"].shape == (12, K, 6) and a["corridor"].shape == (12, K, a["cmax"], 3)
    assert a["ccount"].min() >= 4 and a["ccount"].max() <= a["cmax"]       # the 4 box planes are always live
    assert a["left"].shape[1] == 7 and a["right"].shape[1] == 7
    # every coarse knot lies strictly inside its own (unshrunk) corridor and between the lanes
    for p in range(12):
        for i in range(K):
            n = a["ccount"][p, i]
            pl = a["corridor"][p, i, :n]
            x, y = a["coarse"][p, i, :2]
            assert np.all(pl[:, 0] * x + pl[:, 1] * y < pl[:, 2])
    # lane rows: (a,b) is the segment direction rotated by -90 degrees, c = a*sx + b*sy (corridor.cc:322-331)
    for tab in (a["left"], a["right"]):
        d = tab[:, 5:7] - tab[:, 3:5]
        assert np.allclose(tab[:, 0], d[:, 1]) and np.allclose(tab[:, 1], -d[:, 0])
        assert np.allclose(tab[:, 2], tab[:, 0] * tab[:, 3] + tab[:, 1] * tab[:, 4])
        assert np.all(np.hypot(d[:, 0], d[:, 1]) >= 5.0 - 1e-9)
    # consecutive segments share their end points exactly (ties in the nearest-segment scan)
    assert np.array_equal(a["right"][:-1, 5:7], a["right"][1:, 3:5])
    assert np.array_equal(a["left"][1:, 5:7], a["left"][:-1, 3:5])
    road = scenario.build_road()
    assert road.length == pytest.approx(30 + 10 * np.pi / 2 + 10 + 5 * np.pi + 36 + 12 * np.pi + 50, abs=0.11)


def test_shard_range_partitions():
    for total, world in [(65536, 8), (10, 3), (7, 8), (1, 1)]:
        got = [shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
    sc = scenario.generate("ped6", 6, seed=1)
    s1 = shard_scene(sc, 1, 2)
    assert np.array_equal(s1["coarse"], sc["coarse"][3:]) and s1["left"] is sc["left"]


_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from cilqr_amd import scenario
    from cilqr_amd.distributed import shard_scene, gather_results
    from oracle import oracle as orc
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenario.generate("ped6", 8, seed=21)
    mine = shard_scene(sc, rank, world)
    # stand-in for the per-rank GPU solve: the CPU oracle (tests only)
    r = orc.solve_batch(mine, want_margin=False)
    out = gather_results(torch.from_numpy(r["traj"]), torch.from_numpy(r["cost_hist"]),
                         torch.from_numpy(r["n_cost"]), torch.from_numpy(r["status"]), dst=0)
    if rank == 0:
        full = orc.solve_batch(sc, want_margin=False)
        H = int(full["n_cost"].max())
        assert np.array_equal(out["traj"].numpy(), full["traj"])            # rank order == problem order
        assert np.array_equal(out["cost_hist"].numpy(), full["cost_hist"][:, :H])
        assert np.array_equal(out["n_cost"].numpy(), full["n_cost"])
        assert np.array_equal(out["status"].numpy(), full["status"])
        rag = None
    rag = gather_results(torch.from_numpy(r["traj"]), torch.from_numpy(r["cost_hist"]),
                         torch.from_numpy(r["n_cost"]), torch.from_numpy(r["status"]), dst=0, densify=False)
    if rank == 0:
        rows = np.concatenate([full["cost_hist"][b, :full["n_cost"][b]] for b in range(8)])
        assert np.array_equal(rag["hist_rows"].numpy(), rows) and np.array_equal(rag["traj"].numpy(), full["traj"])
    # only the 8 independent columns of a trajectory point travel; time and kappa are rebuilt on rank 0
    slim = gather_results(torch.from_numpy(r["traj"]), torch.from_numpy(r["cost_hist"]),
                          torch.from_numpy(r["n_cost"]), torch.from_numpy(r["status"]), dst=0, densify=False,
                          derive=(0.1, 1.0))
    if rank == 0:
        t = slim["traj"].numpy()
        assert np.array_equal(t[:, :, [1, 2, 3, 4, 5, 6, 8, 9]], full["traj"][:, :, [1, 2, 3, 4, 5, 6, 8, 9]])
        assert np.allclose(t[:, :, 0], full["traj"][:, :, 0], rtol=0, atol=1e-15)
        assert np.allclose(t[:, :, 7], full["traj"][:, :, 7], rtol=1e-15, atol=1e-15)
        print("GATHER_OK")
    else:
        assert out is None and slim is None
    dist.destroy_process_group()
""")


def test_sharded_solve_and_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def _one_json_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks itself (dry mode: gloo, nothing
    solved) and report them; a launcher-provided environment keeps working; a mismatch is an error, not a 1-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["ranks_reporting"] == 2 and rec["gather_in_rank_order"] is True
    assert rec["spawned_by_bench"] is True and rec["dry_run"] is True and rec["value"] is None
    # the driver's own launcher
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29531", bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["ranks_reporting"] == 2 and rec["spawned_by_bench"] is False
    # --gpus disagrees with the launcher: refuse
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_bench_rehearses_configs3_at_eight_ranks_without_a_gpu():
    """BASELINE configs[3] (8 x 65536, results gathered to rank 0) as far as a box without GPUs allows: `bench.py --gpus 8
    --dry-run` starts EIGHT ranks -- by itself and under the driver's launcher -- which rendezvous over gloo and push made-up
    results of several steps through the SAME gather thread the timed region uses (cilqr_amd/distributed.py: GatherThread):
    rank r's block at offset r * B, steps in order, ragged Cost rows, time / kappa rebuilt on the root; ONE JSON line with
    n_gpus 8.  The one-process form (`--multi`: a thread per GPU, peer copies instead of a collective) prints the same line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    runs = {
        "spawned": [sys.executable, bench, "--gpus", "8", "--dry-run", "--steps", "5"],
        "launcher": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                     "--master-port", "29541", bench, "--gpus", "8", "--steps", "5", "--warmup", "2", "--dry-run"],
        "multi": [sys.executable, bench, "--gpus", "8", "--multi", "--dry-run", "--steps", "5"],
    }
    for name, cmd in runs.items():
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        rec = _one_json_line(r.stdout)
        assert rec["n_gpus"] == 8 and rec["ranks_reporting"] == 8 and rec["dry_run"] is True and rec["value"] is None, (name, rec)
        assert rec["gathers"] == 5 and rec["gather_in_rank_order"] is True and rec["gathers_in_step_order"] is True, (name, rec)
        assert rec["ragged_history_rows_ok"] is True and rec["gather_checks_ok"] is True, (name, rec)
        assert rec["mode"] == ("multi" if name == "multi" else "ranks") and rec["spawned_by_bench"] is (name == "spawned")
        # the host budget of the real run: eight ranks generate their scenes at the same time, so the
        # generator's worker processes are capped by the cores the box grants -- its CPU quota where there is one -- per rank
        cores = min(rec["logical_cpus"], int(rec["cpu_quota_cores"])) if rec["cpu_quota_cores"] else rec["logical_cpus"]
        assert 1 <= rec["scene_workers_per_rank"] <= max(1, cores // 8), (name, rec)
    # mismatches are errors, never a smaller run: the launcher started 2 ranks for --gpus 8; --multi under a launcher
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--multi", "--dry-run"], env=dict(env, WORLD_SIZE="8", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--multi is ONE process" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_distributed_paths_on_one_gpu():
    """What a 1-GPU box can run of the N > 1 paths, with real solves: (a) CILQR_BENCH_FORCE_DIST=1 -- the RCCL process group, the
    rank census, the per-step gather through the gather thread and cilqr_gather_results, all with one rank; (b) --multi with
    both "GPUs" mapped onto device 0: two threads, two pools, peer-copy gather.  Each prints one line; (b) reports n_gpus 2."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--batch", "2048", "--steps", "6", "--warmup", "2", "--no-traffic", "--no-latency", "--cpu-sample", "0", "--cpu-configs", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench] + common, env=dict(env, CILQR_BENCH_FORCE_DIST="1", MASTER_PORT="29551"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["results_gather"] == "rccl" and rec["config"]["rccl_ranks"] == 1
    assert rec["results_gather_thread"]["gathers"] == 6 and rec["c_abi_gather"]["ok"] is True
    assert rec["c_abi_gather"]["identical_to_torch_gather"] is True and rec["c_abi_gather"]["rank0_block_identical_to_local"] is True
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--multi"] + common, env=dict(env, CILQR_BENCH_MULTI_DEVICES="0,0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["config"]["processes"] == 1 and rec["config"]["results_gather"].startswith("peer copies")
    assert rec["results_gather_thread"]["gathers"] == 6 and rec["value"] > 0
    assert abs(rec["value"] - 2 * 2048 * 1e3 / rec["ms_per_step"]) <= 1e-3 * rec["value"]


@pytest.mark.gpu
def test_host_cpu_budget_of_a_rank_and_phase_ranges():
    """eight ranks share the 16 cores a GPU box grants, so a rank may not keep more than two of them busy.
    Submitted solves nap in their host waits (solver.hip: wait_event): the bench line's `host` record -- getrusage around the
    timed region -- must stay under 2 cores for one rank (4.4 with the spinning waits of round 4, which CILQR_HOST_WAIT=spin
    brings back: checked to be the larger figure, so the measurement can see the difference).  Also: CILQR_ROCTX=1 opens the
    roctx ranges of the solve's phases without changing anything else (same status histogram)."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "12", "--warmup", "3", "--no-traffic", "--no-latency", "--cpu-sample", "0", "--cpu-configs", "0", "--no-profile"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    recs = {}
    for name, extra in (("nap", {"CILQR_ROCTX": "1"}), ("spin", {"CILQR_HOST_WAIT": "spin"})):
        r = subprocess.run([sys.executable, bench] + common, env=dict(env, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        recs[name] = _one_json_line(r.stdout)
    nap, spin = recs["nap"]["host"], recs["spin"]["host"]
    assert nap["native_threads_this_rank"] > 4 and nap["scene_workers_per_rank"] >= 1
    # relative, not box-specific: spinning waits cost clearly more host CPU than napping ones -- at least half as
    # much again and half a core; the absolute budget (under two cores per rank) only where a cgroup quota says what a core is
    assert spin["cores_busy_all_ranks"] > 1.5 * nap["cores_busy_all_ranks"] and \
        spin["cores_busy_all_ranks"] > nap["cores_busy_all_ranks"] + 0.5, (nap, spin)
    if nap["cpu_quota_cores"]:
        assert nap["cores_busy_all_ranks"] < 2.0, nap
    assert recs["nap"]["status_histogram"] == recs["spin"]["status_histogram"]
    assert recs["nap"]["value"] > 0.9 * recs["spin"]["value"]


TYPES = ("stand-ins", "reference headers")


def build_cpp_test(name, tmp_path, types="stand-ins"):
    """One of the C++ programs under tests/cpp that drive the adapters of include/cilqr/*.hpp through the C-ABI.
    types = "stand-ins": compiled here against tests/cpp/reference_types.hpp's minimal stand-ins for the reference's types;
    types = "reference headers": the same program compiled against the REFERENCE'S OWN headers (TrajectoryPoint, StartState,
    DiscretizedTrajectory, Vec2d, LineSegment2d, Polygon2d, IlqrConfig, Weights, CorridorConfig, PlannerConfig,
    VehicleParam) and linked against its own objects -- oracle/Makefile target ref_typed_tests builds it into oracle/_ref/
    where the reference tree exists (this container); on the GPU box the prebuilt binary that travelled is used."""
    if types == "stand-ins":
        exe = tmp_path / name
        cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
               os.path.join(ROOT, "tests", "cpp", name + ".cc"), "-o", str(exe),
               "-L" + os.path.dirname(api.LIB_PATH), "-lcilqr_hip", "-Wl,-rpath," + os.path.dirname(api.LIB_PATH),
               "-Wl,-rpath-link,/opt/rocm/lib"]
        subprocess.check_call(cmd)
        return exe
    import pathlib
    exe = pathlib.Path(ROOT) / "oracle" / "_ref" / (name + "_reftypes")
    if os.path.isdir("/root/reference/algorithm"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref/libcilqr_ref.so", f"_ref/{name}_reftypes"])
    if not exe.exists():
        pytest.skip(f"{exe} was not built (no reference tree here and nothing shipped)")
    if not os.access(exe, os.X_OK):
        os.chmod(exe, 0o755)
    return exe


def build_adapter_test(tmp_path, types="stand-ins"):
    return build_cpp_test("adapter_test", tmp_path, types)


def build_corridor_adapter_test(tmp_path, types="stand-ins"):
    return build_cpp_test("corridor_adapter_test", tmp_path, types)


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields_on_the_gpu():
    """`python bench.py` (small batch, a few steps) on the GPU box: ONE JSON line with the fields the driver reads -- metric,
    value, n_gpus, steps, ms_per_step, dtype, config.workload -- the roofline object (bound / achieved / peak / unit / frac /
    traffic) and the cpu_baseline object (value / unit / cores / kind / sample); the timed steps went through the pool of two
    handles and every solve in flight returned the same bits."""
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--batch", "2048", "--steps", "7", "--warmup", "2", "--no-traffic", "--no-latency",
                        "--cpu-sample", "64", "--cpu-configs", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["unit"] == "solves/s" and rec["value"] > 0 and rec["n_gpus"] == 1 and rec["steps"] == 7 and rec["warmup"] == 2
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["dtype"] == "f64" and rec["vs_baseline"] is None
    assert abs(rec["value"] - 2048 * 1e3 / rec["ms_per_step"]) <= 1e-3 * rec["value"]
    assert "workload" in rec["config"] and rec["config"]["handles"] == 2 and rec["config"]["batches_in_flight"] == 4
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0.0 < roof["frac"] <= 1.0 and roof["achieved"] > 0 and "traffic" in roof
    # the kernel's own numbers are flat keys of the line (the driver's parser keeps one level): one solve alone on the GPU at
    # the top, the whole batch in one launch, the timed region with four solves in flight, and the bytes behind them
    for key in ("frac_algorithmic", "frac_full_batch", "frac_contended", "bytes_per_problem_step", "bytes_per_problem_step_minimum",
                "avg_launch_ms", "avg_launch_ms_contended", "launches", "launches_contended", "full_batch_avg_launch_ms"):
        assert isinstance(roof[key], (int, float)) and roof[key] > 0, key
    assert roof["frac_contended"] <= roof["frac"] * 1.25 and roof["bytes_per_problem_step_minimum"] == 400
    # achieved and frac are quoted on the SAME bytes (the HBM bytes really moved); the dense SURVEY 8(d) figure stands beside them
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) <= 2e-4 and "achieved_basis" in roof
    assert abs(roof["achieved_algorithmic"] / roof["peak"] - roof["frac_algorithmic"]) <= 2e-4
    assert abs(roof["achieved_algorithmic"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * roof["achieved_algorithmic"]
    assert roof["launches_contended"] >= 7 * roof["launches"] * 0.5
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["unit"] == "solves/s" and "sample" in cpu
    assert rec["results_identical_across_solves_in_flight"] is True
    assert rec["one_handle"]["value"] > 0 and rec["single_batch"]["value"] > 0
    # the two figures of a caller whose arrays are not in HBM yet: host arrays through the pool, obstacle points -> corridors -> solve
    assert rec["pcie_inclusive"]["value"] > 0 and rec["pcie_inclusive"]["identical_to_device_resident"] is True
    assert rec["end_to_end"]["value"] > 0 and rec["end_to_end"]["identical_across_steps_and_to_the_sequential_call"] is True
    assert rec["end_to_end"]["corridors_failed"] == 0


def test_header_is_plain_c99_and_the_pool_calls_link(built, tmp_path):
    """include/cilqr.h compiled as C99 (-pedantic -Werror) by gcc, linked against the C-ABI library alone; the program
    (tests/cpp/pool_c99.c) walks the cilqr_pool_* argument checks, which need no GPU."""
    exe = tmp_path / "pool_c99"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "pool_c99.c"), "-o", str(exe),
                           "-L" + os.path.dirname(api.LIB_PATH), "-lcilqr_hip", "-Wl,-rpath," + os.path.dirname(api.LIB_PATH),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "pool_c99 ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("types", TYPES)
def test_corridor_adapter_compiles_as_cxx14_against_the_c_abi(built, tmp_path, types):
    """include/cilqr/corridor.hpp (the planning::Corridor call surface) builds with C++14 / g++ and
    links against the C-ABI library only -- with stand-ins for the reference's types and with the reference's own
    headers (corridor.h:27-44's argument types: DiscretizedTrajectory, Vec2d, LineSegment2d, CorridorConfig)."""
    assert build_corridor_adapter_test(tmp_path, types).exists()


@pytest.mark.parametrize("types", TYPES)
def test_cpp_adapter_compiles_as_cxx14_against_the_c_abi(built, tmp_path, types):
    """The drop-in header (include/cilqr/ilqr_optimizer.hpp) must build with the reference's own
    toolchain settings (C++14, g++, no HIP headers) and link against the C-ABI library only -- with stand-ins and with
    the reference's own TrajectoryPoint / DiscretizedTrajectory / LineSegment2d / IlqrConfig / VehicleParam, i.e. exactly
    what trajectory_planner.cpp:26,80-97 hands to IlqrOptimizer (compiled here, not asserted)."""
    exe = build_adapter_test(tmp_path, types)
    assert exe.exists()
    hdr = open(os.path.join(ROOT, "include", "cilqr.h")).read()
    includes = re.findall(r"#include\s+[<\"]([^>\"]+)", hdr)
    assert includes == ["stdint.h"]             # plain C: no HIP, C++ or torch headers at the boundary


def test_peer_gather_pads_history_rows_and_never_hangs_on_a_failing_rank():
    """cilqr_amd/distributed.py: PeerGather (results of several GPUs driven by ONE process, bench.py --multi), on CPU tensors.
    (a) the Cost rows beyond a job's longest history are zero on the root, whatever an earlier job left there (every other host
    path pads them with zeros); (b) a rank whose gather raises breaks the barrier: the other rank gets BrokenBarrierError at
    once instead of waiting for ever."""
    import threading
    import torch
    from cilqr_amd.distributed import GatherThread, PeerGather
    world, B, K, M = 2, 4, 3, 6
    peer = PeerGather(world, B, K, M, root_device=None, derive=(0.1, 1.0), timeout_s=20.0)
    fns = [peer.gather_fn(r) for r in range(world)]

    def job(rank, n_rows, fill, out):
        traj = torch.full((B, K, 10), float(fill), dtype=torch.float64)
        hist = torch.full((B, M + 1, 5), float(fill), dtype=torch.float64)
        nc = torch.full((B,), n_rows, dtype=torch.int32)
        st = torch.full((B,), 2, dtype=torch.int32)
        out[rank] = fns[rank](traj, hist, nc, st)

    for n_rows, fill in ((6, 7.0), (2, 9.0)):          # a long job, then a short one into the same root tensors
        res = [None, None]
        ts = [threading.Thread(target=job, args=(r, n_rows, fill, res)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(30.0) for t in ts]
        assert not any(t.is_alive() for t in ts)
        g = res[0]
        assert torch.all(g["cost_hist"][:, :n_rows] == fill) and torch.all(g["cost_hist"][:, n_rows:] == 0.0), n_rows
        assert torch.all(g["n_cost"] == n_rows) and res[1] is None
    # (b) rank 1 fails before it reaches the barrier: rank 0 must come back with an error, quickly
    errors = {}

    def good():
        try:
            job(0, 3, 1.0, [None, None])
        except BaseException as e:   # noqa: BLE001
            errors[0] = e

    def bad():
        try:
            fns[1](torch.zeros((B, K, 10), dtype=torch.float64), torch.zeros((B, M + 1, 5), dtype=torch.float64),
                   None, torch.zeros(B, dtype=torch.int32))          # n_cost missing: raises inside the gather
        except BaseException as e:   # noqa: BLE001
            errors[1] = e

    ts = [threading.Thread(target=good), threading.Thread(target=bad)]
    [t.start() for t in ts]
    [t.join(15.0) for t in ts]
    assert not any(t.is_alive() for t in ts), "a rank is still waiting in the barrier"
    assert isinstance(errors.get(0), threading.BrokenBarrierError) and 1 in errors, errors
    # the same through GatherThread: drain() re-raises, on_error is called once
    seen = []
    gt = GatherThread(device=None, gather_fn=lambda *a: (_ for _ in ()).throw(RuntimeError("boom")), on_error=seen.append)
    gt.put("tag", None, None, None, None)
    gt.put("tag2", None, None, None, None)
    with pytest.raises(RuntimeError):
        gt.drain()
    gt.close()
    assert len(seen) == 1 and gt.count == 2
