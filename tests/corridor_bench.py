#!/usr/bin/env python
"""(Under tests/ because it times the CPU oracle beside the kernel; only tests/, smoke() and bench.py's
cpu_baseline leg may use oracle/.)
Times cilqr_build_corridors on device-resident inputs at the bench batch size and
the CPU oracle beside it.
    python tests/corridor_bench.py [batch] [scene-family]
Prints one JSON line: corridors/s on the GPU, on one CPU core (oracle), bytes moved per corridor."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cilqr_amd import api, scenario  # noqa: E402
from oracle import oracle as orc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
family = sys.argv[2] if len(sys.argv) > 2 else "mix11"
base_n = min(B, 2048)
sc = scenario.generate(family, base_n, seed=2, obstacle_points=True, workers=8)
K, P, cmax = sc["n_steps"] + 1, sc["obstacle_points"].shape[2], sc["cmax"]
dev = torch.device("cuda", 0)
rep = (B + base_n - 1) // base_n
knots = torch.from_numpy(np.ascontiguousarray(sc["coarse"][:, :, :3])).to(dev).repeat(rep, 1, 1)[:B].contiguous()
pts = torch.from_numpy(sc["obstacle_points"]).to(dev).repeat(rep, 1, 1, 1)[:B].contiguous()
cnt = torch.from_numpy(sc["obstacle_count"]).to(dev).repeat(rep, 1)[:B].contiguous()
cor = torch.zeros((B, K, cmax, 3), dtype=torch.float64, device=dev)
ccnt = torch.zeros((B, K), dtype=torch.int32, device=dev)
opt = api.BatchIlqrOptimizer(n_steps=sc["n_steps"], batch_capacity=256, cmax=cmax)
opt.set_stream(torch.cuda.current_stream().cuda_stream)
cfg = api.default_corridor_config()
torch.cuda.synchronize()
times = []
for it in range(4):
    t0 = time.perf_counter()
    rc, nf = opt.build_corridors_raw(cfg, B, K, knots.data_ptr(), pts.data_ptr(), cnt.data_ptr(), P, cor.data_ptr(),
                                     ccnt.data_ptr(), cmax, api.MEM_DEVICE)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    assert (rc == api.OK and nf == 0) or os.environ.get("CORRIDOR_BENCH_NO_ASSERT")   # (the stage-ablation variant returns garbage)
t = min(times[1:])
# CPU oracle on the first problems
n_cpu = 0
t0 = time.perf_counter()
for b in range(min(base_n, 40)):
    for k in range(K):
        orc.build_corridor(*sc["coarse"][b, k, :3], sc["obstacle_points"][b, k, :sc["obstacle_count"][b, k]], max_out=cmax)
        n_cpu += 1
t_cpu = time.perf_counter() - t0
used = float(cnt.double().mean().item())
bytes_per = 24 + 4 + used * 16 + float(ccnt.double().mean().item()) * 24 + 4
print(json.dumps({
    "kernel": "cilqr::k_build_corridors", "batch": B, "n_knots": K, "family": family,
    "corridors": B * K, "seconds": round(t, 6), "corridors_per_s": round(B * K / t, 1),
    "batches_per_s_equiv_solves_per_s": round(B / t, 1),
    "mean_obstacle_points": round(used, 2), "mean_half_planes": round(float(ccnt.double().mean().item()), 2),
    "algorithmic_bytes_per_corridor": round(bytes_per, 1), "achieved_GBps": round(B * K * bytes_per / t / 1e9, 2),
    "cpu_oracle": {"corridors_per_s": round(n_cpu / t_cpu, 1), "cores": 1, "sample": f"{n_cpu} corridors (incl. ctypes call overhead)"},
    "all_runs_s": [round(x, 6) for x in times]}))
