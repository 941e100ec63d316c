"""How the CPU restatement scales with host threads on this box (oracle_solve_batch_threads); also prints the cgroup CPU quota.
   python tools/cpu_scaling.py [scenes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilqr_amd import scenario
from oracle import oracle as orc
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenario.generate("mix11", n, seed=2, workers=8)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    m = min(n, max(64, th * 16))
    sub = {k: (v[:m] if hasattr(v, "shape") and v.shape[:1] == (n,) else v) for k, v in sc.items()}
    r = orc.solve_batch_threads(sub, threads=th)
    print(f"threads {th:4d}: {m / r['seconds']:9.1f} solves/s ({m} scenes, {r['seconds']:.2f} s)", flush=True)
