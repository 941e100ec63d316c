"""Debug helper: H handles x D solves in flight on the same batch; which results differ from the synchronous solve?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cilqr_amd import api, scenario

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda", 0)
sc = scenario.generate("mix11", B, seed=2, workers=16)
cfg = api.default_config(sc["n_steps"])
K, M = sc["n_steps"] + 1, cfg.max_iter
d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("start", "coarse", "corridor", "ccount")}
left, right = np.ascontiguousarray(sc["left"]), np.ascontiguousarray(sc["right"])
opts = [api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=sc["cmax"], max_lane_segments=64) for _ in range(H)]
prob = opts[0].make_problem(B, d["start"].data_ptr(), d["coarse"].data_ptr(), d["corridor"].data_ptr(), d["ccount"].data_ptr(),
                            sc["cmax"], left.ctypes.data, right.ctypes.data, left.shape[0], right.shape[0], api.MEM_DEVICE)
def bufs():
    t = (torch.zeros((B, K, 10), dtype=torch.float64, device=dev), torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
         torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
    return t, api.SolutionBatch(api.MEM_DEVICE, 0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), None, None)
ref, rsol = bufs()
torch.cuda.synchronize()
assert opts[0].solve_raw(prob, rsol) == api.OK
torch.cuda.synchronize()
slots = [[bufs() for _ in range(D)] for _ in range(H)]
fifo = [[] for _ in range(H)]
free = [list(range(D)) for _ in range(H)]
torch.cuda.synchronize()
bad = 0
def collect(hh, step):
    global bad
    assert opts[hh].wait() == api.OK
    k = fifo[hh].pop(0)
    free[hh].append(k)
    t = slots[hh][k][0]
    torch.cuda.synchronize()
    for name, a, b in zip(("traj", "hist", "nc", "st", "ni"), t, ref):
        if name == "hist":
            continue
        if not torch.equal(a, b):
            diff = (a != b)
            while diff.dim() > 1:
                diff = diff.any(-1)
            idx = diff.nonzero().flatten()
            bad += 1
            print(f"handle {hh} slot {k}: {name} differs for {len(idx)} problems, first {idx[:8].tolist()}; ni there {t[4][idx[:8]].tolist()} ref {ref[4][idx[:8]].tolist()} "
                  f"st {t[3][idx[:8]].tolist()} ref {ref[3][idx[:8]].tolist()}")
    for a in t:
        a.zero_()
    torch.cuda.synchronize()
for s_ in range(steps):
    hh = s_ % H
    if len(fifo[hh]) == D:
        collect(hh, s_)
    k = free[hh].pop(0)
    assert opts[hh].submit_raw(prob, slots[hh][k][1]) == api.OK
    fifo[hh].append(k)
for hh in range(H):
    while fifo[hh]:
        collect(hh, -1)
print("mismatching result sets:", bad)
