#!/usr/bin/env python
"""Does "this problem needs round 2 of the line search" persist from one iteration to the next?  
CPU only: the oracle's alpha traces (accepted step-size index per iteration; -1 = all eleven rejected) on N mix11 scenes.
A problem needs round 2 in an iteration when it rejects alpha_0 and alpha_1 (index >= 2, or -1).  If P(needs it again | needed
it) were high, ordering the survivors by it when they are re-packed would make the pending list of round 2 dense in slot space.
usage: linesearch_round2_persistence.py [N] [out.json]"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilqr_amd import scenario  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sc = scenario.generate("mix11", N, seed=2, workers=8)
cfg = orc.default_config(sc["n_steps"])
chunks = [slice(i, min(N, i + 128)) for i in range(0, N, 128)]


def run(sl):
    sub = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (N,) else v) for k, v in sc.items()}
    return orc.solve_batch(sub, cfg, want_margin=False, want_trace=True)["alpha_trace"]


with ThreadPoolExecutor(8) as ex:
    at = np.concatenate(list(ex.map(run, chunks)))       # [N, max_iter] int8: -3 = not run, -2 = left before the search
ran = at >= -1
need = ran & ((at >= 2) | (at == -1))
both = ran[:, :-1] & ran[:, 1:]
n11 = int((need[:, :-1] & need[:, 1:] & both).sum())
n10 = int((need[:, :-1] & ~need[:, 1:] & both).sum())
n01 = int((~need[:, :-1] & need[:, 1:] & both).sum())
n00 = int((~need[:, :-1] & ~need[:, 1:] & both).sum())
by_iter = []
for i in range(0, 16):
    a = ran[:, i].sum()
    by_iter.append({"iteration": i + 1, "active": int(a), "need_round2_share": round(float(need[:, i].sum() / max(1, a)), 3)})
out = {"scenes": N, "problem_iterations": int(ran.sum()), "need_round2_share": round(float(need.sum() / ran.sum()), 4),
       "P_need_again_given_needed": round(n11 / max(1, n11 + n10), 4), "P_need_given_not_needed": round(n01 / max(1, n01 + n00), 4),
       "pairs": {"needed_then_needed": n11, "needed_then_not": n10, "not_then_needed": n01, "not_then_not": n00},
       "by_iteration": by_iter,
       "reading": "the kill criterion of the verdict: order survivors by it only if P(needs it again | needed it) > 0.6"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
