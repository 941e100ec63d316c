"""Chooses the scenes of tests/golden/zero_tolerance_scenes.json (test_exit_paths, zero-tolerance case).

With both cost tolerances at 0 the solver iterates into the rounding-noise plateau, where most problems are not
reproducible by the oracle itself (accept / reject decisions hang on the last bits of a cost difference).  This script
runs the ORACLE on a pool of generated scenes and keeps 32 whose whole solve survives 8 re-runs at 4e-16 input noise
with every decision at least 1e-9 (relative) from its threshold -- problems on which whole-solve parity means
something -- plus 16 arbitrary ones.  (No stable problem of the pool ends with lambda > 1e11: that exit is reached
through noisy rejections by nature and is held to the per-step replay instead.)

    cd tests && python golden/make_zero_tolerance_set.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cilqr_amd import api, scenario  # noqa: E402
from parity_util import oracle_cfg_from, oracle_reference  # noqa: E402

over = dict(rel_cost_tol=0.0, abs_cost_tol=0.0, max_iter=60)
pool_seed, n_pool = 151, 384
sc = scenario.generate("ped6", n_pool, seed=pool_seed)
cfg = api.default_config(sc["n_steps"], **over)
ref = oracle_reference(sc, oracle_cfg_from(cfg))
good = ref["stable"] & (ref["min_margin"] >= 1e-9)
st = ref["status"]
idx_good = np.nonzero(good)[0]
pri = [i for i in idx_good if st[i] in (3, 4)]
rest = [i for i in idx_good if st[i] not in (3, 4)]
chosen = (pri + rest)[:32]
others = [i for i in range(n_pool) if i not in set(chosen)][:16]
sel = sorted(chosen + others)
print("pool: stable share", good.mean(), "status of the stable ones", np.bincount(st[good], minlength=7))
json.dump({"family": "ped6", "pool_seed": pool_seed, "pool": n_pool, "config": over, "indices": [int(i) for i in sel],
           "note": "48 of 384 generated scenes: 32 whose zero-tolerance solve the oracle itself reproduces under 4e-16 "
                   "input noise with every decision at least 1e-9 (relative) from its threshold, plus 16 arbitrary ones"},
          open(os.path.join(HERE, "zero_tolerance_scenes.json"), "w"), indent=1)
