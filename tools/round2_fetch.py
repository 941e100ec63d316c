#!/usr/bin/env python
"""FETCH_SIZE of the round-1 and round-2 launches of k_round_cost, per lockstep iteration: from a rocprofv3
--pmc FETCH_SIZE --kernel-trace capture of ONE sequential solve.  Dispatches of k_round_cost come in pairs per iteration (round 1:
alpha_0, alpha_1 of every active problem; round 2: alpha_2, alpha_3 of those that rejected both).
    python tools/round2_fetch.py <capture dir>"""
import glob
import json
import os
import sqlite3
import sys

path = sys.argv[1]
db = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select dispatch_id, sum(value), max(grid_size_x) from counters_collection where {kcol} like '%k_round_cost%' and "
                 "counter_name = 'FETCH_SIZE' group by dispatch_id order by dispatch_id").fetchall()
dur = {}
try:
    for d, s, e in c.execute("select dispatch_id, start, end from kernels where name like '%k_round_cost%'"):
        dur[d] = (e - s) / 1e3
except sqlite3.Error:
    pass
# FETCH_SIZE: KiB, and on gfx950 64 B are tallied per 128 B request (MI355X_MICROARCH.md): x 2
solves = c.execute(f"select count(distinct dispatch_id) from counters_collection where {kcol} like '%k_load_goals%'").fetchone()[0]
per = len(rows) // max(1, solves)
rows = rows[-per:]            # the last solve of the capture
out = []
for i in range(0, len(rows) - 1, 2):
    (d1, f1, g1), (d2, f2, g2) = rows[i], rows[i + 1]
    out.append({"iteration": i // 2 + 1, "round1_MB": round(f1 * 2 * 1024 / 1e6, 1), "round2_MB": round(f2 * 2 * 1024 / 1e6, 1),
                "round2_over_round1": round(f2 / f1, 3) if f1 else None, "grid1": g1, "grid2": g2,
                "round1_us": round(dur.get(d1, 0.0), 1), "round2_us": round(dur.get(d2, 0.0), 1)})
print(json.dumps({"solves_in_capture": solves, "k_round_cost_dispatches_per_solve": per, "iterations": out[:24]}, indent=1))
