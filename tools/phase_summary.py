#!/usr/bin/env python
"""Wall time of each lockstep iteration of one solve from a rocprofv3 kernel trace (rocpd .db):
prints cumulative time by iteration ranges and by kernel inside each range."""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    solve = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    # kernel names as rocprofv3 records them carry the template arguments (k_quadratize<5, true>): cut them off
    names = [r[0].split("(")[0].replace("void ", "").replace("cilqr::", "").split("<")[0] for r in rows]
    starts = [i for i, n in enumerate(names) if n == "k_load_corridor"]
    if solve >= len(starts):
        print(f"(no solve {solve} in this capture: {len(starts)} solves found)")
        return
    lo = starts[solve]
    hi = starts[solve + 1] if solve + 1 < len(starts) else len(rows)
    its, cur = [], []
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        if n == "k_quadratize":
            if cur:
                its.append(cur)
            cur = []
        cur.append((n, r[1], r[2], r[3]))
    its.append(cur)
    if len(its) < 2 or not its[0]:
        print(f"(solve {solve}: no lockstep iteration found among {hi - lo} kernels -- kernel names changed?)")
        return
    t0 = its[0][0][1]
    print(f"prologue {(its[1][0][1] - t0) / 1e6:.2f} ms; {len(its) - 1} iterations; total {(its[-1][-1][2] - t0) / 1e6:.2f} ms")
    bounds = [1, 5, 10, 18, 30, 50, 200]
    for a, b in zip(bounds[:-1], bounds[1:]):
        sel = its[a:min(b, len(its))]
        if not sel:
            break
        wall = (sel[-1][-1][2] - sel[0][0][1]) / 1e6
        busy = {}
        for it in sel:
            for n, s, e, g in it:
                busy[n] = busy.get(n, 0) + (e - s) / 1e6
        tb = sum(busy.values())
        top = sorted(busy.items(), key=lambda kv: -kv[1])[:7]
        print(f"iters {a:3d}-{min(b, len(its)) - 1:3d}: wall {wall:7.2f} ms, kernels {tb:7.2f} ms, n_act(first) {sel[0][0][3]:6d} | "
              + ", ".join(f"{k[2:]}={v:.1f}" for k, v in top))


if __name__ == "__main__":
    main()
