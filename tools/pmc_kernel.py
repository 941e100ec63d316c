#!/usr/bin/env python
"""Per-dispatch values of the counters in a rocprofv3 --pmc capture for one kernel.

    python tools/pmc_kernel.py <dir> k_backward
prints dispatch index, grid size, counter values; then the sums.
"""
import glob
import os
import sqlite3
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    gcol = "grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None)
    q = f"select dispatch_id, counter_name, sum(value){', max(' + gcol + ')' if gcol else ''} from counters_collection " \
        f"where {kcol} like ? group by dispatch_id, counter_name order by dispatch_id"
    rows = c.execute(q, (f"%{kern}%",)).fetchall()
    by = {}
    for r in rows:
        by.setdefault(r[0], {})[r[1]] = r[2]
        if gcol:
            by[r[0]]["grid"] = r[3]
    names = sorted({k for d in by.values() for k in d if k != "grid"})
    print("dispatch grid " + " ".join(names))
    tot = {n: 0.0 for n in names}
    for i, (d, v) in enumerate(sorted(by.items())):
        for n in names:
            tot[n] += v.get(n, 0.0)
        if i < 6 or i % 20 == 0:
            print(d, v.get("grid"), " ".join(f"{v.get(n, 0.0):.6g}" for n in names))
    print("SUM", len(by), " ".join(f"{tot[n]:.6g}" for n in names))
    # one k_load_goals dispatch per solve: how many solves the capture holds
    n = c.execute(f"select count(distinct dispatch_id) from counters_collection where {kcol} like '%k_load_goals%'").fetchone()[0]
    print("SOLVES", n)


if __name__ == "__main__":
    main()
