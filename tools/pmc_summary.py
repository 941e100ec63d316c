#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd .db (or counter_collection csv)."""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("tables:", tabs)
        return
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {kcol}, counter_name, sum(value), count(*) from {view} group by {kcol}, counter_name").fetchall()
    agg = {}
    for k, cn, v, n in rows:
        agg.setdefault(k.split("(")[0].replace("void ", "").replace("cilqr::", ""), {})[cn] = (v, n)
    # durations of the SAME dispatches (a capture made with --pmc ... --kernel-trace holds them): column dur_ms
    if "kernels" in tabs:
        try:
            for k, tot_ns, n in c.execute("select name, sum(end - start), count(*) from kernels group by name"):
                key = k.split("(")[0].replace("void ", "").replace("cilqr::", "")
                if key in agg:
                    agg[key]["dur_ms"] = (tot_ns / 1e6, n)
        except sqlite3.Error:
            pass
    names = sorted({cn for d in agg.values() for cn in d})
    print(f"{'kernel':28s} {'disp':>6s} " + " ".join(f"{n[:18]:>18s}" for n in names))
    for k, d in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        if not k.startswith("k_"):
            continue
        disp = max(v[1] for v in d.values())
        print(f"{k[:28]:28s} {disp:6d} " + " ".join(f"{d.get(n, (0, 0))[0]:18.4g}" for n in names))


if __name__ == "__main__":
    main()
