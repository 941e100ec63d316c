#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace capture (rocpd .db or *_kernel_trace.csv): per-kernel
count / total / avg / min / max, and optionally the kernel timeline of chosen lockstep iterations.

    python tools/prof_summary.py <dir-or-file> [--iters 1,2,5,20] [--solve 1]
"""
import argparse
import csv
import glob
import os
import sqlite3
import sys


def load(path):
    if os.path.isdir(path):
        dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
        csvs = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        path = (dbs or csvs or [None])[0]
        if path is None:
            sys.exit("no .db / kernel_trace.csv found")
    rows = []
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, start, end, gx, gy, gz in c.execute(
                "select name, start, end, grid_x, grid_y, grid_z from kernels order by start"):
            rows.append((name, int(start), int(end), gx, gy, gz))
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                             int(r.get("Grid_Size_X", 0)), int(r.get("Grid_Size_Y", 0)), int(r.get("Grid_Size_Z", 0))))
        rows.sort(key=lambda r: r[1])
    return rows


def short(name):
    return name.split("(")[0].replace("void ", "").replace("cilqr::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--iters", default="")
    ap.add_argument("--solve", type=int, default=1, help="which solve (0-based) to print the timeline of")
    ap.add_argument("--bench-json", default="", help="the JSON line bench.py printed in this capture: adds the backward launches of "
                    "its TIMED REGION alone (the figure bench.py's roofline.avg_launch_ms must agree with)")
    a = ap.parse_args()
    rows = load(a.path)
    stats = {}
    for name, s, e, *_ in rows:
        st = stats.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    tot = sum(v[1] for v in stats.values())
    print(f"{'kernel':44s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{k[:44]:44s} {v[0]:7d} {v[1] / 1e6:10.3f} {v[1] / v[0] / 1e3:9.1f} {v[2] / 1e3:9.1f} {v[3] / 1e3:9.1f} {100 * v[1] / tot:6.2f}")
    print(f"{'TOTAL':44s} {sum(v[0] for v in stats.values()):7d} {tot / 1e6:10.3f}")
    # the roofline kernel by launch size (grid x = problems rounded up to 64); the thresholds between the three mappings as the
    # LIBRARY holds them (bench.py writes cilqr_get_option's answers into config.backward_thresholds)
    thr_team = thr_wave = "?"
    if a.bench_json:
        import json as _json
        try:
            thr = _json.loads(open(a.bench_json).read().strip().splitlines()[-1])["config"]["backward_thresholds"]
            thr_team, thr_wave = thr["team"], thr["wave"]
        except Exception:   # noqa: BLE001
            pass
    sizes = {}
    for name, s_, e_, gx, *_ in rows:
        if short(name) not in ("k_backward", "k_backward_team", "k_backward_wave"):
            continue
        if short(name) == "k_backward_team":
            gx = gx // 8          # eight lanes per problem
        if short(name) == "k_backward_wave":
            gx = gx // 64         # a wavefront per problem
        b = ">=65536" if gx >= 65536 else (">=8192" if gx >= 8192 else (f">{thr_team} (one lane)" if short(name) == "k_backward" else
                                                                        ("eight lanes" if short(name) == "k_backward_team" else "a wavefront")))
        st = sizes.setdefault(b, [0, 0, 0])
        st[0] += 1; st[1] += e_ - s_; st[2] += gx
    if sizes:
        print("\nbackward kernels by launch size (problems per launch; k_backward: one lane per problem; k_backward_team,\n"
              f"eight lanes, launches of at most CILQR_OPT_TEAM_THRESHOLD = {thr_team} problems; k_backward_wave, a wavefront,\n"
              f"at most CILQR_OPT_WAVE_THRESHOLD = {thr_wave}; with N = 50 as in the default bench):")
        for b in (">=65536", ">=8192", f">{thr_team} (one lane)", "eight lanes", "a wavefront"):
            if b in sizes:
                n, t, g = sizes[b]
                print(f"  {b:8s} launches {n:5d}  avg {t / n / 1e3:8.1f} us  avg problems {g / n:9.0f}  "
                      f"algorithmic GB/s {(g / n) * (50 * 110 + 44) * 8 / (t / n):9.1f}")
    if a.bench_json:
        import json
        rec = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
        steps, warm, roof = rec["steps"], rec["warmup"], rec["roofline"]
        new_layout = "launches_contended" in roof      # round 4 on: top-level keys = ONE solve alone, *_contended = the timed region
        per_solve = roof["launches"] if new_layout else roof["launches"] // steps   # backward launches of one solve (lockstep iterations)
        # the two calibration solves are synchronous calls, the warm-up and timed steps submitted ones: since the tail threshold
        # depends on the kind of call (CILQR_OPT_TAIL_THRESHOLD: one value for cilqr_solve_batch, one for submitted solves) they differ in the number of lockstep iterations, i.e. of backward launches
        per_sync = per_solve
        if new_layout:
            per_solve = roof["launches_contended"] // steps
        skip = 2 * per_sync + warm * per_solve          # the two calibration solves and the warm-up steps come first
        bw = [(s_, e_) for name, s_, e_, *_ in rows if short(name) in ("k_backward", "k_backward_team", "k_backward_wave")]
        win = bw[skip:skip + steps * per_solve]         # rows are sorted by start time; fences separate the regions
        if len(win) == steps * per_solve:
            avg = sum(e_ - s_ for s_, e_ in win) / len(win) / 1e6
            hip_ms = roof["avg_launch_ms_contended"] if new_layout else roof["avg_launch_ms"]
            print(f"\nbackward launches of bench.py's timed region alone ({steps} steps x {per_solve} launches, after {2 + warm} earlier solves; "
                  f"the rows above also hold the calibration, warm-up, one-handle and sequential legs of the same process):\n"
                  f"  launches {len(win)}  avg {avg * 1e3:.1f} us   -- bench.py roofline.avg_launch_ms_contended (HIP events, same run): {hip_ms * 1e3:.1f} us")
            if new_layout:
                cal = bw[per_sync:2 * per_sync]          # the calibration solve: second solve of the process, alone on the GPU
                avg1 = sum(e_ - s_ for s_, e_ in cal) / len(cal) / 1e6
                print(f"backward launches of the calibration solve (one solve alone on the GPU, {len(cal)} launches):\n"
                      f"  avg {avg1 * 1e3:.1f} us   -- bench.py roofline.avg_launch_ms (HIP events, same run): {roof['avg_launch_ms'] * 1e3:.1f} us")
        else:
            print(f"\n(timed-region window not found: {len(bw)} backward launches in the capture, expected at least {skip + steps * per_solve})")
    if not a.iters:
        return
    want = [int(x) for x in a.iters.split(",")]
    names = [short(r[0]).split("<")[0] for r in rows]          # k_quadratize<5, true> -> k_quadratize
    starts = [i for i, n in enumerate(names) if n == "k_load_corridor"]
    if a.solve >= len(starts):
        return
    lo = starts[a.solve]
    hi = starts[a.solve + 1] if a.solve + 1 < len(starts) else len(rows)
    its, cur = [], []
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        if n == "k_quadratize":
            if cur:
                its.append(cur)
            cur = []
        cur.append((n, r))
    its.append(cur)
    print(f"\nsolve {a.solve}: {len(its) - 1} lockstep iterations")
    for k in want:
        if k >= len(its):
            break
        it = its[k]
        wall = (it[-1][1][2] - it[0][1][1]) / 1e3
        print(f"iter {k}: wall {wall:.0f} us, quadratize grid {it[0][1][3]}")
        print("   " + " ".join(f"{n[2:8]}:{(r[2] - r[1]) / 1e3:.0f}" for n, r in it))


if __name__ == "__main__":
    main()
