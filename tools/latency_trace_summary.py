#!/usr/bin/env python
"""Kernel and copy sequence of ONE batch-of-one Plan call out of the trace tools/latency_trace.sh leaves in
gpurun_out/lat_trace/ (rocprofv3 --kernel-trace --memory-copy-trace of tests/cpp/latency_bench.cc):
    python tools/latency_trace_summary.py [call index, default 10] > profiles/rNN_latency_trace.txt"""
import csv
import json
import os
import sys

d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lat_trace")
k = list(csv.DictReader(open(os.path.join(d, "kernel_trace.csv"))))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in k]
mc = os.path.join(d, "memory_copy_trace.csv")
if os.path.exists(mc):
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")) for r in csv.DictReader(open(mc))]
ev.sort()
tails = [i for i, e in enumerate(ev) if "k_tail" in e[2]]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lo, hi = tails[n - 1] + 1, tails[n]
while hi + 1 < len(ev) and "HOST_TO_DEVICE" not in ev[hi + 1][2]:
    hi += 1
# the call starts with its host-to-device copy
while lo < hi and "HOST_TO_DEVICE" not in ev[lo][2]:
    lo += 1
t0 = ev[lo][0]
print(f"one batch-of-one Plan call (call {n} of the trace; rocprofv3 serialises and pads every launch: the durations are the kernels',")
print("the gaps between them are larger than in an unprofiled run).  result.json of the same run:")
print("  " + json.dumps(json.load(open(os.path.join(d, "result.json")))["plan_b1"]))
print(f"{'start us':>10} {'dur us':>9}  what")
for s, e, name in ev[lo:hi + 1]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  {name}")
print(f"span {(ev[hi][1] - t0) / 1e3:.1f} us, {hi - lo + 1} kernels and copies")
