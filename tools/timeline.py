"""Timeline of a rocprofv3 --kernel-trace CSV of bench.py: per solve (k_load_goals .. k_export_hist), stage durations,
GPU busy time (union over queues) and idle gaps.  usage: timeline.py trace.csv"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    short = n.split("(")[0].split("::")[-1].split("<")[0].strip()
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short))
rows.sort()
t0 = rows[0][0]
# union busy
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
starts = [(s, q) for s, e, q, n in rows if n == "k_load_goals"]
ends = [(e, q) for s, e, q, n in rows if n == "k_export_hist"]
print("solves:", len(starts), "queues:", sorted(set(q for _, _, q, _ in rows)))
for i, (s, q) in enumerate(starts):
    print(f" solve {i}: load at {(s - t0) / 1e6:9.3f} ms on queue {q};", end="")
    if i < len(ends): print(f" export_hist end {(ends[i][0] - t0) / 1e6:9.3f} ms on queue {ends[i][1]}  (latency {(ends[i][0] - s) / 1e6:.2f} ms)")
    else: print()
# steady state window: between the 2nd and the last pipelined load
if len(starts) >= 4:
    a, b = starts[1][0], starts[-3][0] if len(starts) > 5 else starts[-1][0]
    win = [(max(s, a), min(e, b), q, n) for s, e, q, n in rows if e > a and s < b]
    busy = union([(s, e) for s, e, q, n in win])
    print(f"window {(b - a) / 1e6:.2f} ms: GPU busy (>=1 kernel) {busy / 1e6:.2f} ms = {100 * busy / (b - a):.1f} %")
    per = collections.defaultdict(float)
    for s, e, q, n in win: per[(q, n)] += (e - s) / 1e6
    perq = collections.defaultdict(float)
    for (q, n), v in per.items(): perq[q] += v
    for q in sorted(perq):
        print(f"  queue {q}: kernel time {perq[q]:.2f} ms; top:", ", ".join(f"{n}={v:.1f}" for (qq, n), v in sorted(per.items(), key=lambda kv: -kv[1]) if qq == q)[:400])
    # overlap: time where two queues both have a kernel running
    qs = sorted(perq)
    if len(qs) >= 2:
        import itertools
        for q1, q2 in itertools.combinations(qs, 2):
            i1 = sorted((s, e) for s, e, q, n in win if q == q1); i2 = sorted((s, e) for s, e, q, n in win if q == q2)
            ov = 0; j = 0
            for s, e in i1:
                while j < len(i2) and i2[j][1] <= s: j += 1
                k = j
                while k < len(i2) and i2[k][0] < e:
                    ov += min(e, i2[k][1]) - max(s, i2[k][0]); k += 1
            print(f"  queues {q1} & {q2} both running: {ov / 1e6:.2f} ms")
