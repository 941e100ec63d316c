#!/usr/bin/env python
"""Per-kernel roofline record from the committed rocprofv3 summaries of one round:
    python tools/kernel_rooflines.py profiles r02 > profiles/r02_kernel_rooflines.json
Inputs (written by tools/record_profiles.sh, all from `python bench.py --steps 1 --warmup 0 --in-flight 1`):
    <tag>_pmc_all_kernels_1.txt  SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY ...
    <tag>_pmc_all_kernels_2.txt  LDS / VMEM / SALU instruction counts, SQ_LDS_BANK_CONFLICT
    <tag>_pmc_all_kernels_3.txt  FETCH_SIZE (KiB)     <tag>_pmc_all_kernels_4.txt  WRITE_SIZE (KiB)
    <tag>_kernel_stats.txt       kernel durations (rocprofv3 --kernel-trace, same command without counters)
What is derived, per kernel, over ALL its launches of one solve:
    valu_per_wave        SQ_INSTS_VALU / SQ_WAVES
    valu_issue_frac      SQ_INSTS_VALU quad-cycles / (kernel time x 1024 SIMDs x clock/4): the share of the chip's
                         vector-issue slots the kernel used (a wave64 VALU instruction holds its SIMD for one quad-cycle;
                         fp64 add/mul/fma issue at that rate on gfx950) -- the roofline fraction of an fp64-ALU-bound kernel
    valu_active_share    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a wave's lifetime with a VALU instruction in flight
    wait_share           SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt / barrier), issue_stall_share likewise
    hbm_gbs              (FETCH_SIZE x f + WRITE_SIZE) KiB / kernel time, f = 1 and f = 2 (the gfx950 FETCH_SIZE note of
                         MI355X_MICROARCH.md applies to 16 B/lane loads; these kernels mix 8 and 16 B loads)
The SQ counters are sums over the launches of the capture; kernel time is the sum of the same launches' durations
in the trace (captures and trace hold a different number of solves: everything is normalised per solve)."""
import json
import re
import sys

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4
HBM_PEAK = 8000.0


def table(path):
    rows = {}
    lines = open(path).read().strip().split("\n")
    head = lines[0].split()
    for l in lines[1:]:
        name, t = l[:28].strip().replace(" ", ""), l[28:].split()   # the name column is 28 wide (template arguments hold spaces)
        if len(t) != len(head) - 1:
            continue
        rows[name] = {h: float(v) for h, v in zip(head[1:], t)}
    return rows


def main():
    d, tag = sys.argv[1], sys.argv[2]
    sq = table(f"{d}/{tag}_pmc_all_kernels_1.txt")
    mem = table(f"{d}/{tag}_pmc_all_kernels_2.txt")
    fe = table(f"{d}/{tag}_pmc_all_kernels_3.txt")
    wr = table(f"{d}/{tag}_pmc_all_kernels_4.txt")
    # durations
    dur = {}
    for l in open(f"{d}/{tag}_kernel_stats.txt"):
        m = re.match(r"^\s*(\d+)\s+([\d.]+)\s+([\d.]+)", l[44:])   # the name column is 44 wide
        name = l[:44].strip().replace(" ", "")
        if m and name.startswith("k_"):
            dur[name] = (int(m.group(1)), float(m.group(2)))
    solves_trace = dur["k_load_goals"][0]
    solves_pmc = int(sq["k_load_goals"]["disp"])
    # Round 6: the counter passes carry --kernel-trace, so the durations of the very dispatches the counters were summed over are
    # in the same capture (column dur_ms of file 1).  Counters and durations from different runs held different mixes of
    # synchronous and submitted solves (other thresholds, other kernels): a fraction of 1.14 in round 5's record came from that.
    same_run = all("dur_ms" in v for v in sq.values()) and len(sq) > 0
    if same_run:
        dur = {k: (int(v["disp"]), v["dur_ms"]) for k, v in sq.items()}
        solves_trace = solves_pmc
    out = {"source": [f"{tag}_pmc_all_kernels_{i}.txt" for i in (1, 2, 3, 4)] + [f"{tag}_kernel_stats.txt"],
           "command": "python bench.py --steps 1 --warmup 0 --in-flight 1 --cpu-sample 0 (counters: one rocprofv3 --pmc pass per file)",
           "per": "solve of 65536 problems", "clock_hz_assumed": CLOCK_HZ,
           "durations_from": ("the counter capture itself (rocprofv3 --pmc ... --kernel-trace: the same dispatches)" if same_run
                              else f"{tag}_kernel_stats.txt (another run of the same command)"), "kernels": {}}
    for k, (calls, total_ms) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        if k not in sq:
            continue
        s = sq[k]
        t_solve = total_ms / solves_trace * 1e-3
        valu = s["SQ_INSTS_VALU"] / solves_pmc
        e = {
            "launches_per_solve": calls / solves_trace, "ms_per_solve": round(t_solve * 1e3, 3),
            "valu_per_wave": round(s["SQ_INSTS_VALU"] / s["SQ_WAVES"], 1),
            "valu_issue_frac": round(valu / (t_solve * CLOCK_HZ / 4 * SIMDS), 4),
            "valu_active_share": round(s["SQ_ACTIVE_INST_VAL"] / s["SQ_WAVE_CYCLES"], 4),
            "wait_share": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4),
            "issue_stall_share": round(s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
        }
        if k in mem:
            m = mem[k]
            e["lds_bank_conflict_per_lds_inst"] = round(m["SQ_LDS_BANK_CONFLI"] / max(m["SQ_INSTS_LDS"], 1.0), 3)
            e["salu_per_wave"] = round(m["SQ_INSTS_SALU"] / s["SQ_WAVES"], 1)
        if k in fe and k in wr:
            f_kib = fe[k]["FETCH_SIZE"] / solves_pmc
            w_kib = wr[k]["WRITE_SIZE"] / solves_pmc
            g1 = (f_kib + w_kib) * 1024 / t_solve / 1e9
            g2 = (2 * f_kib + w_kib) * 1024 / t_solve / 1e9
            e["hbm_gbs"] = [round(g1, 1), round(g2, 1)]
            e["hbm_frac"] = [round(g1 / HBM_PEAK, 4), round(g2 / HBM_PEAK, 4)]
        e["bound"] = "fp64 VALU issue" if e["valu_issue_frac"] >= max(e.get("hbm_frac", [0, 0])) else "hbm"
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
