#!/bin/bash
# Memory-pipeline and issue counters of cilqr::k_build_corridors over tests/corridor_bench.py (65536 x 51 knots), one
# rocprofv3 --pmc pass per group with --kernel-trace (durations of the same dispatches), summed per dispatch.
#   usage (through gpurun): bash tools/corridor_counters.sh > gpurun_out/<tag>/corridor_counters.txt
# Reading (profiles/r06_experiments.txt item 10): SQ_* are in quad-cycles summed over the SIMDs; waves per CU =
# 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256); TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = cycles of an L1 miss.
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
o=$(mktemp -d)
for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf "$o/p"
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$o/p" -o p --output-format csv -- python "$root/tests/corridor_bench.py" 65536 mix11 > "$o/log" 2>&1
  python - "$o/p" <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
tot = collections.defaultdict(float); disp = set(); durs = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_build_corridors" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_build_corridors" in r["Kernel_Name"]: durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print({k: round(v / max(len(disp), 1) / 1e6, 2) for k, v in sorted(tot.items())}, "x 1e6 per dispatch | kernel ms", [round(x, 2) for x in durs])
PY
done
rm -rf "$o"
