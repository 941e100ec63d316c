#!/bin/bash
# kernel + memory-copy trace of a few batch-of-one Plan calls (tests/cpp/latency_bench.cc) -> gpurun_out/lat_trace/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/lat_trace; rm -rf $out; mkdir -p $out
g++ -std=c++14 -O2 -Iinclude tests/cpp/latency_bench.cc -o $out/latency_bench -Lcilqr_amd/lib -lcilqr_hip -Wl,-rpath,$PWD/cilqr_amd/lib -Wl,-rpath-link,/opt/rocm/lib
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from cilqr_amd import scenario
sc = scenario.generate("mix11", 24, seed=102)
K, cmax = sc["n_steps"] + 1, sc["cmax"]
with open("gpurun_out/lat_trace/s.bin", "wb") as f:
    np.array([24, K, cmax, sc["left"].shape[0], sc["right"].shape[0]], np.int32).tofile(f)
    np.ascontiguousarray(sc["left"]).tofile(f); np.ascontiguousarray(sc["right"]).tofile(f)
    for b in range(24):
        for k, t in (("start", np.float64), ("coarse", np.float64), ("ccount", np.int32), ("corridor", np.float64)):
            np.ascontiguousarray(sc[k][b], t).tofile(f)
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tr -- $out/latency_bench $out/s.bin 1 > $out/result.json 2> $out/err.txt
cat $out/result.json
find $out/tr -name "*.csv" | head
k=$(find $out/tr -name "*kernel_trace.csv" | head -1); m=$(find $out/tr -name "*memory_copy_trace.csv" | head -1)
cp $k $out/kernel_trace.csv; cp $m $out/memory_copy_trace.csv 2>/dev/null
rm -rf $out/tr $out/latency_bench $out/s.bin
