#!/bin/bash
# ThreadSanitizer run of the library's host threads.  Builds, where hipcc is (this container or the GPU
# box): cilqr_amd/lib/libcilqr_hip_tsan.so (make -C cilqr_amd/csrc tsan: the product sources, host side instrumented) and
# tests/cpp/build/tsan_threads (tests/cpp/tsan_threads.cc).  Runs on a GPU: the synchronous call, two solves in flight on one
# handle, a pool of two handles with a second host thread reading it, cilqr_multi over "two devices".
#   usage (through gpurun):  bash tools/tsan_run.sh gpurun_out/r05_tsan.log [tile]
set -u
log=${1:-gpurun_out/tsan.log}
tile=${2:-40}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
mkdir -p "$(dirname "$log")" tests/cpp/build
make -C cilqr_amd/csrc tsan -j8 > /dev/null 2> tests/cpp/build/tsan_make.err || { echo "tsan build failed"; tail -20 tests/cpp/build/tsan_make.err; exit 1; }
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=thread -shared-libsan -Iinclude tests/cpp/tsan_threads.cc -o tests/cpp/build/tsan_threads \
  -Lcilqr_amd/lib -lcilqr_hip_tsan -Wl,-rpath,"$root/cilqr_amd/lib" -Wl,-rpath,"$(dirname "$rt")" -Wl,-rpath-link,/opt/rocm/lib -pthread || exit 1
python - <<'PY' || exit 1
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from cilqr_amd import scenario
n = 64
sc = scenario.generate("mix11", n, seed=123)
K, cmax = sc["n_steps"] + 1, sc["cmax"]
with open("tests/cpp/build/tsan_scenes.bin", "wb") as f:
    np.array([n, K, cmax, sc["left"].shape[0], sc["right"].shape[0]], np.int32).tofile(f)
    np.ascontiguousarray(sc["left"], np.float64).tofile(f)
    np.ascontiguousarray(sc["right"], np.float64).tofile(f)
    for b in range(n):
        np.ascontiguousarray(sc["start"][b], np.float64).tofile(f)
        np.ascontiguousarray(sc["coarse"][b], np.float64).tofile(f)
        np.ascontiguousarray(sc["ccount"][b], np.int32).tofile(f)
        np.ascontiguousarray(sc["corridor"][b], np.float64).tofile(f)
PY
{
  echo "# tools/tsan_run.sh: $(date -u +%FT%TZ), tile $tile ($((64 * tile)) problems), runtime $rt"
  # ThreadSanitizer maps its shadow at fixed addresses: without ASLR (setarch -R) it starts on kernels with many mmap_rnd_bits
  TSAN_OPTIONS="exitcode=66 halt_on_error=0 second_deadlock_stack=1 history_size=4" \
    setarch "$(uname -m)" -R tests/cpp/build/tsan_threads tests/cpp/build/tsan_scenes.bin "$tile"
  echo "# exit code $?"
} > "$log" 2>&1
tail -5 "$log"
grep -c "WARNING: ThreadSanitizer" "$log" | sed 's/^/ThreadSanitizer warnings: /'
