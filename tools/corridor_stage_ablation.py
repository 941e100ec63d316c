#!/usr/bin/env python
"""Where the corridor kernel's time goes: a TEST-ONLY variant of the library whose k_build_corridors returns early after
a chosen stage (CILQR_COR_STOP=1..7; timing only, the outputs are garbage) and reports the mean lifetime of a wavefront
(s_memrealtime at entry and exit; waves in flight = lifetime x waves / kernel time).
    python tools/corridor_stage_ablation.py build      # here: patches a COPY of csrc/, builds cilqr_amd/lib/variants/libcilqr_hip_corstages.so
    gpurun -- python tools/corridor_stage_ablation.py run [batch] [family]     # on the GPU box: one line per stage
The product library is not touched.  Records: profiles/r06_experiments.txt items 9 and 10."""
import json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "cilqr_amd", "lib", "variants", "libcilqr_hip_corstages.so")
STAGES = [(1, "filter + sphere flip"), (7, "rank sort of hull 1"), (2, "chains of hull 1"), (3, "star polygon + interior point"),
          (4, "hull 2"), (5, "dual points"), (6, "hull 3"), (0, "polygon + half-planes out (= the whole kernel)")]


def patch(s):
    def once(old, new):
        nonlocal s
        assert old in s, old
        s = s.replace(old, new, 1)
    once("namespace cilqr {\n\nnamespace {\n",
         "namespace cilqr {\n__device__ unsigned long long g_probe[4];\n__device__ int g_stop;\n"
         "#define STOP_AT(k, expr) if (stop == (k)) { ccount[t] = (int)(expr); return; }\n\nnamespace {\n")
    once("  int k = 0, lo = 0;", "  if (g_stop == 7) return n + order.get(0) + order.get(n - 1);\n  int k = 0, lo = 0;")
    once("  const double ox = knots[3 * t]", "  const unsigned long long probe_t0 = wall_clock64();\n  const int stop = g_stop;\n  const double ox = knots[3 * t]")
    once("  int m = 0;\n  if (nf < 4) {", "  STOP_AT(1, nf + (nf > 0 ? flip[nf - 1].x : 0.0f))\n  int m = 0;\n  if (nf < 4) {")
    s, k = re.subn(r"(    const int n1 = hull_indices<.*\n)", r"\1    STOP_AT(7, n1)\n    STOP_AT(2, n1 + hull.get(0))\n", s, 1); assert k == 1
    s, k = re.subn(r"(      const int n2 = hull_indices<.*\n)", r"      STOP_AT(3, ix + iy + vd[0].x)\n\1      STOP_AT(4, n2 + hull.get(0))\n", s, 1); assert k == 1
    s, k = re.subn(r"(        const int n3 = hull_indices<.*\n)", r"        STOP_AT(5, nt + dual[0].x)\n\1        STOP_AT(6, n3 + hull.get(0))\n", s, 1); assert k == 1
    once("  ccount[t] = code < 0 ? code : m;\n",
         "  ccount[t] = code < 0 ? code : m;\n  if (threadIdx.x == 0) { atomicAdd(&g_probe[0], wall_clock64() - probe_t0); atomicAdd(&g_probe[1], 1ull); }\n")
    once("void launch_build_corridors(int n, const CorridorParams& cp,", "void launch_build_corridors_(int n, const CorridorParams& cp,")
    once("}  // namespace cilqr\n", """void launch_build_corridors(int n, const CorridorParams& cp, const double* knots, const double* points,
                            const int* count, int pmax, double* corridor, int* ccount, int cmax, int* n_failed,
                            double* polygons, hipStream_t st) {
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof z);
  int stop_at = getenv("CILQR_COR_STOP") ? atoi(getenv("CILQR_COR_STOP")) : 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stop), &stop_at, sizeof stop_at);
  launch_build_corridors_(n, cp, knots, points, count, pmax, corridor, ccount, cmax, n_failed, polygons, st);
  (void)hipStreamSynchronize(st);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_probe), sizeof z);
  fprintf(stderr, "PROBE waves %llu mean_life_us %.1f\\n", z[1], z[1] ? (double)z[0] / z[1] / 100.0 : 0.0);
}
}  // namespace cilqr
""")
    once("#include <stdint.h>\n", "#include <stdint.h>\n#include <stdlib.h>\n#include <stdio.h>\n")
    return s


def build():
    tmp = tempfile.mkdtemp(prefix="corstages_")
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    src = os.path.join(tmp, "cilqr_amd", "csrc")
    os.makedirs(src)
    for f in os.listdir(os.path.join(ROOT, "cilqr_amd", "csrc")):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            shutil.copy(os.path.join(ROOT, "cilqr_amd", "csrc", f), src)
    p = os.path.join(src, "kernels_corridor.hip")
    patched = patch(open(p).read())
    open(p, "w").write(patched)
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    subprocess.run(["make", "-C", src, "-j4", "OUT=" + VARIANT], check=True, stdout=subprocess.DEVNULL)
    shutil.rmtree(tmp)
    print("built", VARIANT)


def run(batch, family):
    prev, rows = 0.0, []
    for stop, name in STAGES:
        env = dict(os.environ, CILQR_LIB=VARIANT, CILQR_COR_STOP=str(stop), CORRIDOR_BENCH_NO_ASSERT="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "corridor_bench.py"), str(batch), family],
                           env=env, capture_output=True, text=True)
        ms = float(re.search(r'"seconds": ([0-9.]+)', r.stdout).group(1)) * 1e3
        life = re.findall(r"mean_life_us ([0-9.]+)", r.stderr)
        rows.append({"stop_after": name, "cumulative_ms": round(ms, 3), "stage_ms": round(ms - prev, 3),
                     "mean_wave_life_us": float(life[-1]) if life and stop == 0 else None})
        prev = ms
    print(json.dumps({"batch": batch, "family": family, "stages": rows}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 65536, sys.argv[3] if len(sys.argv) > 3 else "mix11")
    else:
        sys.exit(__doc__)
