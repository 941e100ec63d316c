// ThreadSanitizer driver for the host threads of the library: the two workers of a handle (first stage /
// finishing stage, solver.hip), the pool that deals solves to handles (pool.hip) and the one-process-several-devices form
// (multi.hip).  Built with -fsanitize=thread and linked against libcilqr_hip_tsan.so, the same sources with the host side
// instrumented (cilqr_amd/csrc/Makefile: tsan; device code is not instrumented, the HIP runtime neither: what the report
// covers is this library's own mutexes, condition variables and shared fields).
//
//   tsan_threads <scenes.bin> <tile>        scenes.bin as tests/cpp/latency_bench.cc reads it; the n scenes are tiled <tile> times
// Exit code 0 and "tsan_threads ok" when every solve of the same inputs gave the same bits; ThreadSanitizer's own reports
// go to stderr and make the exit code 66 (TSAN_OPTIONS exitcode).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "cilqr.h"

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

struct Out {
  std::vector<double> traj, hist;
  std::vector<int32_t> nc, st, ni;
  cilqr_solution_batch sol;
  Out(int B, int K, int M) : traj((size_t)B * K * 10), hist((size_t)B * (M + 1) * 5), nc(B), st(B), ni(B) {
    std::memset(&sol, 0, sizeof(sol));
    sol.memory = CILQR_MEM_HOST;
    sol.traj = traj.data(); sol.cost_hist = hist.data(); sol.n_cost = nc.data(); sol.status = st.data(); sol.n_iter = ni.data();
  }
  bool same(const Out& o) const { return traj == o.traj && hist == o.hist && nc == o.nc && st == o.st && ni == o.ni; }
};

#define CHECK(call)                                                                         \
  do {                                                                                      \
    const int rc_ = (call);                                                                 \
    if (rc_ != CILQR_OK) { std::fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, cilqr_error_string(rc_)); return 10; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  const int tile = std::atoi(argv[2]);
  int32_t hdr[5];
  if (!rd(f, hdr, 5)) return 4;
  const int n = hdr[0], K = hdr[1], cmax = hdr[2], nl = hdr[3], nr = hdr[4];
  std::vector<double> left((size_t)nl * 7), right((size_t)nr * 7);
  if (!rd(f, left.data(), left.size()) || !rd(f, right.data(), right.size())) return 5;
  const int B = n * tile;
  std::vector<double> start((size_t)B * 4), coarse((size_t)B * K * 6), cor((size_t)B * K * cmax * 3);
  std::vector<int32_t> counts((size_t)B * K);
  for (int b = 0; b < n; ++b)
    if (!rd(f, &start[(size_t)b * 4], 4) || !rd(f, &coarse[(size_t)b * K * 6], (size_t)K * 6) ||
        !rd(f, &counts[(size_t)b * K], K) || !rd(f, &cor[(size_t)b * K * cmax * 3], (size_t)K * cmax * 3))
      return 6;
  std::fclose(f);
  for (int t = 1; t < tile; ++t) {
    std::memcpy(&start[(size_t)t * n * 4], start.data(), (size_t)n * 4 * 8);
    std::memcpy(&coarse[(size_t)t * n * K * 6], coarse.data(), (size_t)n * K * 6 * 8);
    std::memcpy(&counts[(size_t)t * n * K], counts.data(), (size_t)n * K * 4);
    std::memcpy(&cor[(size_t)t * n * K * cmax * 3], cor.data(), (size_t)n * K * cmax * 3 * 8);
  }
  cilqr_config cfg;
  CHECK(cilqr_default_config(&cfg, K - 1));
  cilqr_problem_batch in;
  std::memset(&in, 0, sizeof(in));
  in.batch = B; in.n_knots = K; in.cmax = cmax; in.memory = CILQR_MEM_HOST;
  in.start = start.data(); in.coarse = coarse.data(); in.corridor = cor.data(); in.corridor_count = counts.data();
  in.n_left = nl; in.n_right = nr; in.left_lane = left.data(); in.right_lane = right.data();
  const int M = cfg.max_iter, smax = nl > nr ? nl : nr;

  // 1. the reference answer: the synchronous call on a handle of its own
  Out ref(B, K, M);
  {
    cilqr_handle h = nullptr;
    CHECK(cilqr_create(&cfg, 0, B, cmax, smax, &h));
    CHECK(cilqr_solve_batch(h, &in, &ref.sol));
    CHECK(cilqr_destroy(h));
  }
  // 2. ONE handle kept full: two solves in flight (first stage / finishing stage on two worker threads) and a third queued
  //    whose host arrays the transfer thread uploads meanwhile (ABI 6); host arrays in and out, so the upload-ahead and the
  //    ragged download run on every solve
  {
    cilqr_handle h = nullptr;
    CHECK(cilqr_create(&cfg, 0, B, cmax, smax, &h));
    Out a(B, K, M), b(B, K, M), c(B, K, M);
    for (int round = 0; round < 3; ++round) {
      CHECK(cilqr_submit(h, &in, &a.sol));
      CHECK(cilqr_submit(h, &in, &b.sol));
      CHECK(cilqr_submit(h, &in, &c.sol));
      if (cilqr_submit(h, &in, &c.sol) != CILQR_ERR_STATE) { std::fprintf(stderr, "a fourth submit was accepted\n"); return 21; }
      CHECK(cilqr_wait(h));                                   // a is complete ...
      if (!a.same(ref)) { std::fprintf(stderr, "first of three: results differ\n"); return 22; }
      CHECK(cilqr_submit(h, &in, &a.sol));                    // ... and its arrays go in again behind b and c
      CHECK(cilqr_wait(h));
      CHECK(cilqr_wait(h));
      CHECK(cilqr_wait(h));
      if (!a.same(ref) || !b.same(ref) || !c.same(ref)) { std::fprintf(stderr, "three solves on one handle: results differ\n"); return 20; }
    }
    CHECK(cilqr_destroy(h));
  }
  // 3. a pool of two handles, kept full from one thread while another asks for its depth / memory
  {
    cilqr_pool_handle p = nullptr;
    CHECK(cilqr_pool_create(&cfg, 0, 2, B, cmax, smax, &p));
    const int depth = cilqr_pool_depth(p);
    std::vector<Out> outs;
    for (int k = 0; k < depth; ++k) outs.emplace_back(B, K, M);
    for (auto& o : outs) o.sol.traj = o.traj.data(), o.sol.cost_hist = o.hist.data(), o.sol.n_cost = o.nc.data(),
                         o.sol.status = o.st.data(), o.sol.n_iter = o.ni.data();   // vectors moved: pointers again
    volatile bool stop = false;
    std::thread reader([&] { while (!stop) { (void)cilqr_pool_depth(p); (void)cilqr_pool_device_bytes(p); std::this_thread::yield(); } });
    int submitted = 0, collected = 0;
    const int total = 3 * depth;
    int rc = CILQR_OK;
    while (collected < total && rc == CILQR_OK) {
      while (submitted < total && submitted - collected < depth && rc == CILQR_OK) {
        rc = cilqr_pool_submit(p, &in, &outs[submitted % depth].sol);
        ++submitted;
      }
      if (rc == CILQR_OK) rc = cilqr_pool_wait(p);
      if (rc == CILQR_OK && !outs[collected % depth].same(ref)) { std::fprintf(stderr, "pool: results differ\n"); rc = -100; }
      ++collected;
    }
    stop = true;
    reader.join();
    if (rc != CILQR_OK) { std::fprintf(stderr, "pool: rc %d\n", rc); return 30; }
    CHECK(cilqr_pool_destroy(p));
  }
  // 4. one process, "two devices" (device 0 listed twice: two handles, two shard threads)
  {
    const int32_t devs[2] = {0, 0};
    cilqr_multi_handle m = nullptr;
    CHECK(cilqr_multi_create(&cfg, devs, 2, B, cmax, smax, &m));
    Out o(B, K, M);
    for (int round = 0; round < 2; ++round) {
      CHECK(cilqr_multi_solve(m, &in, &o.sol));
      if (!o.same(ref)) { std::fprintf(stderr, "multi: results differ\n"); return 40; }
    }
    CHECK(cilqr_multi_destroy(m));
  }
  std::printf("tsan_threads ok: %d problems, sync / two in flight / pool of two / multi bit-identical\n", B);
  return 0;
}
