// Host-path probe (VERDICT r05 missing #3): what the box gives a caller that holds its arrays in host memory.
// Measures, for the payload of one bench batch (1.46 GB in, 0.3-0.8 GB out):
//   pinned and pageable H2D / D2H rates by chunk size, both directions at once, memcpy pageable -> pinned and memset by thread
//   count, hipHostRegister of the caller's block.  Build: hipcc -O2 -std=c++17 tools/host_path_probe.cc -o /tmp/host_path_probe -lpthread
// Prints one JSON object.  No library code involved: this sizes the design of the host-array path, nothing more.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

static void par(int nt, size_t bytes, const std::function<void(size_t, size_t)>& f) {
  std::vector<std::thread> th;
  const size_t per = (bytes / nt + 4095) & ~(size_t)4095;
  for (int t = 0; t < nt; ++t) {
    const size_t a = std::min(bytes, per * t), b = std::min(bytes, per * (t + 1));
    th.emplace_back([=, &f] { if (b > a) f(a, b); });
  }
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  const size_t total = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1460) * (size_t)1000000;
  CK(hipSetDevice(0));
  char *dev = nullptr, *dev2 = nullptr, *pin = nullptr, *pin2 = nullptr;
  CK(hipMalloc(&dev, total));
  CK(hipMalloc(&dev2, total));
  CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
  CK(hipHostMalloc(&pin2, total, hipHostMallocDefault));
  char* page = static_cast<char*>(std::malloc(total));
  char* page2 = static_cast<char*>(std::malloc(total));
  std::memset(page, 1, total);
  std::memset(page2, 2, total);
  std::memset(pin, 3, total);
  std::memset(pin2, 4, total);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  std::printf("{\"bytes\": %zu", total);

  // pinned H2D / D2H by chunk size
  for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, total}) {
    for (int dir = 0; dir < 2; ++dir) {
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = clk::now();
        for (size_t o = 0; o < total; o += chunk) {
          const size_t n = std::min(chunk, total - o);
          if (dir == 0) CK(hipMemcpyAsync(dev + o, pin + o, n, hipMemcpyHostToDevice, s1));
          else CK(hipMemcpyAsync(pin + o, dev + o, n, hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s1));
        best = std::min(best, since(t0));
      }
      std::printf(", \"pinned_%s_chunk%zuMB_GBs\": %.2f", dir ? "d2h" : "h2d", chunk >> 20, total / best / 1e9);
    }
  }
  {  // both directions at once
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = clk::now();
      CK(hipMemcpyAsync(dev, pin, total, hipMemcpyHostToDevice, s1));
      CK(hipMemcpyAsync(pin2, dev2, total, hipMemcpyDeviceToHost, s2));
      CK(hipStreamSynchronize(s1));
      CK(hipStreamSynchronize(s2));
      best = std::min(best, since(t0));
    }
    std::printf(", \"pinned_duplex_each_GBs\": %.2f", total / best / 1e9);
  }
  {  // pageable, as the library did until round 5
    double best = 1e9, bestd = 1e9;
    for (int rep = 0; rep < 2; ++rep) {
      auto t0 = clk::now();
      CK(hipMemcpyAsync(dev, page, total, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
      best = std::min(best, since(t0));
      t0 = clk::now();
      CK(hipMemcpyAsync(page2, dev, total, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s1));
      bestd = std::min(bestd, since(t0));
    }
    std::printf(", \"pageable_h2d_GBs\": %.2f, \"pageable_d2h_GBs\": %.2f", total / best / 1e9, total / bestd / 1e9);
  }
  for (int nt : {1, 2, 4, 8, 12}) {   // host copies and fills
    double best = 1e9, bestz = 1e9, bestb = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = clk::now();
      par(nt, total, [&](size_t a, size_t b) { std::memcpy(pin + a, page + a, b - a); });
      best = std::min(best, since(t0));
      t0 = clk::now();
      par(nt, total, [&](size_t a, size_t b) { std::memset(page2 + a, 0, b - a); });
      bestz = std::min(bestz, since(t0));
      t0 = clk::now();
      par(nt, total, [&](size_t a, size_t b) { std::memcpy(page2 + a, pin2 + a, b - a); });
      bestb = std::min(bestb, since(t0));
    }
    std::printf(", \"memcpy_page_to_pin_%dthr_GBs\": %.2f, \"memcpy_pin_to_page_%dthr_GBs\": %.2f, \"memset_page_%dthr_GBs\": %.2f", nt,
                total / best / 1e9, nt, total / bestb / 1e9, nt, total / bestz / 1e9);
  }
  {  // pipelined: N threads fill chunks of a pinned ring, the copy engine drains them
    for (int nt : {2, 4, 8}) {
      const size_t chunk = (size_t)32 << 20;
      const int ring = 4;
      std::vector<hipEvent_t> ev(ring);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = clk::now();
        int k = 0;
        for (size_t o = 0; o < total; o += chunk, ++k) {
          const size_t n = std::min(chunk, total - o);
          char* slot = pin + (size_t)(k % ring) * chunk;
          if (k >= ring) CK(hipEventSynchronize(ev[k % ring]));
          par(nt, n, [&](size_t a, size_t b) { std::memcpy(slot + a, page + o + a, b - a); });
          CK(hipMemcpyAsync(dev + o, slot, n, hipMemcpyHostToDevice, s1));
          CK(hipEventRecord(ev[k % ring], s1));
        }
        CK(hipStreamSynchronize(s1));
        best = std::min(best, since(t0));
      }
      std::printf(", \"staged_ring_h2d_%dthr_GBs\": %.2f", nt, total / best / 1e9);
      for (auto& e : ev) CK(hipEventDestroy(e));
    }
  }
  {  // pin the caller's block in place
    auto t0 = clk::now();
    hipError_t e = hipHostRegister(page, total, hipHostRegisterDefault);
    const double treg = since(t0);
    if (e == hipSuccess) {
      t0 = clk::now();
      CK(hipMemcpyAsync(dev, page, total, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
      const double tcp = since(t0);
      t0 = clk::now();
      CK(hipHostUnregister(page));
      std::printf(", \"host_register_s\": %.4f, \"registered_h2d_GBs\": %.2f, \"host_unregister_s\": %.4f", treg, total / tcp / 1e9, since(t0));
    } else {
      std::printf(", \"host_register_error\": \"%s\"", hipGetErrorString(e));
    }
  }
  {  // a kernel reading pinned host memory directly (zero-copy load) is what k_load_* would do: plain device copy kernel rate
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    hipError_t e = hipPointerGetAttributes(&at, pin);
    std::printf(", \"attr_pinned_type\": %d", e == hipSuccess ? (int)at.type : -1);
    e = hipPointerGetAttributes(&at, page);
    std::printf(", \"attr_pageable_err\": %d, \"attr_pageable_type\": %d", (int)e, e == hipSuccess ? (int)at.type : -1);
    (void)hipGetLastError();
  }
  std::printf(", \"hw_threads\": %u}\n", std::thread::hardware_concurrency());
  return 0;
}
