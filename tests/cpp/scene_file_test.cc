// Reads a scene file with include/cilqr/scene_file.hpp and prints what tests/test_scene_io.py checks
// against the Python side: counts, a checksum of every array, obstacle points per knot.
//   scene_file_test <scenes.cqs>
#include <cstdio>

#include "cilqr/scene_file.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  cilqr::SceneFile f;
  const std::string err = cilqr::LoadSceneFile(argv[1], &f);
  if (!err.empty()) {
    std::fprintf(stderr, "%s\n", err.c_str());
    return 1;
  }
  std::printf("%zu %zu %.17g\n", f.scenes.size(), f.center.size(), f.dt);
  double cs = 0.0;
  for (const auto& c : f.center)
    for (double v : c) cs += v;
  std::printf("%.17g\n", cs);
  for (const auto& s : f.scenes) {
    double sum = 0.0;
    for (double v : s.start) sum += v;
    for (const auto& k : s.coarse)
      for (double v : k) sum += v;
    std::printf("%zu %zu %zu %.17g", s.coarse.size(), s.statics.size(), s.dynamics.size(), sum);
    for (size_t k = 0; k < s.coarse.size(); ++k) {
      const auto pts = s.ObstaclePoints(f.dt * (double)k);
      double ps = 0.0;
      for (const auto& p : pts) ps += p.x + 2.0 * p.y;
      std::printf(" %zu:%.17g", pts.size(), ps);
    }
    std::printf("\n");
  }
  std::vector<cilqr::ScenePoint2> l, r;
  f.RoadBarriers(&l, &r);
  double bs = 0.0;
  for (const auto& p : l) bs += p.x + 2.0 * p.y;
  for (const auto& p : r) bs -= p.x + 2.0 * p.y;
  std::printf("%.17g\n", bs);
  return 0;
}
