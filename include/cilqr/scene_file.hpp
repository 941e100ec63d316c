// include/cilqr/scene_file.hpp -- header-only C++14 reader of the scene files written by
// cilqr_amd/scene_io.py: one centre line and B scenes, each with a start state, a
// coarse trajectory, static polygons and dynamic obstacles (body-frame polygon + trajectory), i.e.
// the content of the reference's six ROS messages (msg/*.msg) / its pickle {center, static, dynamic}
// (script/reference_publisher.py:232-236).  Little-endian; layout in cilqr_amd/scene_io.py.
//
// ObstaclePoints() restates what the reference's Environment answers from that data
// (Environment::QueryStaticObstaclesPoints / QueryDynamicObstaclesPoints, environment.cpp:134-182,
// with the polygons placed as in PlanningNode::DynamicObstaclesCallback, planning_node.cc:63-78):
// the obstacle corner points valid at a time -- the `points` input of cilqr_build_corridors.
#ifndef CILQR_SCENE_FILE_HPP_
#define CILQR_SCENE_FILE_HPP_

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace cilqr {

struct ScenePoint2 {
  double x, y;
};
struct SceneDynamicObstacle {
  std::vector<ScenePoint2> polygon;                 // body frame
  std::vector<std::array<double, 4>> trajectory;    // time, x, y, theta
};
struct Scene {
  std::array<double, 4> start;                      // x, y, theta, v
  std::vector<std::array<double, 6>> coarse;        // x, y, theta, v, a, delta per knot
  std::vector<std::vector<ScenePoint2>> statics;    // world frame
  std::vector<SceneDynamicObstacle> dynamics;

  // obstacle corner points valid at `time`: static polygons, then the dynamic obstacles whose
  // trajectory spans `time` (to 1e-10), each placed by the first sample later than time - 1e-10
  std::vector<ScenePoint2> ObstaclePoints(double time) const {
    constexpr double kEps = 1e-10;
    std::vector<ScenePoint2> out;
    for (const auto& p : statics) out.insert(out.end(), p.begin(), p.end());
    for (const auto& d : dynamics) {
      if (d.trajectory.empty()) continue;
      if (d.trajectory.front()[0] > time + kEps || d.trajectory.back()[0] < time - kEps) continue;
      size_t i = 0;
      while (i + 1 < d.trajectory.size() && !(time < d.trajectory[i][0] + kEps)) ++i;
      const double x = d.trajectory[i][1], y = d.trajectory[i][2], th = d.trajectory[i][3];
      const double c = std::cos(th), s = std::sin(th);
      for (const auto& v : d.polygon)   // Pose::transform (pose.h:40-46): x + rx cos - ry sin, in that order
        out.push_back(ScenePoint2{x + v.x * c - v.y * s, y + v.x * s + v.y * c});
    }
    return out;
  }
};
struct SceneFile {
  double dt = 0.0;
  std::vector<std::array<double, 7>> center;        // s, x, y, theta, kappa, left_bound, right_bound
  std::vector<Scene> scenes;

  // road barriers: the centre line shifted by +left_bound / -right_bound along its normal
  void RoadBarriers(std::vector<ScenePoint2>* left, std::vector<ScenePoint2>* right) const {
    left->clear();
    right->clear();
    for (const auto& c : center) {
      left->push_back(ScenePoint2{c[1] - c[5] * std::sin(c[3]), c[2] + c[5] * std::cos(c[3])});
      right->push_back(ScenePoint2{c[1] + c[6] * std::sin(c[3]), c[2] - c[6] * std::cos(c[3])});
    }
  }
};

// returns an empty string on success, a message otherwise
inline std::string LoadSceneFile(const char* path, SceneFile* out) {
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return "cannot open file";
  auto fail = [&](const char* why) {
    std::fclose(f);
    return std::string(why);
  };
  char magic[8];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "CILQRSC1", 8) != 0) return fail("not a CILQR scene file");
  uint32_t hdr[3];
  double dt;
  if (std::fread(hdr, 4, 3, f) != 3 || std::fread(&dt, 8, 1, f) != 1) return fail("truncated header");
  if (hdr[0] != 1) return fail("unsupported version");
  const uint32_t B = hdr[1], K = hdr[2];
  auto u32 = [&](uint32_t* v) { return std::fread(v, 4, 1, f) == 1; };
  auto f64 = [&](double* p, size_t n) { return n == 0 || std::fread(p, 8, n, f) == n; };
  auto poly = [&](std::vector<ScenePoint2>* p) {
    uint32_t m;
    if (!u32(&m)) return false;
    p->resize(m);
    return f64(reinterpret_cast<double*>(p->data()), (size_t)m * 2);
  };
  out->dt = dt;
  uint32_t nc;
  if (!u32(&nc)) return fail("truncated centre line");
  out->center.resize(nc);
  if (!f64(reinterpret_cast<double*>(out->center.data()), (size_t)nc * 7)) return fail("truncated centre line");
  out->scenes.assign(B, Scene());
  for (auto& s : out->scenes) {
    s.coarse.resize(K);
    if (!f64(s.start.data(), 4) || !f64(reinterpret_cast<double*>(s.coarse.data()), (size_t)K * 6))
      return fail("truncated scene");
    uint32_t ns, nd;
    if (!u32(&ns)) return fail("truncated scene");
    s.statics.resize(ns);
    for (auto& p : s.statics)
      if (!poly(&p)) return fail("truncated polygon");
    if (!u32(&nd)) return fail("truncated scene");
    s.dynamics.resize(nd);
    for (auto& d : s.dynamics) {
      uint32_t T;
      if (!poly(&d.polygon) || !u32(&T)) return fail("truncated obstacle");
      d.trajectory.resize(T);
      if (!f64(reinterpret_cast<double*>(d.trajectory.data()), (size_t)T * 4)) return fail("truncated trajectory");
    }
  }
  if (std::fgetc(f) != EOF) return fail("trailing bytes");
  std::fclose(f);
  return std::string();
}

}  // namespace cilqr

#endif  // CILQR_SCENE_FILE_HPP_
