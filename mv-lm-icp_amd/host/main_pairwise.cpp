// Headless counterpart of the reference's src/main_pairwise.cpp:29-134: load one cloud, apply a known noisy transform,
// recover it with each parameterization from index-aligned pairs, print timings and poseDiff.  Flags: --pointToPlane
// (false), --cloud FILE (../samples/Bunny_RealData/cloudXYZ_0.xyz), --device, --dump_P FILE (write the ground-truth
// transform P of this run, 4x4 row-major), --precision N, --drop_phantom_row (load exactly the file's rows; default: the reference's
// loadXYZ, which appends a duplicate of the last row), --noise_stream libstdc++|libc++ (std::normal_distribution's variate order;
// default libstdc++ = this build's own).  `pairwise --noise_stream libc++` on the reference's cloudXYZ_0.xyz prints the README's own
// lines (README.md:141-146): ceres CeresAngleAxis diff_tra:7.76957e-11, ceres EigenQuaternion diff_tra:6.31278e-11.
#include <chrono>
#include <iostream>

#include "common_io.h"
#include "flags.h"
#include "frame.h"

using namespace mvicp;

int main(int argc, char** argv) {
  Flags F(argc, argv);
  const bool pointToPlane = F.b("pointToPlane", false);
  Session::get().device = F.i("device", 0);
  std::vector<Vector3d> pts, nor;
  if (!loadXYZ(F.s("cloud", "../samples/Bunny_RealData/cloudXYZ_0.xyz"), pts, nor, !F.b("drop_phantom_row", false)) || pts.empty()) return 1;
  noiseStream() = F.s("noise_stream", "libstdc++") == "libc++" ? 1 : F.s("noise_stream", "libstdc++") == "g++" ? 2 : 0;
  // main_pairwise.cpp:44-56: q = Rx(pi/4) Ry(1) Rz(-0.2), t = (.01,-.01,-.005), P = addNoise(Pclean, 0.1, 0.1)
  auto rot = [](int axis, double a) {
    Isometry3d R;
    const double c = std::cos(a), s = std::sin(a);
    const int i = (axis + 1) % 3, j = (axis + 2) % 3;
    R(i, i) = c; R(i, j) = -s; R(j, i) = s; R(j, j) = c;
    return R;
  };
  Isometry3d Pclean = rot(0, M_PI_4) * rot(1, 1.0) * rot(2, -0.2);
  Pclean.setTranslation(Vector3d(.01, -0.01, -0.005));
  const Isometry3d P = addNoise(Pclean, 0.1, 0.1);
  std::vector<Vector3d> ptsTra(pts.size()), norTra(nor.size());
  for (size_t i = 0; i < pts.size(); ++i) { ptsTra[i] = P * pts[i]; norTra[i] = P.rotate(nor[i]); }
  struct Run { const char* name; Isometry3d out; double ms; };
  std::vector<Run> runs;
  if (!F.s("dump_P", "").empty()) saveMatrix4d(F.s("dump_P", ""), P);   // the ground-truth transform of this run (tests)
  try {
    {  // main_pairwise.cpp:74-76,93-95: closed form first
      const auto t0 = std::chrono::steady_clock::now();
      const Isometry3d out = pointToPlane ? ICP_Closedform::pointToPlane(pts, ptsTra, norTra) : ICP_Closedform::pointToPoint(pts, ptsTra);
      runs.push_back(Run{"closed form      ", out, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()});
    }
    auto timed = [&](const char* name, Isometry3d (*fn)(std::vector<Vector3d>&, std::vector<Vector3d>&, std::vector<Vector3d>&, bool)) {
      const auto t0 = std::chrono::steady_clock::now();
      const Isometry3d out = fn(pts, ptsTra, norTra, pointToPlane);
      runs.push_back(Run{name, out, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()});
    };
    timed("ceres CeresAngleAxis", [](std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n, bool pl) {
      return pl ? ICP_Ceres::pointToPlane_CeresAngleAxis(s, d, n) : ICP_Ceres::pointToPoint_CeresAngleAxis(s, d); });
    timed("ceres EigenQuaternion", [](std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n, bool pl) {
      return pl ? ICP_Ceres::pointToPlane_EigenQuaternion(s, d, n) : ICP_Ceres::pointToPoint_EigenQuaternion(s, d); });
    timed("ceres SophusSE3", [](std::vector<Vector3d>& s, std::vector<Vector3d>& d, std::vector<Vector3d>& n, bool pl) {
      return pl ? ICP_Ceres::pointToPlane_SophusSE3(s, d, n) : ICP_Ceres::pointToPoint_SophusSE3(s, d); });
  } catch (const std::exception& ex) {
    std::cerr << ex.what() << std::endl;
    return 2;
  }
  std::cout << "=====  TIMINGS ====" << std::endl;
  for (const Run& r : runs) std::cout << r.name << ":\t" << r.ms / 1e3 << std::endl;
  std::cout << std::endl << "=====  Accurracy ====" << std::endl;
  std::cout.precision(F.i("precision", 6));
  for (const Run& r : runs) std::cout << r.name << poseDiff(P, r.out) << std::endl;
  Session::get().reset();
  return 0;
}
