// sweep_driver.cpp -- TEST DRIVER for lsm2d_host::LoopClosureSweep (the C++ mirror of the multi-device loop-closure sweep, no Python
// in the process): reads a packed batch written by the GPU test, runs the sweep on the listed devices, writes poses / information /
// status / last statistics / acceptance bits back as raw arrays.
//   sweep_driver dir n_devices dev0 [dev1 ...]      files in dir: scans.bin offsets.bin map.bin index.bin x0.bin params.txt
#include <lsm2d.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

using namespace lsm2d_host;

template <typename T>
static std::vector<T> readAll(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
  f.seekg(0, std::ios::end); const size_t bytes = (size_t) f.tellg(); f.seekg(0);
  std::vector<T> v(bytes / sizeof(T)); f.read((char*) v.data(), (std::streamsize) bytes);
  return v;
}
template <typename T>
static void writeAll(const std::string& path, const T* p, size_t n) { std::ofstream f(path, std::ios::binary); f.write((const char*) p, (std::streamsize) (n * sizeof(T))); }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: sweep_driver dir n_devices dev0 [dev1 ...]\n"); return 2; }
  const std::string dir = argv[1];
  const int nd = atoi(argv[2]);
  std::vector<int> devs; for (int i = 0; i < nd; ++i) devs.push_back(atoi(argv[3 + i]));
  const auto scans = readAll<float>(dir + "/scans.bin"); const auto offs = readAll<int32_t>(dir + "/offsets.bin");
  const auto map = readAll<float>(dir + "/map.bin"); const auto index = readAll<int32_t>(dir + "/index.bin"); const auto x0 = readAll<float>(dir + "/x0.bin");
  int cols = 1081, iters = 20; float tau = 0.05f, range_max = 30.f;
  { std::ifstream p(dir + "/params.txt"); p >> cols >> iters >> tau >> range_max; }
  try {
    LoopClosureSweep sweep(devs);
    PointNormal2fVectorCloud m(map.size() / 4);
    for (size_t i = 0; i < m.size(); ++i) m[i] = PointNormal2f{map[4 * i], map[4 * i + 1], map[4 * i + 2], map[4 * i + 3]};
    sweep.setMap(m);
    sweep.setScansPacked(scans.data(), offs.data(), (int) offs.size() - 1);
    PointNormal2fProjectorPolar projector;
    projector.param_canvas_cols = cols; projector.param_range_max = range_max;
    projector.param_angle_col_min = -3.14159274f; projector.param_angle_col_max = 3.14159274f;
    sweep.param_slice = LoopClosureSweep::projectiveSlice(projector, 0.5f, 0.8f, tau, 10);
    sweep.param_max_iterations = iters;
    const size_t n = index.size();
    std::vector<Vector3f> init(n);
    for (size_t i = 0; i < n; ++i) init[i] = Vector3f{{x0[3 * i], x0[3 * i + 1], x0[3 * i + 2]}};
    sweep.compute(index, init);
    std::vector<uint8_t> acc(n); for (size_t i = 0; i < n; ++i) acc[i] = sweep.accept(i) ? 1 : 0;
    writeAll(dir + "/out_pose.bin", sweep.pose[0].data(), 3 * n); writeAll(dir + "/out_H.bin", sweep.information[0].data(), 9 * n);
    writeAll(dir + "/out_status.bin", sweep.status.data(), n); writeAll(dir + "/out_stats.bin", sweep.last_stats.data(), n);
    writeAll(dir + "/out_accept.bin", acc.data(), n);
    printf("{\"devices\": %d, \"candidates\": %zu}\n", sweep.numDevices(), n);
  } catch (const std::exception& e) { fprintf(stderr, "sweep_driver: %s\n", e.what()); return 1; }
  return 0;
}
