// test_kitti_io.cpp -- command-line driver for visual-odometry-gpu_amd/host/kitti_io.hpp, used by
// tests/test_kitti_io.py (CPU only).  Test infrastructure.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "../../visual-odometry-gpu_amd/host/kitti_io.hpp"

using namespace orbx::io;

static std::vector<std::vector<double>> read_table(const std::string& path) {
  std::ifstream f(path);
  std::vector<std::vector<double>> rows;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    rows.emplace_back((std::istream_iterator<double>(ss)), std::istream_iterator<double>());
  }
  return rows;
}

int main(int argc, char** argv) {
  try {
    if (argc == 4 && !std::strcmp(argv[1], "png")) {
      const GrayImage im = read_png_gray(argv[2]);
      FILE* f = std::fopen(argv[3], "wb");
      if (!f) return 2;
      std::fwrite(im.pixels.data(), 1, im.pixels.size(), f);
      std::fclose(f);
      std::printf("%d %d\n", im.width, im.height);
      return 0;
    }
    if (argc == 4 && !std::strcmp(argv[1], "seq")) {
      const auto images = list_sequence_images(argv[2], argv[3]);
      const auto poses = read_poses(argv[2], argv[3]);
      const Mat3 K = read_calib(argv[2], argv[3]);
      std::printf("images %zu\n", images.size());
      for (const auto& s : images) std::printf("%s\n", s.c_str());
      std::printf("poses %zu\n", poses.size());
      for (const auto& T : poses) {
        for (double v : T) std::printf("%.17g ", v);
        std::printf("\n");
      }
      std::printf("K");
      for (double v : K) std::printf(" %.17g", v);
      std::printf("\n");
      return 0;
    }
    if (argc == 8 && !std::strcmp(argv[1], "paths")) {
      std::vector<Point2d> gt, est;
      std::vector<double> gs, es;
      for (const auto& r : read_table(argv[2])) gt.push_back(Point2d{r.at(0), r.at(1)});
      for (const auto& r : read_table(argv[3])) est.push_back(Point2d{r.at(0), r.at(1)});
      for (const auto& r : read_table(argv[4])) {
        gs.push_back(r.at(0));
        es.push_back(r.at(1));
      }
      save_paths(argv[5], argv[6], argv[7], gt, est, gs, es);
      return 0;
    }
  } catch (const std::exception& e) {
    std::printf("ERROR %s\n", e.what());
    return 3;
  }
  return 2;
}
