// kitti_io.hpp -- the data formats either side of the path, host code.
//
// What the reference's VO executables read and write around the ORB / matcher / LK calls:
//   * the KITTI odometry sequence directory: 8-bit gray PNG frames listed and sorted
//     (src/feature_matching.cpp:20-24), loaded with cv::imread(path, IMREAD_GRAYSCALE)
//     (src/feature_matching.cpp:60; src/feature_tracking.cpp:59,66);
//   * ground-truth poses, one 3x4 row-major matrix per line -> 4x4 (readPoses,
//     src/feature_matching.cpp:126-140) and the calibration file's first line "P0: ..."
//     -> K = P[:, :3] (readCalib, src/feature_matching.cpp:142-153);
//   * the trajectory files gt_path.txt / est_path.txt ("x z" per line) and scale.txt
//     ("gt est" per line) that metric.py:49-51 loads back (savePaths,
//     src/feature_matching.cpp:295-322): plain `ostream << double`, 6 significant digits.
// OpenCV (imread) is absent from the image this was written in, so the PNG reader is written
// against the PNG specification directly on top of zlib: non-interlaced, 8-bit gray (the KITTI
// format), gray+alpha, RGB and RGBA (converted with cv::cvtColor's fixed-point BT.601 weights).
// PARITY UNPINNED for colour PNGs: cv::imread(IMREAD_GRAYSCALE) lets libpng convert (png_set_rgb_to_gray),
// whose coefficients and rounding differ from cvtColor's in the last bit; KITTI frames are 8-bit gray
// and are not affected (pinned by the reference's own 000000.png, tests/test_kitti_io.py).
// Link with -lz.  Nothing here touches the GPU; frames go to liborbx as orbx::Image.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace orbx {
namespace io {

struct GrayImage {
  int width = 0, height = 0;
  std::vector<uint8_t> pixels;  // tight rows (stride == width)
};

namespace detail {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace detail

// cv::imread(path, cv::IMREAD_GRAYSCALE) for PNG files
inline GrayImage read_png_gray(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("read_png_gray: cannot open " + path);
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) throw std::runtime_error("read_png_gray: not a PNG: " + path);
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat;
  bool end = false;
  while (!end && pos + 12 <= buf.size()) {
    const uint32_t len = detail::be32(&buf[pos]);
    const uint8_t* type = &buf[pos + 4];
    if (pos + 12 + (size_t)len > buf.size()) throw std::runtime_error("read_png_gray: truncated chunk in " + path);
    const uint8_t* data = &buf[pos + 8];
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != detail::be32(data + len))
      throw std::runtime_error("read_png_gray: CRC mismatch in " + path);
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error("read_png_gray: bad IHDR");
      w = detail::be32(data);
      h = detail::be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w == 0 || h == 0 || w > (1u << 20) || h > (1u << 20)) throw std::runtime_error("read_png_gray: bad header in " + path);
  if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 4 || ctype == 6))
    throw std::runtime_error("read_png_gray: only non-interlaced 8-bit gray / gray+alpha / RGB / RGBA PNGs are supported: " + path);
  const int bpp = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : 4;
  const size_t stride = (size_t)w * bpp;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error("read_png_gray: zlib stream is corrupt in " + path);
  // undo the per-row filters (PNG spec 9.2) in place
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  GrayImage img;
  img.width = (int)w;
  img.height = (int)h;
  img.pixels.resize((size_t)w * h);
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += detail::paeth(a, b, c); break;
        default: throw std::runtime_error("read_png_gray: bad filter type in " + path);
      }
      cur[i] = (uint8_t)v;
    }
    uint8_t* dst = &img.pixels[(size_t)y * w];
    if (bpp <= 2) {
      for (uint32_t x = 0; x < w; x++) dst[x] = cur[(size_t)x * bpp];
    } else {  // cv::cvtColor RGB2GRAY for 8U: (R*4899 + G*9617 + B*1868 + 8192) >> 14
      for (uint32_t x = 0; x < w; x++) {
        const uint8_t* p = &cur[(size_t)x * bpp];
        dst[x] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
      }
    }
    prev.swap(cur);
  }
  return img;
}

// the sorted frame list of a sequence (VisualOdom constructor, src/feature_matching.cpp:20-24)
inline std::vector<std::string> list_sequence_images(const std::string& kitti_dir, const std::string& seq) {
  const std::string dir = kitti_dir + "/data_odometry_gray/dataset/sequences/" + seq + "/image_0";
  std::vector<std::string> images;
  for (auto& p : std::filesystem::directory_iterator(dir)) images.push_back(p.path().string());
  std::sort(images.begin(), images.end());
  return images;
}

using Mat4 = std::array<double, 16>;  // row-major 4x4
using Mat3 = std::array<double, 9>;   // row-major 3x3

// readPoses (src/feature_matching.cpp:126-140): one 3x4 row-major pose per line -> 4x4, last row (0,0,0,1)
inline std::vector<Mat4> read_poses_file(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("read_poses: cannot open " + path);
  std::vector<Mat4> poses;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::vector<double> d((std::istream_iterator<double>(ss)), std::istream_iterator<double>());
    if (d.size() < 12) throw std::runtime_error("read_poses: a line holds fewer than 12 numbers in " + path);
    Mat4 T{};
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T[r * 4 + c] = d[r * 4 + c];
    T[15] = 1.0;
    poses.push_back(T);
  }
  return poses;
}
inline std::vector<Mat4> read_poses(const std::string& kitti_dir, const std::string& seq) {
  return read_poses_file(kitti_dir + "/data_odometry_poses/dataset/poses/" + seq + ".txt");
}

// readCalib (src/feature_matching.cpp:142-153): first line "P0: p00 ... p23" -> K = P[:, :3]
inline Mat3 read_calib_file(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("read_calib: cannot open " + path);
  std::string line;
  std::getline(f, line);
  if (line.size() < 4) throw std::runtime_error("read_calib: empty first line in " + path);
  std::istringstream ss(line.substr(4));
  std::vector<double> p((std::istream_iterator<double>(ss)), std::istream_iterator<double>());
  if (p.size() < 12) throw std::runtime_error("read_calib: the first line holds fewer than 12 numbers in " + path);
  Mat3 K{};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) K[r * 3 + c] = p[r * 4 + c];
  return K;
}
inline Mat3 read_calib(const std::string& kitti_dir, const std::string& seq) {
  return read_calib_file(kitti_dir + "/data_odometry_gray/dataset/sequences/" + seq + "/calib.txt");
}

struct Point2d {  // cv::Point2d
  double x = 0, y = 0;
};

// savePaths (src/feature_matching.cpp:295-322): "x y\n" per path point, "gt est\n" per scale
// pair, formatted by a default-constructed ofstream (6 significant digits)
inline void save_paths(const std::string& gt_file, const std::string& est_file, const std::string& scale_file,
                       const std::vector<Point2d>& gt_path, const std::vector<Point2d>& est_path,
                       const std::vector<double>& gt_scale, const std::vector<double>& est_scale) {
  std::ofstream gt_out(gt_file), est_out(est_file), scale_out(scale_file);
  if (!gt_out || !est_out || !scale_out) throw std::runtime_error("save_paths: cannot open an output file");
  for (size_t i = 0; i < gt_path.size(); i++) gt_out << gt_path[i].x << " " << gt_path[i].y << "\n";
  for (size_t i = 0; i < est_path.size(); i++) est_out << est_path[i].x << " " << est_path[i].y << "\n";
  for (size_t i = 0; i < gt_scale.size() && i < est_scale.size(); i++) scale_out << gt_scale[i] << " " << est_scale[i] << "\n";
}

}  // namespace io
}  // namespace orbx
