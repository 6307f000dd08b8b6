// orb.hpp -- C++ host-side mirror of the reference's ORB interface, on top of
// the liborbx C ABI (include/orbx.h).  Header-only, C++17, no OpenCV needed.
//
// Same class / function names, constructor defaults, argument meaning and
// ownership as the reference, so code written against the reference headers
// keeps compiling after `#include "orb.hpp"` is pointed here:
//
//   reference                                   this header
//   ------------------------------------------  ---------------------------------
//   include/orb.hpp:4      struct Keypoint       Keypoint (layout == orbx_keypoint)
//   include/orb.hpp:6-8    struct ORBDescriptor  ORBDescriptor (== orbx_descriptor)
//   include/orb.hpp:10-22  class OrientedFAST    OrientedFAST
//   include/orb.hpp:24-32  class RotatedBRIEF    RotatedBRIEF
//   include/orb.hpp:34-49  class ORB             ORB
//   include/orb_cpu.hpp    *CPU twins            OrientedFASTCPU / RotatedBRIEFCPU / ORBCPU
//                                                (same GPU kernels, CPU-flavour semantics)
//   include/Fast.cuh:5-6   Fast, Orientations    Fast, Orientations
//   include/NMS.cuh:5      NMS                   NMS
//   include/HarrisScore.cuh:5  HarrisScore       HarrisScore
//   include/Brief.cuh:5    Brief                 Brief
//   include/Convolution.cuh:5  conv2d            conv2d
//   include/GaussianBlur.cuh:3-4  GaussianBlur, GaussianBlur1D
//   include/GaussianBlur.hpp:6  GaussianBlurCUDA GaussianBlurCUDA
//   include/Sobel.hpp:6    SobelCUDA             SobelCUDA
//
// Differences, all deliberate:
//   * images are passed as orbx::Image {data,width,height,stride}; a cv::Mat
//     overload set is enabled with -DORBX_WITH_OPENCV when OpenCV exists;
//   * ORB::detectAndCompute ASSIGNS its outputs like ORBCPU does
//     (orb_cpu.cpp:272-275) instead of appending (orb.cpp:100-102, D12);
//   * errors throw std::runtime_error (the reference prints and exit(1)s,
//     Fast.cu:8-18) and constructors print nothing;
//   * HarrisScore takes `float k` (the reference's `int k` truncates 0.04 to 0, D7).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/orbx.h"

#ifdef ORBX_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

struct Keypoint {
  int x, y;
};
struct ORBDescriptor {
  uint8_t data[32];
};
static_assert(sizeof(Keypoint) == sizeof(orbx_keypoint), "Keypoint layout");
static_assert(sizeof(ORBDescriptor) == sizeof(orbx_descriptor), "ORBDescriptor layout");

namespace orbx {

// 8-bit single-channel image view (what the reference passes as CV_8UC1 cv::Mat)
struct Image {
  const uint8_t* data = nullptr;
  int width = 0, height = 0, stride = 0;
  Image() = default;
  Image(const uint8_t* d, int w, int h, int s = 0) : data(d), width(w), height(h), stride(s ? s : w) {}
#ifdef ORBX_WITH_OPENCV
  Image(const cv::Mat& m) : data(m.data), width(m.cols), height(m.rows), stride((int)m.step) {  // NOLINT
    if (m.type() != CV_8UC1) throw std::runtime_error("orbx: image must be CV_8UC1");  // CV_Assert, orb_cpu.cpp:26
  }
#endif
};

// owned 8-bit image (what the reference returns in a cv::Mat `dst`)
struct Image8 {
  std::vector<uint8_t> pixels;
  int width = 0, height = 0;
  Image view() const { return Image(pixels.data(), width, height, width); }
};

namespace detail {

inline void check(orbx_ctx* c, int st, const char* what) {
  if (st != ORBX_OK)
    throw std::runtime_error(std::string(what) + ": " + orbx_status_string(st) + ": " + orbx_last_error_string(c));
}

// A context that grows with the largest image it has seen (the reference
// allocates per call; here device memory is owned by the object).
class Ctx {
 public:
  explicit Ctx(const orbx_params& p) : p_(p) {}
  ~Ctx() { orbx_destroy(c_); }
  Ctx(const Ctx&) = delete;
  Ctx& operator=(const Ctx&) = delete;
  orbx_ctx* get(int w, int h) {
    if (!c_ || w > p_.max_width || h > p_.max_height) {
      orbx_destroy(c_);
      c_ = nullptr;
      p_.max_width = w > p_.max_width ? w : p_.max_width;
      p_.max_height = h > p_.max_height ? h : p_.max_height;
      check(nullptr, orbx_create(&p_, &c_), "orbx_create");
    }
    return c_;
  }
  const orbx_params& params() const { return p_; }

 private:
  orbx_params p_;
  orbx_ctx* c_ = nullptr;
};

inline orbx_params gpu_defaults() {
  orbx_params p;
  orbx_params_default_gpu(&p);
  p.max_width = 8;
  p.max_height = 8;
  return p;
}
inline orbx_params cpu_defaults() {
  orbx_params p;
  orbx_params_default_cpu(&p);
  p.max_width = 8;
  p.max_height = 8;
  return p;
}
inline std::shared_ptr<Ctx>& stage_ctx() {  // shared by the free stage functions, the matcher and the tracker
  static std::shared_ptr<Ctx> c = [] {
    orbx_params p = gpu_defaults();
    p.nlevels = 1;  // the stage operators work on single images: any size >= 8x8 is a valid plan
    return std::make_shared<Ctx>(p);
  }();
  return c;
}
inline orbx_keypoint* kp(std::vector<Keypoint>& v) { return reinterpret_cast<orbx_keypoint*>(v.data()); }
inline const orbx_keypoint* kp(const std::vector<Keypoint>& v) {
  return reinterpret_cast<const orbx_keypoint*>(v.data());
}
inline orbx_descriptor* ds(std::vector<ORBDescriptor>& v) { return reinterpret_cast<orbx_descriptor*>(v.data()); }

}  // namespace detail
}  // namespace orbx

// ---- free stage functions (reference: include/*.cuh, *.hpp) -----------------

// Fast.cuh:5 -- returns the keypoint count; `keypoints` is resized to it.
inline int Fast(const orbx::Image& image, std::vector<Keypoint>& keypoints, int threshold, int n, int nms_window,
                int nfeatures) {
  orbx_ctx* c = orbx::detail::stage_ctx()->get(image.width, image.height);
  keypoints.resize(nfeatures > 0 ? nfeatures : 0);
  int count = 0;
  orbx::detail::check(c,
                      orbx_fast(c, image.data, image.width, image.height, image.stride, threshold, n, nms_window,
                                nfeatures, orbx::detail::kp(keypoints), &count, nullptr),
                      "Fast");
  keypoints.resize(count);
  return count;
}

// Fast.cuh:6
inline void Orientations(const orbx::Image& image, const std::vector<Keypoint>& keypoints,
                         std::vector<float>& orientations, int patch_size) {
  orbx_ctx* c = orbx::detail::stage_ctx()->get(image.width, image.height);
  orientations.assign(keypoints.size(), 0.0f);
  orbx::detail::check(c,
                      orbx_orientations(c, image.data, image.width, image.height, image.stride,
                                        orbx::detail::kp(keypoints), (int)keypoints.size(), patch_size,
                                        orientations.data()),
                      "Orientations");
}

// NMS.cuh:5 -- `scores` is a width*height float map (CV_32F in the reference)
inline void NMS(const float* scores, int width, int height, std::vector<Keypoint>& keypoints, int nms_window,
                int nfeatures, float threshold) {
  orbx_ctx* c = orbx::detail::stage_ctx()->get(width < 8 ? 8 : width, height < 8 ? 8 : height);
  keypoints.resize(nfeatures > 0 ? nfeatures : 0);
  int count = 0;
  orbx::detail::check(
      c, orbx_nms(c, scores, width, height, nms_window, nfeatures, threshold, orbx::detail::kp(keypoints), &count, nullptr),
      "NMS");
  keypoints.resize(count);
}

// HarrisScore.cuh:5
inline void HarrisScore(const orbx::Image& image, std::vector<Keypoint>& keypoints, std::vector<float>& harris_scores,
                        int corner_window, float k) {
  orbx_ctx* c = orbx::detail::stage_ctx()->get(image.width, image.height);
  harris_scores.assign(keypoints.size(), 0.0f);
  orbx::detail::check(c,
                      orbx_harris(c, image.data, image.width, image.height, image.stride, orbx::detail::kp(keypoints),
                                  (int)keypoints.size(), corner_window, k, harris_scores.data()),
                      "HarrisScore");
}

// Brief.cuh:5 -- n_bits must be 256 and patch_size 31, as in the reference
inline void Brief(const orbx::Image& image, const std::vector<Keypoint>& keypoints,
                  const std::vector<float>& orientations, std::vector<ORBDescriptor>& descriptors, int n_bits = 256,
                  int patch_size = 31) {
  if (n_bits != 256 || patch_size != 31) throw std::runtime_error("Brief: only n_bits=256, patch_size=31 exist");
  if (orientations.size() != keypoints.size()) throw std::runtime_error("Brief: size mismatch");
  orbx_ctx* c = orbx::detail::stage_ctx()->get(image.width, image.height);
  descriptors.assign(keypoints.size(), ORBDescriptor{});
  orbx::detail::check(c,
                      orbx_brief(c, image.data, image.width, image.height, image.stride, orbx::detail::kp(keypoints),
                                 orientations.data(), (int)keypoints.size(), orbx::detail::ds(descriptors)),
                      "Brief");
}

// Convolution.cuh:5 -- `image` is pre-padded; dst is (h-K+1) x (w-K+1)
inline void conv2d(const orbx::Image& image, orbx::Image8& dst, const float* kernel, int kernel_size) {
  orbx_ctx* c = orbx::detail::stage_ctx()->get(image.width, image.height);
  dst.width = image.width - kernel_size + 1;
  dst.height = image.height - kernel_size + 1;
  if (dst.width < 1 || dst.height < 1) throw std::runtime_error("conv2d: image smaller than kernel");
  dst.pixels.assign((size_t)dst.width * dst.height, 0);
  orbx::detail::check(
      c, orbx_conv2d(c, image.data, image.width, image.height, image.stride, kernel, kernel_size, dst.pixels.data()),
      "conv2d");
}

namespace orbx::detail {
template <class F>
inline void same_size_stage(const Image& image, Image8& dst, const char* what, F&& call) {
  orbx_ctx* c = stage_ctx()->get(image.width, image.height);
  dst.width = image.width;
  dst.height = image.height;
  dst.pixels.assign((size_t)dst.width * dst.height, 0);
  check(c, call(c), what);
}
}  // namespace orbx::detail

// GaussianBlur.cuh:3 (5x5 /273)
inline void GaussianBlur(const orbx::Image& image, orbx::Image8& dst) {
  orbx::detail::same_size_stage(image, dst, "GaussianBlur", [&](orbx_ctx* c) {
    return orbx_blur5_273(c, image.data, image.width, image.height, image.stride, dst.pixels.data(), dst.width);
  });
}
// GaussianBlur.cuh:4 (separable [1 4 6 4 1]/16)
inline void GaussianBlur1D(const orbx::Image& image, orbx::Image8& dst) {
  orbx::detail::same_size_stage(image, dst, "GaussianBlur1D", [&](orbx_ctx* c) {
    return orbx_blur5_sep(c, image.data, image.width, image.height, image.stride, dst.pixels.data(), dst.width);
  });
}
// GaussianBlur.hpp:6
inline void GaussianBlurCUDA(const orbx::Image& image, orbx::Image8& dst, int kernel_size) {
  orbx::detail::same_size_stage(image, dst, "GaussianBlurCUDA", [&](orbx_ctx* c) {
    return orbx_gaussian_blur_conv(c, image.data, image.width, image.height, image.stride, kernel_size,
                                   dst.pixels.data());
  });
}
// Sobel.hpp:6
inline void SobelCUDA(const orbx::Image& image, orbx::Image8& dst, int dir) {
  orbx::detail::same_size_stage(image, dst, "SobelCUDA", [&](orbx_ctx* c) {
    return orbx_sobel(c, image.data, image.width, image.height, image.stride, dir, dst.pixels.data());
  });
}

// ---- classes (reference: include/orb.hpp, include/orb_cpu.hpp) ---------------

class OrientedFAST {
 public:
  OrientedFAST(int threshold = 20, int n = 9, int nms_window = 3, int patch_size = 31)
      : threshold(threshold), n(n), nms_window(nms_window), patch_size(patch_size) {}
  // orb.cpp:22-27
  std::vector<Keypoint> detect(const orbx::Image& image, int nfeatures) {
    std::vector<Keypoint> keypoints;
    Fast(image, keypoints, threshold, n, nms_window, nfeatures);
    return keypoints;
  }
  // orb.cpp:29-33
  std::vector<float> compute_orientations(const orbx::Image& image, const std::vector<Keypoint>& keypoints) {
    std::vector<float> orientations;
    Orientations(image, keypoints, orientations, patch_size);
    return orientations;
  }

 private:
  int threshold, n, nms_window, patch_size;
};

class RotatedBRIEF {
 public:
  RotatedBRIEF() = default;
  // orb.cpp:40-44
  std::vector<ORBDescriptor> compute(const orbx::Image& image, const std::vector<Keypoint>& keypoints,
                                     const std::vector<float>& orientations) {
    std::vector<ORBDescriptor> descriptors;
    Brief(image, keypoints, orientations, descriptors, n_bits, patch_size);
    return descriptors;
  }

 private:
  int n_bits = 256;
  int patch_size = 31;
};

class ORB {
 public:
  // orb.hpp:36; the FAST/Harris knobs the reference hard-codes are exposed through params()
  ORB(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8) : ctx_(make(nfeatures, scaleFactor, nlevels)) {}
  explicit ORB(const orbx_params& p) : ctx_(std::make_shared<orbx::detail::Ctx>(p)) {}

  // orb.hpp:37 / orb.cpp:58-109.  Outputs are assigned (not appended).
  void detectAndCompute(const orbx::Image& image, std::vector<Keypoint>& keypoints, std::vector<float>& orientations,
                        std::vector<ORBDescriptor>& descriptors) {
    detectAndCompute(image, keypoints, orientations, descriptors, nullptr, nullptr);
  }
  // extended form: Harris responses and pyramid level per keypoint
  void detectAndCompute(const orbx::Image& image, std::vector<Keypoint>& keypoints, std::vector<float>& orientations,
                        std::vector<ORBDescriptor>& descriptors, std::vector<float>* responses,
                        std::vector<int32_t>* levels) {
    orbx_ctx* c = ctx_->get(image.width, image.height);
    int32_t cap = 0;
    orbx::detail::check(c, orbx_get_plan(c, image.width, image.height, nullptr, nullptr, nullptr, nullptr, nullptr, &cap),
                        "orbx_get_plan");
    if (cap < 1) cap = 1;
    keypoints.resize(cap);
    orientations.resize(cap);
    descriptors.resize(cap);
    if (responses) responses->resize(cap);
    if (levels) levels->resize(cap);
    int count = 0;
    orbx::detail::check(c,
                        orbx_detect_and_compute(c, image.data, image.width, image.height, image.stride,
                                                orbx::detail::kp(keypoints), orientations.data(),
                                                orbx::detail::ds(descriptors), responses ? responses->data() : nullptr,
                                                levels ? levels->data() : nullptr, nullptr, cap, &count),
                        "ORB::detectAndCompute");
    keypoints.resize(count);
    orientations.resize(count);
    descriptors.resize(count);
    if (responses) responses->resize(count);
    if (levels) levels->resize(count);
  }
  const orbx_params& params() const { return ctx_->params(); }

 private:
  static std::shared_ptr<orbx::detail::Ctx> make(int nfeatures, float sf, int nlevels) {
    orbx_params p = orbx::detail::gpu_defaults();
    p.nfeatures = nfeatures;
    p.scale_factor = sf;
    p.nlevels = nlevels;
    return std::make_shared<orbx::detail::Ctx>(p);
  }
  std::shared_ptr<orbx::detail::Ctx> ctx_;
};

// ---- CPU-flavour twins (include/orb_cpu.hpp): same kernels, CPU semantics ----

class OrientedFASTCPU {
 public:
  OrientedFASTCPU(int nfeatures = 3000, int threshold = 50, int n = 9, int nms_window = 3, int patch_size = 9)
      : nfeatures(nfeatures), threshold(threshold), n(n), nms_window(nms_window), patch_size(patch_size) {}
  std::vector<Keypoint> detect(const orbx::Image& image) {  // orb_cpu.cpp:23-137
    std::vector<Keypoint> keypoints;
    Fast(image, keypoints, threshold, n, nms_window, nfeatures);
    return keypoints;
  }
  std::vector<float> compute_orientations(const orbx::Image& image, const std::vector<Keypoint>& keypoints) {
    std::vector<float> o;  // orb_cpu.cpp:139-183
    Orientations(image, keypoints, o, patch_size);
    return o;
  }

 private:
  int nfeatures, threshold, n, nms_window, patch_size;
};

using RotatedBRIEFCPU = RotatedBRIEF;  // orb_cpu.cpp:185-258: identical arithmetic

class ORBCPU {
 public:
  // orb_cpu.hpp:30; like the reference, nfeatures/scaleFactor/nlevels are
  // accepted and ignored (orb_cpu.cpp:271-276 runs one level with the
  // OrientedFASTCPU defaults, D16)
  ORBCPU(int = 500, float = 1.2f, int = 8) : orb_(orbx::detail::cpu_defaults()) {}
  void detectAndCompute(const orbx::Image& image, std::vector<Keypoint>& keypoints, std::vector<float>& orientations,
                        std::vector<ORBDescriptor>& descriptors) {
    orb_.detectAndCompute(image, keypoints, orientations, descriptors);
  }

 private:
  ORB orb_;
};

// ---- descriptor matching (next row: src/feature_matching.cpp:166-181) ---------
// Call shape of cv::DescriptorMatcher::knnMatch(des1, des2, matches, 2) as used by
// VisualOdom::get_matches; exact brute-force Hamming instead of FLANN's LSH.

struct DMatch {  // cv::DMatch
  int queryIdx = -1, trainIdx = -1;
  float distance = 0.f;
};

class HammingMatcher {
 public:
  HammingMatcher() : ctx_(orbx::detail::stage_ctx()) {}
  // matches[i] holds up to k (= 2) neighbours of des1[i], best first
  void knnMatch(const std::vector<ORBDescriptor>& des1, const std::vector<ORBDescriptor>& des2,
                std::vector<std::vector<DMatch>>& matches, int k = 2) {
    if (k != 2) throw std::runtime_error("HammingMatcher::knnMatch: only k = 2 (the reference's call) exists");
    orbx_ctx* c = ctx_->get(8, 8);
    std::vector<int32_t> idx(2 * des1.size()), dist(2 * des1.size());
    orbx::detail::check(c,
                        orbx_knn2(c, reinterpret_cast<const orbx_descriptor*>(des1.data()), (int)des1.size(),
                                  reinterpret_cast<const orbx_descriptor*>(des2.data()), (int)des2.size(), idx.data(),
                                  dist.data()),
                        "knnMatch");
    matches.assign(des1.size(), {});
    for (size_t i = 0; i < des1.size(); i++)
      for (int j = 0; j < 2; j++)
        if (idx[2 * i + j] >= 0) matches[i].push_back(DMatch{(int)i, idx[2 * i + j], (float)dist[2 * i + j]});
  }
  // knnMatch + `m.distance < ratio * n.distance` (feature_matching.cpp:172-181), in one device pass
  std::vector<DMatch> ratioMatch(const std::vector<ORBDescriptor>& des1, const std::vector<ORBDescriptor>& des2,
                                 double ratio = 0.8) {
    orbx_ctx* c = ctx_->get(8, 8);
    std::vector<int32_t> qi(des1.size()), ti(des1.size()), d1(des1.size());
    int n = 0;
    orbx::detail::check(c,
                        orbx_match_ratio(c, reinterpret_cast<const orbx_descriptor*>(des1.data()), (int)des1.size(),
                                         reinterpret_cast<const orbx_descriptor*>(des2.data()), (int)des2.size(), ratio,
                                         qi.data(), ti.data(), d1.data(), (int)des1.size(), &n),
                        "ratioMatch");
    std::vector<DMatch> out(n);
    for (int i = 0; i < n; i++) out[i] = DMatch{qi[i], ti[i], (float)d1[i]};
    return out;
  }

 private:
  std::shared_ptr<orbx::detail::Ctx> ctx_;
};

// ---- cv::Feature2D-shaped adapter (next row) ---------------
// What the VO executables hold is a `cv::Ptr<cv::Feature2D>` (`cv::ORB::create(3000)`,
// src/feature_tracking.cpp:31) on which they call `detectAndCompute(img, cv::noArray(),
// kp, des)` (src/feature_matching.cpp:164, src/feature_tracking.cpp:201) and `detect(img,
// kp)` (src/feature_tracking.cpp:61), and of whose results they read `kp.pt` and an
// N x 32 CV_8U descriptor matrix (src/feature_matching.cpp:179-180).  orbx::Feature2D gives
// this front-end that call shape; orbx::KeyPoint carries the cv::KeyPoint fields with the
// conventions cv::ORB uses for them (size = patch size x level scale, angle in degrees in
// [0, 360), response = Harris response, octave = pyramid level).  With -DORBX_WITH_OPENCV
// the cv::KeyPoint / cv::Mat overloads make it a literal replacement.
namespace orbx {

struct Point2f {  // cv::Point2f
  float x = 0.f, y = 0.f;
};

struct KeyPoint {  // cv::KeyPoint
  Point2f pt;
  float size = 0.f;
  float angle = -1.f;  // degrees, [0, 360); -1 = not computed (detect())
  float response = 0.f;
  int octave = 0;
  int class_id = -1;
  // cv::KeyPoint::convert(kp1, pts1), src/feature_tracking.cpp:62
  static void convert(const std::vector<KeyPoint>& keypoints, std::vector<Point2f>& points2f) {
    points2f.resize(keypoints.size());
    for (size_t i = 0; i < keypoints.size(); i++) points2f[i] = keypoints[i].pt;
  }
};

// N x 32 CV_8U descriptor matrix (row i = descriptor of keypoint i)
struct DescriptorMat {
  int rows = 0;
  static constexpr int cols = 32;
  std::vector<ORBDescriptor> d;
  const uint8_t* ptr(int r) const { return d[(size_t)r].data; }
  const uint8_t* data() const { return d.empty() ? nullptr : d[0].data; }  // rows x 32 contiguous bytes
  bool empty() const { return rows == 0; }
};

inline float angle_degrees(float radians) {  // atan2f's (-pi, pi] -> cv::KeyPoint's [0, 360)
  float a = radians * 57.29577951308232f;
  if (a < 0.f) a += 360.f;
  if (a >= 360.f) a -= 360.f;
  return a;
}

class Feature2D {
 public:
  // cv::ORB::create(nfeatures, scaleFactor, nlevels), src/feature_tracking.cpp:31
  static std::shared_ptr<Feature2D> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8) {
    return std::shared_ptr<Feature2D>(new Feature2D(::ORB(nfeatures, scaleFactor, nlevels)));
  }
  static std::shared_ptr<Feature2D> create(const orbx_params& p) {
    return std::shared_ptr<Feature2D>(new Feature2D(::ORB(p)));
  }
  // orb->detectAndCompute(img, cv::noArray(), kp, des); a mask is not supported (the
  // reference never passes one)
  void detectAndCompute(const Image& image, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors) {
    run(image, keypoints, &descriptors);
  }
  // orb->detect(img, kp)
  void detect(const Image& image, std::vector<KeyPoint>& keypoints) { run(image, keypoints, nullptr); }
  int descriptorSize() const { return 32; }
  const orbx_params& params() const { return orb_.params(); }

#ifdef ORBX_WITH_OPENCV
  void detectAndCompute(const cv::Mat& image, cv::InputArray /*mask = cv::noArray()*/,
                        std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors) {
    std::vector<KeyPoint> k;
    DescriptorMat d;
    run(Image(image), k, &d);
    to_cv(k, keypoints);
    descriptors.create((int)k.size(), 32, CV_8U);
    if (!k.empty()) std::memcpy(descriptors.data, d.data(), k.size() * 32);
  }
  void detect(const cv::Mat& image, std::vector<cv::KeyPoint>& keypoints) {
    std::vector<KeyPoint> k;
    run(Image(image), k, nullptr);
    to_cv(k, keypoints);
  }
#endif

 private:
  explicit Feature2D(::ORB orb) : orb_(std::move(orb)) {}
  void run(const Image& image, std::vector<KeyPoint>& keypoints, DescriptorMat* descriptors) {
    orb_.detectAndCompute(image, kps_, angles_, desc_, &resp_, &levels_);
    const orbx_params& p = orb_.params();
    keypoints.resize(kps_.size());
    for (size_t i = 0; i < kps_.size(); i++) {
      KeyPoint& k = keypoints[i];
      k.pt.x = (float)kps_[i].x;
      k.pt.y = (float)kps_[i].y;
      k.octave = levels_[i];
      k.size = (float)p.patch_size * (float)std::pow((double)p.scale_factor, (double)levels_[i]);
      k.angle = descriptors ? angle_degrees(angles_[i]) : -1.f;
      k.response = resp_[i];
      k.class_id = -1;
    }
    if (descriptors) {
      descriptors->rows = (int)desc_.size();
      descriptors->d = desc_;
    }
  }
#ifdef ORBX_WITH_OPENCV
  static void to_cv(const std::vector<KeyPoint>& k, std::vector<cv::KeyPoint>& out) {
    out.resize(k.size());
    for (size_t i = 0; i < k.size(); i++)
      out[i] = cv::KeyPoint(k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
  }
#endif
  ::ORB orb_;
  std::vector<Keypoint> kps_;
  std::vector<float> angles_, resp_;
  std::vector<ORBDescriptor> desc_;
  std::vector<int32_t> levels_;
};

// VisualOdom::get_matches (src/feature_matching.cpp:155-183): detect + describe frame 2,
// 2-NN match frame 1 -> 2, keep `m.distance < 0.8 * n.distance`, return the matched points.
inline void get_matches(Feature2D& orb, HammingMatcher& matcher, const std::vector<KeyPoint>& kp1,
                        const DescriptorMat& des1, const Image& img2, std::vector<KeyPoint>& kp2, DescriptorMat& des2,
                        std::vector<Point2f>& pts1, std::vector<Point2f>& pts2) {
  orb.detectAndCompute(img2, kp2, des2);
  std::vector<std::vector<DMatch>> matches;
  matcher.knnMatch(des1.d, des2.d, matches, 2);
  pts1.clear();
  pts2.clear();
  for (size_t i = 0; i < matches.size(); i++) {
    if (matches[i].size() < 2) continue;
    const DMatch& m = matches[i][0];
    const DMatch& n = matches[i][1];
    if (m.distance < 0.8 * n.distance) {
      pts1.push_back(kp1[(size_t)m.queryIdx].pt);
      pts2.push_back(kp2[(size_t)m.trainIdx].pt);
    }
  }
}

// ---- pyramidal Lucas-Kanade tracking (next row) ---------------
// cv::calcOpticalFlowPyrLK(img1, img2, pts1, pts2, status, err, cv::Size(21,21), 3,
//     cv::TermCriteria(COUNT + EPS, 30, 0.01))                 src/feature_tracking.cpp:175-181
struct Size {  // cv::Size
  int width = 21, height = 21;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
};
struct TermCriteria {  // cv::TermCriteria(COUNT + EPS, maxCount, epsilon)
  int maxCount = 30;
  double epsilon = 0.01;
  TermCriteria() = default;
  TermCriteria(int count, double eps) : maxCount(count), epsilon(eps) {}
};

class LKTracker {
 public:
  // the tracker owns its context: the pyramid of the previous frame is cached in it, and a free stage
  // function called on a larger image re-creates the shared stage context (which would drop that cache)
  LKTracker() : ctx_([] {
    orbx_params p = detail::gpu_defaults();
    p.nlevels = 1;
    return std::make_shared<detail::Ctx>(p);
  }()) {}
  // prevImg == nullptr: the previous call's nextImg is this call's prevImg (its pyramid is still on the
  // device): the `img1 = img2.clone()` of the reference's loop, src/feature_tracking.cpp:112
  void calcOpticalFlowPyrLK(const Image* prevImg, const Image& nextImg, const std::vector<Point2f>& prevPts,
                            std::vector<Point2f>& nextPts, std::vector<uint8_t>& status, std::vector<float>& err,
                            Size winSize = Size(21, 21), int maxLevel = 3, TermCriteria criteria = TermCriteria()) {
    if (winSize.width != winSize.height) throw std::runtime_error("calcOpticalFlowPyrLK: square windows only");
    if (prevImg && (prevImg->width != nextImg.width || prevImg->height != nextImg.height))
      throw std::runtime_error("calcOpticalFlowPyrLK: image sizes differ");
    orbx_ctx* c = ctx_->get(nextImg.width, nextImg.height);
    const int n = (int)prevPts.size();
    nextPts.resize(prevPts.size());
    status.resize(prevPts.size());
    err.resize(prevPts.size());
    detail::check(c,
                  orbx_lk_track(c, prevImg ? prevImg->data : nullptr, prevImg ? prevImg->stride : 0, nextImg.data,
                                nextImg.stride, nextImg.width, nextImg.height,
                                reinterpret_cast<const float*>(prevPts.data()), n,
                                reinterpret_cast<float*>(nextPts.data()), status.data(), err.data(), winSize.width,
                                maxLevel, criteria.maxCount, criteria.epsilon),
                  "calcOpticalFlowPyrLK");
  }

 private:
  std::shared_ptr<detail::Ctx> ctx_;
};

// VisualOdom::track_optical_flow (src/feature_tracking.cpp:166-193): track pts1 into img2, drop lost tracks
inline void track_optical_flow(LKTracker& lk, const Image* img1, const Image& img2, std::vector<Point2f>& pts1,
                               std::vector<Point2f>& pts2) {
  std::vector<uint8_t> status;
  std::vector<float> err;
  // also with no points: the call uploads img2, so that the tracker's cached "previous frame" stays
  // in step with the caller's loop (img1 == nullptr on the next call means THIS img2)
  lk.calcOpticalFlowPyrLK(img1, img2, pts1, pts2, status, err, Size(21, 21), 3, TermCriteria(30, 0.01));
  if (pts1.empty()) return;
  std::vector<Point2f> v1, v2;
  for (size_t i = 0; i < status.size(); i++)
    if (status[i]) {
      v1.push_back(pts1[i]);
      v2.push_back(pts2[i]);
    }
  pts1 = v1;
  pts2 = v2;
}

}  // namespace orbx
