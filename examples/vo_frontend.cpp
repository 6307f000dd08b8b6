// vo_frontend.cpp -- the image-side half of the reference's tracking VO loop
// (VisualOdom::run, src/feature_tracking.cpp:44-126) on liborbx: everything up to the point
// where the reference hands the correspondences to cv::findEssentialMat (get_pose, out of scope).
//
//   frame 0 : imread -> orb->detect -> KeyPoint::convert                    (:56-62)
//   frame i : imread -> track_optical_flow (pyramidal LK, drop lost tracks)  (:64-67, :166-193)
//             fewer than 150 tracks -> get_matches (ORB + 2-NN + ratio test)  (:70-72, :195-220)
//             img1 = img2; pts1 = pts2                                        (:112-113)
//
// Usage: vo_frontend <kitti_dir> <seq> [max_frames] [nfeatures]
//   reads <kitti_dir>/data_odometry_gray/dataset/sequences/<seq>/image_0/*.png and prints one line
//   per frame: index, correspondences handed to the pose stage, how they were obtained, median flow.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../visual-odometry-gpu_amd/host/kitti_io.hpp"
#include "../visual-odometry-gpu_amd/host/orb.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <kitti_dir> <seq> [max_frames] [nfeatures]\n", argv[0]);
    return 2;
  }
  try {
    const size_t max_frames = argc > 3 ? (size_t)std::atoi(argv[3]) : 1000;
    const int nfeatures = argc > 4 ? std::atoi(argv[4]) : 3000;  // cv::ORB::create(3000), :31
    const std::vector<std::string> images = orbx::io::list_sequence_images(argv[1], argv[2]);
    auto orb = orbx::Feature2D::create(nfeatures);
    orbx::LKTracker lk;
    HammingMatcher matcher;
    orbx::io::GrayImage img1;
    std::vector<orbx::Point2f> pts1;
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (size_t i = 0; i < max_frames && i < images.size(); i++) {
      orbx::io::GrayImage img2 = orbx::io::read_png_gray(images[i]);
      const orbx::Image view2(img2.pixels.data(), img2.width, img2.height);
      if (i == 0) {
        std::vector<orbx::KeyPoint> kp1;
        orb->detect(view2, kp1);
        orbx::KeyPoint::convert(kp1, pts1);
        std::printf("%zu %zu detect 0 0\n", i, pts1.size());
      } else {
        const orbx::Image view1(img1.pixels.data(), img1.width, img1.height);
        std::vector<orbx::Point2f> pts2;
        // the previous frame's pyramid is still on the device unless the matcher path ran in between
        orbx::track_optical_flow(lk, i == 1 ? &view1 : nullptr, view2, pts1, pts2);
        const char* how = "track";
        if (pts2.size() < 150) {  // :70-72
          std::vector<orbx::KeyPoint> kp1, kp2;
          orbx::DescriptorMat des1, des2;
          orb->detectAndCompute(view1, kp1, des1);
          orbx::get_matches(*orb, matcher, kp1, des1, view2, kp2, des2, pts1, pts2);
          how = "match";
        }
        std::vector<float> fx(pts2.size()), fy(pts2.size());
        for (size_t k = 0; k < pts2.size(); k++) {
          fx[k] = pts2[k].x - pts1[k].x;
          fy[k] = pts2[k].y - pts1[k].y;
        }
        auto median = [](std::vector<float>& v) {
          if (v.empty()) return 0.f;
          std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
          return v[v.size() / 2];
        };
        std::printf("%zu %zu %s %.3f %.3f\n", i, pts2.size(), how, median(fx), median(fy));
        pts1 = pts2;  // :113
      }
      img1 = std::move(img2);  // :112
    }
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::printf("elapsed %.3f s\n", sec);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
