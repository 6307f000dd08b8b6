// test_host_mirror.cpp -- the reference's scratch harness src/compare.cpp
// (:39-62, :82-107: "for every CPU keypoint look up the GPU keypoint at the same
// (x,y) and print the Hamming distance of the descriptors") turned into a real
// test of the C++ host mirror (visual-odometry-gpu_amd/host/orb.hpp), with the
// CPU oracle as the checker.  Usage: test_host_mirror <raw-u8-file> <w> <h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>

#include "../../oracle/orb_oracle.h"
#include "../../visual-odometry-gpu_amd/host/orb.hpp"

static int fails = 0;
#define EXPECT(cond, ...)                        \
  do {                                           \
    if (!(cond)) {                               \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                  \
      std::printf("\n");                         \
      fails++;                                   \
    }                                            \
  } while (0)

static int hamming(const uint8_t* a, const uint8_t* b, const uint8_t* mask) {
  int d = 0;
  for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)((a[i] ^ b[i]) & mask[i]));
  return d;
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<uint8_t> px((size_t)w * h);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(px.data(), 1, px.size(), f) != px.size()) return 2;
  std::fclose(f);
  orbx::Image image(px.data(), w, h);

  // ---- ORBCPU twin vs oracle (compare.cpp:40 `orb_cpu.detectAndCompute`)
  {
    ORBCPU orb_cpu;
    std::vector<Keypoint> kps;
    std::vector<float> angles;
    std::vector<ORBDescriptor> descs;
    orb_cpu.detectAndCompute(image, kps, angles, descs);
    std::vector<int32_t> okp(2 * 3000);
    std::vector<float> oang(3000);
    std::vector<uint8_t> odesc(32 * 3000), ovalid(32 * 3000);
    const int n = oracle_detect_and_compute_cpu(px.data(), w, h, w, 3000, 50, 9, 3, 9, okp.data(), oang.data(),
                                                odesc.data(), ovalid.data());
    EXPECT((int)kps.size() == n, "ORBCPU count %zu vs oracle %d", kps.size(), n);
    std::map<std::pair<int, int>, int> gpu_map;  // compare.cpp:45-48
    for (size_t i = 0; i < kps.size(); i++) gpu_map[{kps[i].x, kps[i].y}] = (int)i;
    int not_found = 0, max_hd = 0;
    for (int i = 0; i < n; i++) {  // compare.cpp:50-62
      auto it = gpu_map.find({okp[2 * i], okp[2 * i + 1]});
      if (it == gpu_map.end()) {
        not_found++;
        continue;
      }
      const int hd = hamming(descs[it->second].data, &odesc[32 * i], &ovalid[32 * i]);
      if (hd > max_hd) max_hd = hd;
      EXPECT(it->second == i, "keypoint order differs at %d", i);
      EXPECT(std::memcmp(&angles[it->second], &oang[i], 4) == 0, "angle differs at %d", i);
    }
    EXPECT(not_found == 0, "%d oracle keypoints not found on the GPU", not_found);
    EXPECT(max_hd == 0, "max Hamming distance %d", max_hd);
    std::printf("ORBCPU: %zu keypoints, %d not found, max Hamming %d\n", kps.size(), not_found, max_hd);
  }

  // ---- ORB (GPU flavour, compare.cpp:65 `orb_gpu.detectAndCompute`) vs oracle intent
  {
    ORB orb_gpu(1000, 1.2f, 8);
    std::vector<Keypoint> kps;
    std::vector<float> angles, resp;
    std::vector<ORBDescriptor> descs;
    std::vector<int32_t> levels;
    orb_gpu.detectAndCompute(image, kps, angles, descs, &resp, &levels);
    // assign semantics: a second call must not append (D12)
    orb_gpu.detectAndCompute(image, kps, angles, descs, &resp, &levels);
    oracle_orb_params p = {1000, 1.2f, 8, 20, 9, 3, 31, 7, 0.04f, 0, 0};
    const int cap = 1000;
    std::vector<int32_t> okp(2 * cap), olkp(2 * cap), olev(cap);
    std::vector<float> oang(cap), oresp(cap);
    std::vector<uint8_t> odesc(32 * cap), ovalid(32 * cap);
    const int n = oracle_detect_and_compute_gpu(px.data(), w, h, w, &p, okp.data(), olkp.data(), olev.data(),
                                                oang.data(), oresp.data(), odesc.data(), ovalid.data(), cap);
    EXPECT((int)kps.size() == n, "ORB count %zu vs oracle %d", kps.size(), n);
    int bad = 0;
    for (int i = 0; i < n && i < (int)kps.size(); i++) {
      if (kps[i].x != okp[2 * i] || kps[i].y != okp[2 * i + 1] || levels[i] != olev[i]) bad++;
      if (hamming(descs[i].data, &odesc[32 * i], &ovalid[32 * i]) != 0) bad++;
      if (std::memcmp(&angles[i], &oang[i], 4) != 0) bad++;
      if (std::memcmp(&resp[i], &oresp[i], 4) != 0) bad++;
    }
    EXPECT(bad == 0, "%d mismatching fields", bad);
    std::printf("ORB: %zu keypoints over 8 levels, %d mismatches\n", kps.size(), bad);
  }

  // ---- sub-objects and free functions keep the reference call shapes
  {
    OrientedFAST fast;  // (20, 9, 3, 31)
    std::vector<Keypoint> kps = fast.detect(image, 434);
    std::vector<int32_t> okp(2 * 434);
    const int n = oracle_fast_detect(px.data(), w, h, w, 20, 9, 3, 434, okp.data());
    EXPECT((int)kps.size() == n && std::memcmp(kps.data(), okp.data(), sizeof(Keypoint) * n) == 0, "OrientedFAST::detect");
    std::vector<float> ang = fast.compute_orientations(image, kps);
    std::vector<float> oang(n);
    oracle_orientations(px.data(), w, h, w, okp.data(), n, 31, oang.data());
    EXPECT(std::memcmp(ang.data(), oang.data(), 4 * n) == 0, "compute_orientations");
    RotatedBRIEF brief;
    std::vector<ORBDescriptor> d = brief.compute(image, kps, ang);
    std::vector<uint8_t> od(32 * n), ov(32 * n);
    oracle_brief(px.data(), w, h, w, okp.data(), oang.data(), n, od.data(), ov.data(), nullptr, nullptr);
    EXPECT(std::memcmp(d.data(), od.data(), 32 * n) == 0, "RotatedBRIEF::compute");
    std::vector<float> hs;
    HarrisScore(image, kps, hs, 7, 0.04f);
    std::vector<float> ohs(n);
    oracle_harris(px.data(), w, h, w, okp.data(), n, 7, 0.04f, ohs.data());
    EXPECT(std::memcmp(hs.data(), ohs.data(), 4 * n) == 0, "HarrisScore");
    orbx::Image8 b1, b2, b3, sx;
    GaussianBlur1D(image, b1);
    GaussianBlur(image, b2);
    GaussianBlurCUDA(image, b3, 7);
    SobelCUDA(image, sx, 0);
    std::vector<uint8_t> o(px.size());
    oracle_blur5_sep(px.data(), w, h, w, o.data(), w);
    EXPECT(o == b1.pixels, "GaussianBlur1D");
    oracle_blur5_273(px.data(), w, h, w, o.data(), w);
    EXPECT(o == b2.pixels, "GaussianBlur");
    oracle_gaussian_blur_conv(px.data(), w, h, w, 7, o.data());
    EXPECT(o == b3.pixels, "GaussianBlurCUDA");
    oracle_sobel_u8(px.data(), w, h, w, 0, o.data());
    EXPECT(o == sx.pixels, "SobelCUDA");
    // error behaviour: exceptions, never exit()
    bool threw = false;
    try {
      std::vector<ORBDescriptor> dd;
      Brief(image, kps, std::vector<float>(1), dd);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw, "size mismatch must throw");
  }
  // ---- get_matches() shape (feature_matching.cpp:155-183): detect on two frames, knnMatch, ratio test
  {
    std::vector<uint8_t> px2(px);  // second frame: the first one shifted by (3, 2)
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) px2[(size_t)y * w + x] = px[(size_t)((y + 2) % h) * w + (x + 3) % w];
    ORB orb(1000);
    std::vector<Keypoint> k1, k2;
    std::vector<float> a1, a2;
    std::vector<ORBDescriptor> d1, d2;
    orb.detectAndCompute(image, k1, a1, d1);
    orb.detectAndCompute(orbx::Image(px2.data(), w, h), k2, a2, d2);
    HammingMatcher matcher;
    std::vector<std::vector<DMatch>> matches;
    matcher.knnMatch(d1, d2, matches, 2);
    std::vector<int32_t> oi(2 * d1.size()), od(2 * d1.size());
    oracle_knn2(d1[0].data, (int)d1.size(), d2[0].data, (int)d2.size(), oi.data(), od.data());
    int bad = 0, good = 0;
    for (size_t i = 0; i < matches.size(); i++) {
      if (matches[i].size() < 2) continue;  // :174
      const DMatch& m = matches[i][0];
      const DMatch& n = matches[i][1];
      if (m.trainIdx != oi[2 * i] || n.trainIdx != oi[2 * i + 1] || (int)m.distance != od[2 * i]) bad++;
      if (m.distance < 0.8 * n.distance) good++;  // :177
    }
    std::vector<DMatch> rm = matcher.ratioMatch(d1, d2, 0.8);
    EXPECT(bad == 0, "%d knnMatch mismatches", bad);
    EXPECT((int)rm.size() == good && good > 100, "ratio matches %zu vs %d", rm.size(), good);
    std::printf("HammingMatcher: %zu x %zu descriptors, %d ratio matches\n", d1.size(), d2.size(), good);
  }
  // ---- cv::Feature2D-shaped adapter + VisualOdom::get_matches (feature_tracking.cpp:31,61; feature_matching.cpp:155-183)
  {
    std::vector<uint8_t> px2(px);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) px2[(size_t)y * w + x] = px[(size_t)((y + 2) % h) * w + (x + 3) % w];
    auto orb = orbx::Feature2D::create(1000);
    std::vector<orbx::KeyPoint> kp1, kp2, kd;
    orbx::DescriptorMat des1, des2;
    orb->detectAndCompute(image, kp1, des1);
    // same results as the reference-shaped ORB class, in cv::KeyPoint conventions
    ORB ref(1000);
    std::vector<Keypoint> k;
    std::vector<float> a, r;
    std::vector<ORBDescriptor> d;
    std::vector<int32_t> lv;
    ref.detectAndCompute(image, k, a, d, &r, &lv);
    EXPECT(kp1.size() == k.size() && des1.rows == (int)k.size() && des1.cols == 32, "Feature2D sizes");
    int badkp = 0;
    for (size_t i = 0; i < k.size() && i < kp1.size(); i++) {
      const float deg = a[i] * 57.29577951308232f;
      const bool ok = kp1[i].pt.x == (float)k[i].x && kp1[i].pt.y == (float)k[i].y && kp1[i].octave == lv[i] &&
                      kp1[i].response == r[i] && kp1[i].angle >= 0.f && kp1[i].angle < 360.f &&
                      std::fabs(std::fmod(kp1[i].angle - deg + 720.f, 360.f)) < 1e-3f &&
                      std::fabs(kp1[i].size - 31.f * std::pow(1.2f, (float)lv[i])) < 1e-3f &&
                      std::memcmp(des1.ptr((int)i), d[i].data, 32) == 0;
      if (!ok) badkp++;
    }
    EXPECT(badkp == 0, "%d Feature2D keypoints differ from ORB::detectAndCompute", badkp);
    orb->detect(image, kd);
    EXPECT(kd.size() == kp1.size() && (kd.empty() || kd[0].angle == -1.f), "Feature2D::detect");
    std::vector<orbx::Point2f> p0;
    orbx::KeyPoint::convert(kd, p0);
    EXPECT(p0.size() == kd.size() && (p0.empty() || (p0[0].x == kd[0].pt.x && p0[0].y == kd[0].pt.y)), "convert");
    HammingMatcher matcher;
    std::vector<orbx::Point2f> pts1, pts2;
    orbx::get_matches(*orb, matcher, kp1, des1, orbx::Image(px2.data(), w, h), kp2, des2, pts1, pts2);
    // the second frame is the first one moved by (-3, -2): matched level-0 points must agree with that
    int consistent = 0;
    for (size_t i = 0; i < pts1.size(); i++)
      if (std::fabs(pts1[i].x - pts2[i].x - 3.f) <= 4.f && std::fabs(pts1[i].y - pts2[i].y - 2.f) <= 4.f) consistent++;
    std::vector<DMatch> rm = matcher.ratioMatch(des1.d, des2.d, 0.8);
    EXPECT(pts1.size() == rm.size() && pts1.size() == pts2.size() && pts1.size() > 100, "get_matches: %zu vs %zu",
           pts1.size(), rm.size());
    EXPECT(consistent * 10 >= (int)pts1.size() * 9, "only %d of %zu matches follow the shift", consistent, pts1.size());
    std::printf("Feature2D/get_matches: %zu keypoints, %zu matches, %d consistent with the shift\n", kp1.size(),
                pts1.size(), consistent);
  }
  // ---- track_optical_flow() shape (feature_tracking.cpp:166-193) vs the LK oracle
  {
    std::vector<uint8_t> px2(px);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) px2[(size_t)y * w + x] = px[(size_t)((y + 2) % h) * w + (x + 3) % w];
    auto orb = orbx::Feature2D::create(600);
    std::vector<orbx::KeyPoint> kp1;
    orb->detect(image, kp1);
    std::vector<orbx::Point2f> pts1, pts2;
    orbx::KeyPoint::convert(kp1, pts1);
    const std::vector<orbx::Point2f> pts0 = pts1;
    std::vector<float> onext(2 * pts0.size()), oerr(pts0.size());
    std::vector<uint8_t> ost(pts0.size());
    oracle_lk_track(px.data(), px2.data(), w, h, w, w, reinterpret_cast<const float*>(pts0.data()), (int)pts0.size(),
                    onext.data(), ost.data(), oerr.data(), 21, 3, 30, 0.01);
    orbx::LKTracker lk;
    const orbx::Image img2(px2.data(), w, h);
    orbx::track_optical_flow(lk, &image, img2, pts1, pts2);
    size_t k = 0;
    int bad = 0, shift_ok = 0;
    for (size_t i = 0; i < pts0.size(); i++) {
      if (!ost[i]) continue;
      if (k >= pts2.size() || std::memcmp(&pts2[k], &onext[2 * i], 8) != 0 || std::memcmp(&pts1[k], &pts0[i], 8) != 0)
        bad++;
      else if (std::fabs(pts1[k].x - pts2[k].x - 3.f) < 0.5f && std::fabs(pts1[k].y - pts2[k].y - 2.f) < 0.5f)
        shift_ok++;
      k++;
    }
    EXPECT(k == pts2.size() && bad == 0, "track_optical_flow: %zu tracks vs oracle %zu, %d differ", pts2.size(), k, bad);
    EXPECT(shift_ok * 10 >= (int)k * 8, "only %d of %zu tracks follow the shift", shift_ok, k);
    // second call with the cached pyramid (img1 = img2.clone()): track back to the first frame
    std::vector<orbx::Point2f> back;
    std::vector<orbx::Point2f> fwd = pts2;
    orbx::track_optical_flow(lk, nullptr, image, fwd, back);
    EXPECT(back.size() > pts2.size() / 2, "backward tracking lost too many points");
    std::printf("LKTracker: %zu of %zu points tracked, %d consistent with the shift, %zu tracked back\n", pts2.size(),
                pts0.size(), shift_ok, back.size());
    // a frame with NO points (the VO loop's fallback produced none) still advances the cached previous
    // frame: after (empty -> img2), tracking pts0 with prev == nullptr into `image` must equal tracking
    // them with prev = img2 given explicitly
    {
      std::vector<orbx::Point2f> none, none2;
      orbx::track_optical_flow(lk, nullptr, img2, none, none2);  // the tracker's "previous" is now img2
      std::vector<orbx::Point2f> a1 = pts0, a2, b1 = pts0, b2;
      orbx::track_optical_flow(lk, nullptr, image, a1, a2);
      orbx::LKTracker fresh;
      orbx::track_optical_flow(fresh, &img2, image, b1, b2);
      EXPECT(a2.size() == b2.size() && !a2.empty() && std::memcmp(a2.data(), b2.data(), a2.size() * 8) == 0,
             "cached previous frame lags after an empty-points call (%zu vs %zu tracks)", a2.size(), b2.size());
    }
  }
  std::printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
  return fails ? 1 : 0;
}
