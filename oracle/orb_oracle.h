/* orb_oracle.h -- CPU restatement of the reference's ORB front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under visual-odometry-gpu_amd/ (the
 * product) may include, link or call this; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PINNING STATUS: partially pinned.  The reference's own implementation of
 * this path (src/orb_cpu.cpp) needs OpenCV headers, which are absent from
 * this image, and the task rules forbid writing stand-ins for them, so
 * oracle/_ref cannot be built and no per-element golden vectors of the
 * reference exist.  What pins this restatement is
 *   (1) the known answers SURVEY.md §7/§8 recorded from a run of the real
 *       orb_cpu.cpp on the reference's own 000000.png (keypoint / corner /
 *       pre-test counts at two thresholds, first & last keypoint, number of
 *       border angles, the row at which the 3000 cap is hit, the D15
 *       statistics, the pyramid level sizes and per-level quotas) --
 *       tests/test_oracle_known_answers.py;
 *   (2) an independent numpy restatement written from the Python prototype
 *       orb.py / blur.py -- tests/test_oracle_vs_numpy.py.
 * Stages with no runnable reference at all (Harris, top-N, pyramid, the
 * multi-level orchestrator; SURVEY.md §8c "What pins each stage") are
 * "parity unpinned": this file restates the reference's INTENT (SURVEY.md
 * §2.3) and says so at each function.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the learned test pattern, 256 x (x1,y1,x2,y2)  [src/orb_pattern.cpp:4-260] */
extern const int8_t oracle_pattern_31[1024];

/* ---- CPU flavour (src/orb_cpu.cpp) ------------------------------------- */

/* FAST-n segment test + score  [orb_cpu.cpp:23-103].
 * scores: h*w floats (pitch w), fully overwritten.  Optional outputs:
 * n_pretest = pixels passing the 4-point pre-test, n_corners = corners. */
void oracle_fast_score(const uint8_t* img, int w, int h, int stride, int threshold, int n,
                       float* scores, int64_t* n_pretest, int64_t* n_corners);

/* NMS + row-major cap  [orb_cpu.cpp:105-134].  Writes at most `nfeatures`
 * keypoints as (x,y) int pairs; returns the number written.  If total_out is
 * non-NULL it receives the number of survivors ignoring the cap. */
int oracle_nms(const float* scores, int w, int h, int nms_window, int nfeatures, int32_t* kps_xy,
               int64_t* total_out);

/* detect() = fast_score + nms  [orb_cpu.cpp:23-137] */
int oracle_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int n, int nms_window,
                       int nfeatures, int32_t* kps_xy);

/* intensity-centroid angle  [orb_cpu.cpp:139-183] */
void oracle_orientations(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, int nkp,
                         int patch_size, float* angles);

/* cv::integral restated: (h+1)*(w+1) int32, first row/col zero [orb_cpu.cpp:207-208] */
void oracle_integral(const uint8_t* img, int w, int h, int stride, int32_t* integral);

/* rotated BRIEF-256  [orb_cpu.cpp:190-258].
 * desc: nkp*32 bytes.  valid (optional): nkp*32 bytes, bit set iff that
 * descriptor bit is DEFINED BEHAVIOUR in the reference (both 5x5 boxes lie
 * wholly inside the image, or the reference's own bounds check skipped the
 * bit).  Where the reference's check admits a centre whose box leaves the
 * image (defect D15: it then reads outside the integral image) this
 * restatement sums the in-image pixels only (zero-extended image) and clears
 * the valid bit.  n_skipped / n_oob (optional) count bits skipped by the
 * reference's check / bits with undefined reads. */
void oracle_brief(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, const float* angles,
                  int nkp, uint8_t* desc, uint8_t* valid, int64_t* n_skipped, int64_t* n_oob);

/* ORBCPU::detectAndCompute  [orb_cpu.cpp:271-276]; single level.
 * Returns keypoint count (<= nfeatures). */
int oracle_detect_and_compute_cpu(const uint8_t* img, int w, int h, int stride, int nfeatures, int threshold,
                                  int n, int nms_window, int patch_size, int32_t* kps_xy, float* angles,
                                  uint8_t* desc, uint8_t* valid);

/* ---- stage kernels of the GPU flavour (restated arithmetic) ------------ */

/* BORDER_REFLECT_101 index  [GaussianBlur1D.cu:27-32] */
int oracle_reflect101(int p, int len);

/* separable [1 4 6 4 1]/16 H then V, float, then convertTo(CV_8U)
 * [GaussianBlur1D.cu:34-163] */
void oracle_blur5_sep(const uint8_t* img, int w, int h, int stride, uint8_t* dst, int dst_stride);

/* 5x5 /273 kernel on a REFLECT_101 padded image, then convertTo(CV_8U)
 * [GaussianBlur.cu:21-130] */
void oracle_blur5_273(const uint8_t* img, int w, int h, int stride, uint8_t* dst, int dst_stride);

/* createGaussianKernel(K, sigma<=0 -> heuristic)  [GaussianBlur.cpp:7-37] */
void oracle_gaussian_kernel(int K, float sigma, float* kernel);

/* valid KxK correlation of a pre-padded float image, float accumulate in
 * (i,j) row-major order  [Convolution.cu:40-53]; out is (h-K+1)*(w-K+1). */
void oracle_conv2d_f32(const float* in, int w, int h, const float* kernel, int K, float* out);

/* conv2d() wrapper semantics: u8 in (pre-padded) -> float conv -> CV_8U
 * (round-half-even, saturate)  [Convolution.cu:57-101] */
void oracle_conv2d_u8(const uint8_t* in, int w, int h, int stride, const float* kernel, int K, uint8_t* out);

/* GaussianBlurCUDA(image,dst,K) = REFLECT_101 pad + conv2d  [GaussianBlur.cpp:39-49] */
void oracle_gaussian_blur_conv(const uint8_t* img, int w, int h, int stride, int K, uint8_t* dst);

/* SobelCUDA(image,dst,dir) = REFLECT_101 pad + conv2d 3x3, output CV_8U as
 * the reference does (D4)  [Sobel.cpp:6-32] */
void oracle_sobel_u8(const uint8_t* img, int w, int h, int stride, int dir, uint8_t* dst);

/* Harris response at keypoints -- INTENT of HarrisScore.cu:23-89 with
 * defects D4-D8 repaired (float Sobel, Sxy from IxIy, float k):
 *   Ix,Iy = 3x3 Sobel (REFLECT_101); A=G(Ix^2), B=G(Ix*Iy), C=G(Iy^2) with
 *   G = window x window Gaussian from oracle_gaussian_kernel (REFLECT_101 on
 *   the product images); R = (A*C - B*B) - (k*(A+C))*(A+C).
 * PARITY UNPINNED (no runnable reference). */
void oracle_harris(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, int nkp, int window,
                   float k, float* out);

/* ---- pyramid + multi-level orchestrator (GPU flavour intent) ----------- */

/* level size  [orb.cpp:117-118]: scale=(float)pow(sf,l); round(W/scale) */
void oracle_level_size(int w0, int h0, float scale_factor, int level, int* wl, int* hl);
/* per-level quota  [orb.cpp:62] */
int oracle_level_quota(int nfeatures, float scale_factor, int nlevels, int level);
/* keypoint rescale factor  [orb.cpp:95]: (float)pow(sf,l) */
float oracle_level_scale(float scale_factor, int level);

/* 8-bit bilinear resize.  The reference calls cv::resize(INTER_LINEAR)
 * [orb.cpp:119]; OpenCV is not in this image, so this restates OpenCV 4.x's
 * published generic 8UC1 fixed-point path (11-bit coefficients,
 * imgproc/src/resize.cpp HResizeLinear/VResizeLinear).  PARITY UNPINNED. */
void oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride);

/* top-N by response: stable order (R desc, index asc)  [orb.cpp:67-86 intent,
 * D9].  Writes min(n,keep) indices; returns that count. */
int oracle_select_top(const float* resp, int n, int keep, int32_t* idx_out);

typedef struct {
  int nfeatures;      /* ORB(nfeatures)              orb.hpp:36 */
  float scale_factor; /* ORB(scaleFactor)            orb.hpp:36 */
  int nlevels;        /* ORB(nlevels)                orb.hpp:36 */
  int threshold;      /* OrientedFAST(threshold)     orb.hpp:12 */
  int n;              /* OrientedFAST(n)             orb.hpp:12 */
  int nms_window;     /* OrientedFAST(nms_window)    orb.hpp:12 */
  int patch_size;     /* OrientedFAST(patch_size)    orb.hpp:12 */
  int harris_window;  /* HarrisScore(...,7,...)      orb.cpp:65 */
  float harris_k;     /* HarrisScore(...,0.04)       orb.cpp:65 */
  int blur_levels;    /* 0 none (orb.cpp:111-120), 1 levels>=1 (orb_cpu.cpp:278-290), 2 all levels */
  int blur_kind;      /* 0 separable /16 (GaussianBlur1D.cu), 1 5x5 /273 (GaussianBlur.cu) */
} oracle_orb_params;

/* ORB::detectAndCompute  [orb.cpp:58-109], intent (D9-D12 repaired):
 * outputs are ASSIGNED, level order, within a level sorted by
 * (Harris desc, row-major index asc).  kps_xy are rescaled to level-0
 * coordinates by (int)((float)x * scale_l); kps_level_xy (optional) keeps
 * the level coordinates; levels (optional) the level index.
 * Returns the keypoint count (<= capacity required: sum of quotas). */
int oracle_detect_and_compute_gpu(const uint8_t* img, int w, int h, int stride, const oracle_orb_params* p,
                                  int32_t* kps_xy, int32_t* kps_level_xy, int32_t* levels, float* angles,
                                  float* responses, uint8_t* desc, uint8_t* valid, int capacity);

/* builds pyramid level l (resize from level 0 + optional blur) into dst
 * (pitch = level width); used by tests to compare intermediate images. */
void oracle_build_level(const uint8_t* img, int w, int h, int stride, const oracle_orb_params* p, int level,
                        uint8_t* dst);

/* ---- next row: descriptor matching ---------------- */

/* flann->knnMatch(des1, des2, matches, 2)  [feature_matching.cpp:166-168,
 * feature_tracking.cpp:203-204] restated as EXACT brute-force Hamming 2-NN (the
 * reference's FLANN index -- LSH for ORB, feature_tracking.cpp:32 -- is an
 * approximate search inside OpenCV, which is absent: PARITY UNPINNED).
 * idx/dist: nq x 2 (best, second best); ties keep the lower train index;
 * entries are -1 when the train set has fewer than 1 / 2 descriptors. */
void oracle_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* idx, int32_t* dist);

/* the ratio test of feature_matching.cpp:172-181: keep query i iff it has two
 * neighbours and (float)d1 < ratio * (float)d2 evaluated in double like the
 * reference's `m.distance < 0.8 * n.distance`.  Returns the number of matches;
 * query_idx/train_idx/dist1 (optional) hold them in query order. */
int oracle_match_ratio(const uint8_t* query, int nq, const uint8_t* train, int nt, double ratio,
                       int32_t* query_idx, int32_t* train_idx, int32_t* dist1);

/* ---- next row: pyramidal Lucas-Kanade tracking ---- */
/* (lk_oracle.c; PARITY UNPINNED -- restates OpenCV 4.x's calcOpticalFlowPyrLK as
 * called at feature_tracking.cpp:175-181, see the header of lk_oracle.c) */

/* cv::pyrDown for 8UC1: 5x5 [1 4 6 4 1]^2, (sum + 128) >> 8, BORDER_REFLECT_101;
 * dst is ((sw+1)/2) x ((sh+1)/2), tight. */
void oracle_lk_pyr_down(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst);
/* Scharr derivatives (dx, dy) as interleaved int16, REFLECT_101 (calcSharrDeriv) */
void oracle_lk_scharr(const uint8_t* img, int w, int h, int stride, int16_t* deriv);
/* calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(win,win),
 * max_level, TermCriteria(COUNT+EPS, max_iters, epsilon)), flags = 0,
 * minEigThreshold = 1e-4.  prev_pts / next_pts: n (x, y) float pairs; err may be
 * NULL.  Returns the top pyramid level actually used (<= max_level), -1 on bad
 * arguments. */
int oracle_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, int stride_prev, int stride_next,
                    const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int win,
                    int max_level, int max_iters, double epsilon);

#ifdef __cplusplus
}
#endif
#endif
