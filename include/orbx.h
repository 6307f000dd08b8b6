/* orbx.h -- C ABI of the MI355X-native ORB feature front-end (liborbx.so).
 *
 * Drop-in boundary for the reference's `orb.hpp` detectAndCompute path
 * (WeeFav/Visual-Odometry-GPU).  Every entry point names the reference
 * interface it replaces (paths relative to the reference root).  Plain
 * pointers and sizes only; no C++ or torch types.  All functions return an
 * orbx_status (0 = ok); none of them calls exit() or prints.
 *
 * A context owns every device allocation (pyramids, masks, candidate and
 * result slots for `max_batch` frames of up to max_width x max_height), its
 * own HIP stream, and is single-threaded; distinct contexts are independent
 * (one per GPU / per host thread).  Nothing is allocated on the per-frame
 * path.
 */
#ifndef ORBX_H
#define ORBX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_MAX_LEVELS 16

typedef enum {
  ORBX_OK = 0,
  ORBX_ERR_INVALID_ARG = 1,  /* bad pointer / size / parameter */
  ORBX_ERR_CAPACITY = 2,     /* caller buffer too small; *count still reports the required size */
  ORBX_ERR_HIP = 3,          /* HIP runtime error; see orbx_last_error_string */
  ORBX_ERR_NO_DEVICE = 4,    /* no gfx950 device / kernels not loadable */
  ORBX_ERR_UNSUPPORTED = 5   /* parameter combination outside the supported range */
} orbx_status;

/* include/orb.hpp:4  struct Keypoint { int x, y; } */
typedef struct {
  int32_t x, y;
} orbx_keypoint;

/* include/orb.hpp:6-8  struct ORBDescriptor { uint8_t data[32]; }
 * bit i lives at data[i>>3] & (1 << (i&7))  (src/orb_cpu.cpp:251) */
typedef struct {
  uint8_t data[32];
} orbx_descriptor;

typedef enum {
  /* GPU flavour, src/orb.cpp:58-109: per level FAST cap = 2*quota in row-major
   * order, Harris response, keep the `quota` best, output sorted by
   * (response desc, row-major index asc) within a level. */
  ORBX_SELECT_HARRIS = 0,
  /* CPU flavour, src/orb_cpu.cpp:271-276: keep the first `cap` NMS survivors
   * in row-major order, no Harris (responses are reported as 0). */
  ORBX_SELECT_ROWMAJOR = 1
} orbx_select_mode;

typedef enum { ORBX_BLUR_NONE = 0, ORBX_BLUR_UPPER = 1, ORBX_BLUR_ALL = 2 } orbx_blur_levels;
typedef enum { ORBX_BLUR_SEP16 = 0, ORBX_BLUR_K273 = 1 } orbx_blur_kind;

/* One POD for every knob of both reference flavours (SURVEY.md §5 "Config"). */
typedef struct {
  int32_t nfeatures;     /* ORB(nfeatures=500)            include/orb.hpp:36 */
  float scale_factor;    /* ORB(scaleFactor=1.2f)         include/orb.hpp:36 */
  int32_t nlevels;       /* ORB(nlevels=8)                include/orb.hpp:36 */
  int32_t threshold;     /* OrientedFAST(threshold=20)    include/orb.hpp:12 */
  int32_t n;             /* OrientedFAST(n=9)             include/orb.hpp:12 */
  int32_t nms_window;    /* OrientedFAST(nms_window=3)    include/orb.hpp:12 */
  int32_t patch_size;    /* OrientedFAST(patch_size=31)   include/orb.hpp:12 */
  int32_t harris_window; /* HarrisScore(..., 7, ...)      src/orb.cpp:65 */
  float harris_k;        /* HarrisScore(..., 0.04)        src/orb.cpp:65 */
  int32_t select_mode;   /* orbx_select_mode */
  int32_t blur_levels;   /* orbx_blur_levels: none = src/orb.cpp:111-120, upper = src/orb_cpu.cpp:278-290 */
  int32_t blur_kind;     /* orbx_blur_kind: src/cuda/GaussianBlur1D.cu / src/cuda/GaussianBlur.cu */
  int32_t max_width;     /* largest frame the context must handle */
  int32_t max_height;
  int32_t max_batch;     /* frames in flight per batched call */
  int32_t device;        /* HIP device ordinal, -1 = current device */
} orbx_params;

typedef struct orbx_ctx orbx_ctx;

/* Defaults of the GPU flavour: ORB(500,1.2f,8) + OrientedFAST(20,9,3,31)
 * (include/orb.hpp:12,36), HarrisScore(7,0.04) (src/orb.cpp:65), no blur. */
int orbx_params_default_gpu(orbx_params* p);
/* Defaults of the CPU flavour: OrientedFASTCPU(3000,50,9,3,9), one level,
 * row-major selection (include/orb_cpu.hpp:6, src/orb_cpu.cpp:271-276). */
int orbx_params_default_cpu(orbx_params* p);

/* replaces the ORB / OrientedFAST / RotatedBRIEF constructors
 * (src/orb.cpp:10-56); no stdout noise. */
int orbx_create(const orbx_params* p, orbx_ctx** out);
void orbx_destroy(orbx_ctx* ctx);

/* replaces cudaCheckErrors -> fprintf + exit(1) (src/cuda/Fast.cu:8-18).
 * ctx may be NULL (reports the last create failure of this thread). */
const char* orbx_last_error_string(const orbx_ctx* ctx);
const char* orbx_status_string(int status);
/* "liborbx <version> gfx950" */
const char* orbx_version(void);

/* Per-level geometry the context derived for a w x h frame (src/orb.cpp:62,
 * :117-118): sizes, quota and FAST cap.  Arrays must hold nlevels entries. */
int orbx_get_plan(orbx_ctx* ctx, int width, int height, int32_t* level_w, int32_t* level_h, int32_t* quota,
                  int32_t* fast_cap, float* level_scale, int32_t* out_capacity);

/* ---- whole path ---------------------------------------------------------- */

/* ORB::detectAndCompute(image, keypoints, orientations, descriptors)
 * (include/orb.hpp:37, src/orb.cpp:58-109; CPU twin src/orb_cpu.cpp:271-276).
 * Host image in, host arrays out (ASSIGN semantics).
 * keypoints are in level-0 coordinates ((int)(x*scale_l), src/orb.cpp:94-98).
 * responses / levels / level_kps may be NULL.  *count = keypoints produced;
 * if it exceeds `capacity` only `capacity` entries are written and
 * ORBX_ERR_CAPACITY is returned. */
int orbx_detect_and_compute(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                            orbx_keypoint* keypoints, float* orientations, orbx_descriptor* descriptors,
                            float* responses, int32_t* levels, orbx_keypoint* level_kps, int capacity,
                            int* count);

/* Batched, device-resident variant (the benchmark path; BASELINE.json configs
 * 2-4): `n` frames already in HBM at d_frames + i*frame_stride, each
 * height rows of row_stride bytes.  Runs asynchronously on the context's
 * stream (or `stream`, a hipStream_t, if non-NULL); results stay in the
 * context's device-side result slots until the next batched call. */
int orbx_detect_and_compute_batch_device(orbx_ctx* ctx, const void* d_frames, int n, int width, int height,
                                         int row_stride, size_t frame_stride, void* stream);
/* Same for `n` host frames (H2D copy from the caller's memory included; pass pinned memory for an
 * asynchronous copy, pageable memory is staged by the HIP runtime). */
int orbx_detect_and_compute_batch_host(orbx_ctx* ctx, const uint8_t* frames, int n, int width, int height,
                                       int row_stride, size_t frame_stride);
/* Blocks until the last batched call has finished. */
int orbx_wait(orbx_ctx* ctx);

/* Device-side result slots of the last batch (fixed stride `slot_capacity`
 * entries per frame), for consumers that stay on the GPU. */
typedef struct {
  const int32_t* counts;            /* [n] */
  const orbx_keypoint* keypoints;   /* [n][slot_capacity] level-0 coords */
  const orbx_keypoint* level_kps;   /* [n][slot_capacity] level coords */
  const float* orientations;        /* [n][slot_capacity] */
  const float* responses;           /* [n][slot_capacity] */
  const int32_t* levels;            /* [n][slot_capacity] */
  const orbx_descriptor* descriptors; /* [n][slot_capacity] */
  int32_t slot_capacity;
  int32_t n;
  const uint32_t* keypoints16;      /* [n][slot_capacity] the same level-0 coords packed, x | y << 16 (both <= 16384) */
} orbx_batch_view;
int orbx_batch_results_device(orbx_ctx* ctx, orbx_batch_view* view);

/* Copies frames [first, first+n) of the last batch to host arrays with
 * `capacity` entries per frame (any output may be NULL except counts). */
int orbx_batch_fetch(orbx_ctx* ctx, int first, int n, int32_t* counts, orbx_keypoint* keypoints,
                     float* orientations, orbx_descriptor* descriptors, float* responses, int32_t* levels,
                     orbx_keypoint* level_kps, int capacity);

/* Pipelined consumers (a VO loop that wants the keypoints of frame batch i on the host while
 * batch i+1 is being processed; the reference copies results back synchronously after every
 * kernel, src/cuda/Fast.cu:238-239, src/cuda/Brief.cu:131).  The context keeps a ring of FOUR result
 * blocks, one per batch in turn:
 *   orbx_detect_and_compute_batch_device(batch i);  orbx_batch_prefetch();
 *   orbx_detect_and_compute_batch_device(batch i+1);          -- kernels overlap the copy of i
 *   orbx_batch_fetch_previous(...)  -> results of batch i (waits for the copy only);  orbx_batch_prefetch();  ...
 * (The copy is the runtime's copy kernel behind a wait for the batch's end.  Enqueued once the previous batch's
 * results have been read, as above, it costs 5-7 % of the frame rate; enqueued right behind its batch it may sit in
 * a hardware queue in front of the other lane's kernels for the whole batch -- measured between 0.7 and 0.99 of the
 * rate without copies, depending on the streams the process has.  orbx_set_host_results below needs no copy at all.)
 * orbx_batch_prefetch starts an asynchronous D2H copy of the last batch's block into its pinned
 * mirror on the context's copy stream; orbx_batch_fetch / _previous then wait for that copy
 * instead of copying.  The results of batch i stay in its block until batch i+4 is submitted; the views
 * reach batch i (the last one) and batch i-1. */
int orbx_batch_prefetch(orbx_ctx* ctx);
/* The same for consumers that want what the reference's detectAndCompute returns and nothing else (keypoints,
 * orientations, descriptors: include/orb.hpp:37): only the counts | keypoints16 | orientations | descriptors
 * sections of the block are copied, the keypoints as packed 16-bit pairs -- 40 instead of 64 bytes per keypoint
 * slot, which is what keeps the copy under the host link's rate at the benchmark's frame rate.
 * orbx_batch_results_host then reports keypoints / responses / levels / level_kps as NULL (keypoints16 is always
 * there); a fetch unpacks the keypoints from keypoints16, and one that asks for responses / levels / level_kps
 * copies the remaining sections first (blocking). */
int orbx_batch_prefetch_compact(orbx_ctx* ctx);
/* Host results without a copy (default off): with enable = 1 the kernel that finishes a batch (orientation + BRIEF)
 * writes the compact record -- counts | keypoints16 | orientations | descriptors -- of every keypoint into the block's
 * PINNED HOST mirror as well, in coalesced stores that travel the host link while the kernel runs.
 * The block then counts as compact-copied from the moment its batch is enqueued: orbx_batch_prefetch_compact has
 * nothing to do (nothing is copied, no copy kernel competes with the next batch, the copy stream is not involved),
 * orbx_batch_prefetch copies the other sections, and the host views / fetches wait for the batch's end and deliver
 * the same bytes as before (orbx_batch_results_host without a preceding orbx_batch_prefetch: the compact view).  The device-side result block is written as always.  (The reference copies every stage's results back
 * with a blocking cudaMemcpy: src/cuda/Fast.cu:238-239, src/cuda/Brief.cu:131.)  bench.py: fps_with_d2h. */
int orbx_set_host_results(orbx_ctx* ctx, int enable);
/* Zero-copy host view of a result block: pointers into the context's PINNED mirror of the last batch
 * (previous = 0) or of the batch `previous` calls before it (1..3: the ring has four blocks), same layout
 * as the device view (fixed stride `slot_capacity` entries per frame; only the first counts[f] entries of
 * frame f are valid).  Waits for the block's copy (starts it if orbx_batch_prefetch was not called).  The
 * view stays valid until the block is written again, i.e. until the fourth batched call after the one it
 * belongs to.  A streaming consumer that reads batch i - 2 while batches i - 1 and i run keeps both lanes of
 * the pipelined mode busy (bench.py: fps_with_d2h). */
int orbx_batch_results_host(orbx_ctx* ctx, int previous, orbx_batch_view* view);
int orbx_batch_fetch_previous(orbx_ctx* ctx, int first, int n, int32_t* counts, orbx_keypoint* keypoints,
                              float* orientations, orbx_descriptor* descriptors, float* responses, int32_t* levels,
                              orbx_keypoint* level_kps, int capacity);

/* Per-stage device timings (ms) of the last batched call, in order:
 * pyramid, blur, fast+nms, compact, harris, select, orient+brief, total. */
#define ORBX_NUM_STAGE_TIMES 8
/* enable: 0 = off, 1 = events around every stage, 2 = only around the two
 * roofline stages (blur, fast+nms; the other entries and `total` read 0). */
int orbx_enable_stage_timing(orbx_ctx* ctx, int enable);
int orbx_last_stage_times(orbx_ctx* ctx, float* ms);
/* Same for the timed batched call `back` calls ago (0 = the last one; up to
 * ORBX_EVENT_SETS-1): several timed calls may be enqueued back to back and read
 * after one orbx_wait(), so timing adds no host synchronisation between steps. */
#define ORBX_EVENT_SETS 64
int orbx_stage_times_history(orbx_ctx* ctx, int back, float* ms);

/* FAST/NMS tiles that provably cannot contribute to the first `cap` row-major
 * survivors exit early in the batched path (results are identical either way;
 * DESIGN.md "Early exit").  enable = 0 makes every tile do the full work (used
 * to measure the kernel's full-work throughput).  Default: enabled. */
int orbx_set_fast_early_exit(orbx_ctx* ctx, int enable);

/* With blur on every level (ORBX_BLUR_ALL, separable kind) the batched path builds and blurs the
 * pyramid in ONE kernel (buildPyramid + GaussianBlur of src/orb_cpu.cpp:278-290 fused: the
 * un-blurred pyramid is never written).  enable = 0 runs the two kernels separately (identical
 * results; used to time / profile each kernel on its own).  Default: enabled. */
int orbx_set_fused_pyramid_blur(orbx_ctx* ctx, int enable);

/* Diagnostics of the last whole-path batch: how many FAST/NMS tiles did the full
 * work (`worked`) out of all tiles of the batch (`total`); the rest took the early
 * exit.  With the early exit disabled worked == total. */
int orbx_fast_tile_counts(orbx_ctx* ctx, long long* worked, long long* total);

/* Top-rows-first pyramid (whole path, blur on every level, FAST early exit on, large batches): the pyramid
 * rows the top FAST tile rows need are produced first, FAST runs on those tile rows, and the remaining rows
 * of a level are produced only if the level does not yet hold its `cap` survivors -- keypoints are kept in
 * row-major order up to the cap (src/orb_cpu.cpp:108-110, src/orb.cpp:63), so nothing below is ever read.
 * Results are identical in every mode.  mode 0: never (one pass), 1: whenever eligible, 2 (default): adaptive
 * -- the second pass reports how many levels it could skip, and while that is less than a quarter the batches
 * run in one pass (with a probe every 128th batch); and the DEPTH of the first pass follows the stream: the
 * selection reports in which row each level's cap filled, and the FAST tile rows of the level are sized so that
 * the first pass ends just below it (the work is re-partitioned a few times per stream; never a result changes). */
int orbx_set_top_rows_first(orbx_ctx* ctx, int mode);

/* Pipelined batches (default off).  With enable = 1, consecutive orbx_detect_and_compute_batch_device calls on the
 * context's own stream (stream = NULL) alternate between two LANES -- each with its own stream and its own working
 * pools; batch k uses the lane of its result block -- so the kernels of one batch overlap the tails and the nearly
 * empty launches of the other (KITTI, 256 frames per batch: ~8 % more frames/s).  Results are unchanged.  What a
 * caller may do between two such calls without losing the overlap: orbx_batch_prefetch, orbx_batch_results_host /
 * orbx_batch_fetch_previous (they follow the result block's own events); orbx_wait and orbx_batch_fetch wait for the
 * batches concerned; set_plan (a new frame size), the stage operators that use the pools, orbx_batch_match_consecutive
 * and orbx_destroy wait for both lanes first.  Costs a second set of pools (the first call allocates it; if that
 * fails -- ORBX_ERR_HIP -- what had been allocated is freed and the mode stays off).  Batches on a caller's stream,
 * host-frame batches and single frames are not pipelined, and may be mixed freely with pipelined ones: a batch that
 * comes to a lane's pools or to a result block on another stream than their previous user makes its stream wait
 * for that user's event (device-side, no host stall). */
int orbx_set_pipelined_batches(orbx_ctx* ctx, int enable);

/* Same for the pyramid: pyramid pixels the last whole-path batch PRODUCED out of all pyramid pixels of its
 * frames.  With blur on every level, the FAST early exit on and a large batch, the pyramid is built top rows
 * first and the remaining rows of a level are produced only if its top FAST tile rows did not already hold
 * the level's `cap` survivors (they are never read otherwise; results are identical).  Otherwise
 * produced == total. */
int orbx_pyramid_pixel_counts(orbx_ctx* ctx, long long* produced, long long* total);

/* Runs only the blur + FAST/NMS stages of the last-built pyramid `reps` times
 * (the roofline kernels) and reports the average duration of
 * each, measured with HIP events on the context's stream. */
int orbx_bench_stage(orbx_ctx* ctx, int n_frames, int stage, int reps, float* avg_ms);
#define ORBX_STAGE_PYRAMID 0
#define ORBX_STAGE_BLUR 1
#define ORBX_STAGE_FAST 2
#define ORBX_STAGE_COMPACT 3
#define ORBX_STAGE_HARRIS 4
#define ORBX_STAGE_SELECT 5
#define ORBX_STAGE_DESCRIBE 6

/* ---- stage-level operators (host buffers; each testable alone) ----------- */

/* d_Fast (src/cuda/Fast.cu:30-209) / OrientedFASTCPU::detect part 1
 * (src/orb_cpu.cpp:23-103): scores[h*w] float, 0 where not a corner. */
int orbx_fast_score(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, int threshold, int n,
                    float* scores);

/* NMS(scores, keypoints, nms_window, nfeatures, threshold)
 * (include/NMS.cuh:5, src/cuda/NMS.cu:21-161) with the CPU flavour's
 * deterministic row-major order and cap (src/orb_cpu.cpp:105-134).
 * *count = min(survivors, nfeatures); *total (optional) = survivors. */
int orbx_nms(orbx_ctx* ctx, const float* scores, int width, int height, int nms_window, int nfeatures,
             float threshold, orbx_keypoint* keypoints, int* count, int* total);

/* Fast(image, keypoints, threshold, n, nms_window, nfeatures) -> count
 * (include/Fast.cuh:5, src/cuda/Fast.cu:211-269); OrientedFAST::detect
 * (src/orb.cpp:22-27), OrientedFASTCPU::detect (src/orb_cpu.cpp:23-137). */
int orbx_fast(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, int threshold, int n,
              int nms_window, int nfeatures, orbx_keypoint* keypoints, int* count, int* total);

/* Orientations(image, keypoints, orientations, patch_size)
 * (include/Fast.cuh:6, src/cuda/Orientations.cu:22-97;
 * src/orb_cpu.cpp:139-183). */
int orbx_orientations(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                      const orbx_keypoint* keypoints, int nkp, int patch_size, float* orientations);

/* Brief(image, keypoints, orientations, descriptors, 256, 31)
 * (include/Brief.cuh:5, src/cuda/Brief.cu:40-136; src/orb_cpu.cpp:203-258). */
int orbx_brief(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride,
               const orbx_keypoint* keypoints, const float* orientations, int nkp, orbx_descriptor* descriptors);

/* HarrisScore(image, keypoints, scores, corner_window, k)
 * (include/HarrisScore.cuh:5, src/cuda/HarrisScore.cu:23-89; intent, see
 * DESIGN.md "Harris"). */
int orbx_harris(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                const orbx_keypoint* keypoints, int nkp, int window, float k, float* responses);

/* GaussianBlur1D(image, dst) (include/GaussianBlur.cuh:4,
 * src/cuda/GaussianBlur1D.cu:34-163): separable [1 4 6 4 1]/16, REFLECT_101. */
int orbx_blur5_sep(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, uint8_t* dst,
                   int dst_stride);
/* GaussianBlur(image, dst) (include/GaussianBlur.cuh:3,
 * src/cuda/GaussianBlur.cu:35-130): 5x5 /273, REFLECT_101. */
int orbx_blur5_273(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, uint8_t* dst,
                   int dst_stride);

/* conv2d(image, dst, kernel, kernel_size) (include/Convolution.cuh:5,
 * src/cuda/Convolution.cu:20-101): valid KxK correlation of a pre-padded u8
 * image, float accumulate, result rounded half-to-even and saturated to u8.
 * dst is (height-K+1) x (width-K+1), pitch width-K+1. */
int orbx_conv2d(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, const float* kernel,
                int kernel_size, uint8_t* dst);
/* GaussianBlurCUDA(image, dst, kernel_size) (include/GaussianBlur.hpp:6,
 * src/GaussianBlur.cpp:39-49). dst pitch = width. */
int orbx_gaussian_blur_conv(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride,
                            int kernel_size, uint8_t* dst);
/* createGaussianKernel(kernelSize, sigma) (src/GaussianBlur.cpp:7-37). */
int orbx_gaussian_kernel(int kernel_size, float sigma, float* kernel);
/* SobelCUDA(image, dst, dir) (include/Sobel.hpp:6, src/Sobel.cpp:18-32). */
int orbx_sobel(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, int dir, uint8_t* dst);

/* ORB::buildPyramid (src/orb.cpp:111-120; with blur src/orb_cpu.cpp:278-290):
 * writes level `level` (tightly packed, pitch = level width) to dst. */
int orbx_build_pyramid_level(orbx_ctx* ctx, const uint8_t* image, int width, int height, int stride, int level,
                             uint8_t* dst, int* level_w, int* level_h);

/* keep-top-N of src/orb.cpp:67-86 (intent): indices of the `keep` largest
 * responses ordered by (response desc, index asc). */
int orbx_select_top(orbx_ctx* ctx, const float* responses, int n, int keep, int32_t* indices, int* kept);

/* ---- next row: descriptor matching ------------------ */

/* flann->knnMatch(des1, des2, matches, 2) (src/feature_matching.cpp:166-168,
 * src/feature_tracking.cpp:203-204) as an EXACT brute-force Hamming 2-NN search
 * (the reference's FLANN LSH index is approximate).  idx/dist: nq x 2 (best,
 * second best; -1 when the train set has fewer descriptors); ties keep the lower
 * train index. */
int orbx_knn2(orbx_ctx* ctx, const orbx_descriptor* query, int nq, const orbx_descriptor* train, int nt,
              int32_t* idx, int32_t* dist);

/* knnMatch + the ratio test `m.distance < ratio * n.distance`
 * (src/feature_matching.cpp:172-181; ratio = 0.8 there), matches in query order.
 * dist1 may be NULL.  *count = number of matches; ORBX_ERR_CAPACITY if > capacity. */
int orbx_match_ratio(orbx_ctx* ctx, const orbx_descriptor* query, int nq, const orbx_descriptor* train, int nt,
                     double ratio, int32_t* query_idx, int32_t* train_idx, int32_t* dist1, int capacity, int* count);

/* Device-resident: matches frame i (query) against frame i+1 (train) for every
 * consecutive pair of the last batch -- the VO loop's get_matches() shape -- on
 * the batch's stream, straight from the result slots (no host round trip). */
int orbx_batch_match_consecutive(orbx_ctx* ctx, double ratio);
/* matches of pair `pair` (frames pair, pair+1) of the last orbx_batch_match_consecutive. */
int orbx_batch_match_fetch(orbx_ctx* ctx, int pair, int32_t* query_idx, int32_t* train_idx, int32_t* dist1,
                           int capacity, int* count);

/* ---- next row: pyramidal Lucas-Kanade tracking ---------
 * Replaces
 *   cv::calcOpticalFlowPyrLK(img1, img2, pts1, pts2, status, err, cv::Size(21,21), 3,
 *       cv::TermCriteria(COUNT + EPS, 30, 0.01))            src/feature_tracking.cpp:175-181
 * (flags = 0, minEigThreshold = 1e-4: the defaults the reference leaves in place).
 * prev / next: 8-bit gray images of the same size.  prev == NULL: the `next` image of
 * the previous call on this context is this call's `prev` (the reference's
 * `img1 = img2.clone()`, src/feature_tracking.cpp:112; its pyramid is still on the
 * device).  prev_pts_xy / next_pts_xy: n (x, y) float pairs; status: n bytes (1 =
 * tracked); err (optional): n floats, mean absolute window difference at level 0.
 * OpenCV is absent from the image this library was written in: the arithmetic
 * restates OpenCV 4.x's published algorithm (parity unpinned). */
int orbx_lk_track(orbx_ctx* ctx, const uint8_t* prev, int prev_stride, const uint8_t* next, int next_stride,
                  int width, int height, const float* prev_pts_xy, int n, float* next_pts_xy, uint8_t* status,
                  float* err, int win_size, int max_level, int max_iters, double epsilon);
/* number of pyramid levels calcOpticalFlowPyrLK would use for this geometry
 * (max_level + 1 unless a level would not be larger than the window); -1 on bad arguments */
int orbx_lk_pyramid_levels(int width, int height, int win_size, int max_level);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
