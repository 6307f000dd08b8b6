// orbx_internal.h -- shared between the HIP kernels (orbx_kernels.hip) and the
// C-ABI host layer (orbx_api.cpp).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbx.h"

// ---- HBM layout ------------------------------------------------------------
// A "pyramid frame" holds all levels of one input frame back to back:
//   level l at byte offset img_off, `h` rows of `pitch` bytes, pitch = W_l
//   rounded up to 64 (so every tile row starts 4-byte aligned and a 64-pixel
//   tile row never straddles the allocation), level offsets 256-B aligned.
// The NMS survivor mask of a level is h rows of `mask_wpr` 64-bit words
// (bit x&63 of word x>>6), all levels back to back at mask_off (in words).
// Candidate keypoints / Harris responses of a frame live in `cand_total`
// slots, level l owning [cand_off, cand_off + cap).
struct OrbxLevel {
  int32_t w, h, pitch;
  int32_t img_off;   // bytes, within a pyramid frame
  int32_t mask_wpr;  // u64 words per mask row
  int32_t mask_off;  // u64 words, within a frame's mask block
  int32_t cap;       // FAST cap (row-major)        src/orb.cpp:63
  int32_t quota;     // kept after selection        src/orb.cpp:62
  int32_t cand_off;  // first candidate slot
  int32_t xtab_off;  // first entry of this level's resize x-table
  int32_t ytab_off;  // first entry of this level's resize y-table
  float scale;       // (float)pow(scaleFactor, l)  src/orb.cpp:95
  int32_t out_off;   // first STATIC selection slot of this level = sum of the lower levels' quotas
  int32_t win8;      // resize: 1: the 4 source pairs of any aligned group of 4 outputs fit one 8-byte window; 3: a strip's source span fits the LDS staging rows of k_pyrblur; 0: neither (2-byte gathers)
  // 0: classic mask rows (bit x & 63 of word x >> 6).  > 0: STRIP layout of the streaming FAST kernel
  // (orbx_fast4.hip): 4 words per strip and row, word 4 s + q of a row holds pixels mask_strip_px * s + 64 q + bit
  // (a strip's first / last halo pixels are zero bits), so x = (xw >> 2) * mask_strip_px + (xw & 3) * 64 + bit
  int32_t mask_strip_px;
};
// pixel x of bit `b` of mask word `xw` of a row
__host__ __device__ inline int orbx_mask_x(const OrbxLevel& L, int xw, int b) {
  return L.mask_strip_px ? (xw >> 2) * L.mask_strip_px + (xw & 3) * 64 + b : xw * 64 + b;
}

struct OrbxPlan {
  int32_t nlevels;
  int32_t w0, h0;
  int32_t frame_bytes;  // pyramid frame stride (bytes)
  int32_t mask_words;   // mask stride per frame (u64 words)
  int32_t cand_total;   // candidate slots per frame
  int32_t out_cap;      // result slots per frame (sum of quotas)
  OrbxLevel L[ORBX_MAX_LEVELS];
};

// blockIdx.x -> (level, tile) map for one kernel's tile size
struct OrbxTileMap {
  int32_t begin[ORBX_MAX_LEVELS + 1];  // first tile id of each level (+ total)
  int32_t tiles_x[ORBX_MAX_LEVELS];
};

// band-major workgroup order of the FAST kernel (see decode_band)
#define ORBX_MAX_BANDS 64
// FAST early-exit state per frame (u64 words): tile-row statistics [level][band], then one
// "dead from band" word per level
#define ORBX_FAST_STAT_WORDS (ORBX_MAX_LEVELS * ORBX_MAX_BANDS + ORBX_MAX_LEVELS)
// feedback words of the top-rows-first pipeline (device, mirrored in pinned host memory by the batch's last kernel):
// [0] levels skipped, [1] levels produced (running totals), [2 + l] rows level l needed to fill its cap (maximum
// over the frames of the last batch; 0: not reported)
#define ORBX_FEEDBACK_WORDS (2 + ORBX_MAX_LEVELS)
// smallest FAST tile-row height the adaptive first pass may choose (orbx_api.cpp, adapt_tile_rows)
#define ORBX_MIN_TILE_H 16
struct OrbxBandMap {
  int32_t nbands;
  int32_t band_begin[ORBX_MAX_BANDS + 1];  // tiles PER FRAME before band b (+ total)
  int32_t tiles_x[ORBX_MAX_LEVELS];
  int32_t tiles_y[ORBX_MAX_LEVELS];
  int32_t tile_h[ORBX_MAX_LEVELS];       // rows per FAST tile of the level (balanced: ceil(h / tiles_y))
  int32_t xprefix[ORBX_MAX_LEVELS + 1];  // prefix sums of tiles_x
};

// One 64-byte record per workgroup, read with a single scalar load: everything a
// tile kernel needs to know about its tile.  Replaces the chains of dependent
// scalar loads that decoding blockIdx through the plan / tile maps costs at the
// start of every wave (~20 s_load round trips for the FAST kernel).
//   FAST table   : one entry per (tile row, level, tx) of ONE frame in band-major order
//                  (grid = frames x tiles, frame index dispatched fastest); `f` = rows per tile
//                  of this level.
//   pyramid      : one entry per (level, tx, ty) of ONE frame (blockIdx.y = frame); u0/u1/u2
//                  carry xtab_off / ytab_off / win8 and `f` the rows per wave.
//   blur         : one entry per (level, 256-px strip, row band): tx = strip, ty = first row,
//                  f = rows of the band.
//                  fused pyramid + blur, second pass of the top-rows-first pipeline: stat_index = first
//                  tile-row statistic of the level, mask_off = (FAST tile rows of the first pass) << 32 | cap,
//                  bit 62 set in ONE strip per frame (its wave reports how many levels of the frame were skipped).
//   img_off / mask_off are offsets inside one frame's pyramid / mask block.
struct OrbxTileDesc {
  int32_t l, tx, ty, f;
  int32_t w, h, pitch;
  int32_t u0;  // FAST: cap            pyramid: xtab_off
  int32_t u1;  // FAST: mask_wpr       pyramid: ytab_off
  int32_t u2;  // FAST: tiles_x        pyramid: win8
  uint32_t stat_index;  // FAST: first tile-row statistic of (frame, level)
  uint32_t pad;
  uint64_t img_off;   // bytes from the pyramid base
  uint64_t mask_off;  // u64 words from the mask base (FAST)
};
static_assert(sizeof(OrbxTileDesc) == 64, "one 64-byte scalar load per workgroup");

// 8-bit bilinear resize coefficient (OpenCV-style 11-bit fixed point)
struct OrbxResizeTap {
  int32_t ofs;     // source index (clamped)
  int16_t c0, c1;  // weights of src[ofs], src[ofs+1]; c0+c1 ~ 2048
};

// the levels whose lower rows the second pass of the top-rows-first pipeline may skip (orbx_api.cpp, enqueue_batch)
struct OrbxTopLevels {
  int32_t n;
  int32_t stat_index[ORBX_MAX_LEVELS];  // first tile-row statistic of the level
  int32_t rows[ORBX_MAX_LEVELS];        // FAST tile rows of the first pass
  int32_t cap[ORBX_MAX_LEVELS];
};

struct OrbxFastParams {
  int32_t threshold, n, nms_radius;
};

// tile geometry of the FAST/NMS kernel (orbx_fast.hip): 128 output pixels wide (two mask words per
// row); 34 dword columns x 7 row segments of walking threads, 7 rows per walk (at most 8: a flag byte per
// pixel column), so the score region of a tile has 49 rows and a tile 49 - 2 * nms_radius output rows
#define ORBX_FAST3_TW 128
#define ORBX_FAST3_K 7
constexpr int orbx_fast3_tile_h(int nms_radius) {
  return (256 / (ORBX_FAST3_TW / 4 + 2)) * ORBX_FAST3_K - 2 * nms_radius;
}
// streaming FAST kernel (orbx_fast4.hip): a wave owns a strip of 64 dwords; the outer `halo` dwords of a side are
// context for the ring (3 px) and the NMS window (R px) of the pixels next to them; tile rows as above
constexpr int orbx_fast4_halo(int nms_radius) { return (nms_radius + 3 + 3) / 4; }
constexpr int orbx_fast4_strip_lanes(int nms_radius) { return 64 - 2 * orbx_fast4_halo(nms_radius); }
constexpr int orbx_fast4_strips(int w, int nms_radius) {
  const int ndw = (w + 3) / 4, s = orbx_fast4_strip_lanes(nms_radius), n = (ndw - 2 * orbx_fast4_halo(nms_radius) + s - 1) / s;
  return n < 1 ? 1 : n;
}
// tile geometry of the blur kernel
#define ORBX_BLUR_TW 64
#define ORBX_BLUR_TH 16
// register-streaming separable blur (orbx_blur.hip): a wave owns a strip of 256 pixels (one aligned
// 256-byte segment per row) over a band of at most ORBX_BLUR3_RH rows
#define ORBX_BLUR3_TW 256
#define ORBX_BLUR3_RH 64
// k_blur4: 16 pixels per lane, a wave = a 256-px strip x 4 row bands of at most ORBX_BLUR4_RH rows
#define ORBX_BLUR4_TW 256
#define ORBX_BLUR4_RH 96
// fused pyramid + blur: the halo dwords are computed, not loaded, so lanes 0 / 63 are halo-only
#define ORBX_PYRBLUR_TW 248
// LDS staging of the source rows of the levels whose pairs do not fit the 8-byte window (scale > 2): bytes of a
// source row a strip may need.  (Measured per level of 1241x376, 256 frames: scale 2.1: 41 us staged / 56 us with
// 2-byte gathers, 2.5: 35 / 43, 3.0: 27 / 32, 3.5 (896 bytes): 30 / 29 -- the staged loads cost the texture
// addresser a cycle per four lane-dwords like any other, and at 8 x 104 bytes they are as many as the gathers'.)
#define ORBX_PYR_STAGE_BYTES 832
// rows per band of the fused kernel: the y taps of a band's input rows (rows + 6) sit one per lane
#define ORBX_PYRBLUR_RH 58
#define ORBX_PYRBLUR_RH_SMALL 12  // few frames per call: many short waves instead
// pyramid kernel: a wave owns 256 x 8 pixels (level 0 and the levels resized through 8-byte
// windows) or 256 x 4, a workgroup four times that; OrbxTileDesc::f carries the rows per wave
#define ORBX_PYR2_TW 256
#define ORBX_PYR2_TH 16  // smallest tile height (sizes the tile table pool)

#define ORBX_MAX_SELECT 4096  // largest per-level FAST cap the selection kernel holds in LDS (16 B per candidate)

// ---- pyramidal Lucas-Kanade tracking (orbx_lk.hip) ---------------------------
#define ORBX_LK_MAX_LEVELS 8
struct OrbxLkLevel {
  const uint8_t* img;    // w x h, row pitch `pitch`
  const int16_t* deriv;  // Scharr (dx, dy) pairs, w x h tight; NULL in the `next` pyramid
  int32_t w, h, pitch, pad;
};
struct OrbxLkPyr {
  OrbxLkLevel L[ORBX_LK_MAX_LEVELS];
  int32_t top;  // highest level present
  int32_t pad;
};
hipError_t orbx_launch_lk_pyrdown(hipStream_t s, const uint8_t* d_src, int sw, int sh, int spitch, uint8_t* d_dst,
                                  int dw, int dh, int dpitch);
hipError_t orbx_launch_lk_scharr(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch, int16_t* d_deriv);
hipError_t orbx_launch_lk_track(hipStream_t s, const OrbxLkPyr& prev, const OrbxLkPyr& next, int n,
                                const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, float* d_err, int win,
                                int max_iters, double eps2);

// ---- launchers (orbx_kernels.hip) ------------------------------------------
// All take the stream explicitly and never synchronise or allocate.
// d_tiles: tiles of ONE frame for 256 x 16 tiles
hipError_t orbx_launch_pyramid2(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int w0,
                                int h0, int n_frames, const uint8_t* d_in, int in_stride, size_t in_frame_stride,
                                const OrbxResizeTap* d_taps, uint8_t* d_pyr);
hipError_t orbx_launch_blur(hipStream_t s, const OrbxPlan& plan, const OrbxTileMap& tm, int n_frames,
                            const uint8_t* d_src, uint8_t* d_dst, int first_level, int kind);
// d_tiles: strip table of ONE frame (level, strip, first row, rows)
hipError_t orbx_launch_blur3(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int n_frames,
                             const uint8_t* d_src, uint8_t* d_dst, int first_level);
hipError_t orbx_launch_blur4(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int n_frames,
                             const uint8_t* d_src, uint8_t* d_dst, int first_level);
// d_tiles: strip table of ONE frame with the pyramid fields (u0 / u1 / u2 = xtab_off / ytab_off / win8)
hipError_t orbx_launch_pyrblur(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int w0, int h0,
                               int n_frames, const uint8_t* d_in, int in_stride, size_t in_frame_stride,
                               const OrbxResizeTap* d_taps, uint8_t* d_dst, int group = 0,
                               const unsigned long long* d_row_stat = nullptr, uint32_t* d_feedback = nullptr,
                               const OrbxTopLevels* top = nullptr, unsigned long long* d_zero_stat = nullptr);
// d_tiles: n_tiles OrbxTileDesc in band-major order (orbx_api.cpp: build_fast_tiles, tile height
// orbx_fast3_tile_h(fp.nms_radius)); d_scores: optional dense u16 score map of ONE frame (stage operator)
hipError_t orbx_launch_fast_nms(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames,
                                const uint8_t* d_pyr, int frame_bytes, int mask_words, OrbxFastParams fp,
                                unsigned long long* d_mask, uint16_t* d_scores,
                                unsigned long long* d_row_stat, int chunk_scale = 1);
// streaming FAST + NMS of the whole path: d_tiles = (strip, tile row) units of ONE frame in band-major order, masks in
// strip layout (OrbxLevel::mask_strip_px > 0)
hipError_t orbx_launch_fast4(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames, const uint8_t* d_pyr,
                             int frame_bytes, int mask_words, OrbxFastParams fp, unsigned long long* d_mask,
                             unsigned long long* d_row_stat);
hipError_t orbx_launch_compact(hipStream_t s, const OrbxPlan& plan, int n_frames,
                               const unsigned long long* d_mask, orbx_keypoint* d_cand, int32_t* d_cand_count,
                               int32_t* d_cand_total, int need_total);
hipError_t orbx_launch_harris_flat(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch,
                                   const orbx_keypoint* d_kps, int nkp, const float* d_gauss, int K, float kk,
                                   float* d_resp);
// fused compaction + Harris + selection, one workgroup per (level, frame)
hipError_t orbx_launch_level_select(hipStream_t s, const OrbxPlan& plan, int n_frames, int mode,
                                    const unsigned long long* d_mask, const uint8_t* d_pyr, const float* d_gauss,
                                    int window, float k, orbx_keypoint* d_sel_lkp, float* d_sel_resp,
                                    int32_t* d_sel_count, uint32_t* d_need = nullptr);
// fused kernel or the three spread kernels, chosen by shape (force: 0 fused, 1 spread, -1 auto);
// d_cand / d_cresp: cand_total slots per frame, d_ncand: nlevels per frame (spread path only)
hipError_t orbx_launch_level_select_auto(hipStream_t s, const OrbxPlan& plan, int n_frames, int mode, int force,
                                         const unsigned long long* d_mask, const uint8_t* d_pyr,
                                         const float* d_gauss, int window, float k, uint32_t* d_cand,
                                         int32_t* d_ncand, float* d_cresp, orbx_keypoint* d_sel_lkp,
                                         float* d_sel_resp, int32_t* d_sel_count, uint32_t* d_need = nullptr);
// device-visible addresses of the compact sections of a result block's pinned host mirror (all null: not wanted)
struct OrbxHostRecord {
  int32_t* counts;
  uint32_t* kp16;
  float* angle;
  orbx_descriptor* desc;
};
hipError_t orbx_launch_describe(hipStream_t s, const OrbxPlan& plan, int n_frames, const uint8_t* d_pyr,
                                int patch_size, const int32_t* d_sel_count, const orbx_keypoint* d_sel_lkp,
                                const float* d_sel_resp, int32_t* d_out_count, orbx_keypoint* d_out_lkp,
                                float* d_out_resp, int32_t* d_out_level, orbx_keypoint* d_out_kp, uint32_t* d_out_kp16,
                                float* d_out_angle, orbx_descriptor* d_out_desc, const uint32_t* d_feedback = nullptr,
                                uint32_t* h_feedback = nullptr, const OrbxHostRecord* host_record = nullptr);

// stage-level helpers on plain (single-image, arbitrary pitch) buffers
hipError_t orbx_launch_describe_flat(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch,
                                     const orbx_keypoint* d_kps, int nkp, int patch_size, int use_given_angles,
                                     int do_brief, float* d_angles, orbx_descriptor* d_desc);
hipError_t orbx_launch_nms_f32(hipStream_t s, const float* d_scores, int w, int h, int radius, float threshold,
                               unsigned long long* d_mask, int mask_wpr);
hipError_t orbx_launch_conv2d(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch, const float* d_kernel,
                              int K, int reflect_pad, uint8_t* d_dst, int dst_pitch);
hipError_t orbx_launch_select_flat(hipStream_t s, const float* d_resp, int n, int keep, int32_t* d_idx);

// 256-bit Hamming 2-NN + ratio test over `npairs` (query set, train set) pairs
hipError_t orbx_launch_knn2(hipStream_t s, int npairs, int max_nq, const orbx_descriptor* d_q, const int32_t* d_qcount,
                            size_t qstride, const orbx_descriptor* d_t, const int32_t* d_tcount, size_t tstride,
                            double ratio, int32_t* d_idx, int32_t* d_dist, int32_t* d_match, size_t ostride);
