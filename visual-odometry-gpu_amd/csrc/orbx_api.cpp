// orbx_api.cpp -- C-ABI host layer of liborbx.so (see include/orbx.h).
//
// Owns the context (device memory pools, stream, per-size plan), validates
// arguments, sequences the kernel launches of orbx_kernels.hip and moves
// results.  There is NO CPU fallback anywhere in this file: if the HIP
// runtime or a gfx950 device is missing every entry point fails loudly with
// ORBX_ERR_NO_DEVICE / ORBX_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "orbx_internal.h"

namespace {

thread_local std::string g_create_error;

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// result block of a batch: one device allocation + one pinned host mirror so
// a whole batch comes back with a single D2H copy
// sections: counts | kp | angle | desc || lkp | resp | level -- what the reference's own output consists of
// (keypoints, orientations, descriptors: include/orb.hpp:37) first, so that orbx_batch_prefetch_compact moves one
// contiguous prefix of `compact` bytes
struct OutLayout {
  size_t counts, kp16, kp, lkp, angle, resp, level, desc, compact, total;
};

OutLayout make_out_layout(int n, int cap) {
  OutLayout o;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t r = off;
    off = align_up_sz(off + bytes, 256);
    return r;
  };
  const size_t e = (size_t)n * (size_t)cap;
  // the compact record first (orbx_batch_prefetch_compact copies [0, compact)): 40 bytes per slot
  o.counts = take(sizeof(int32_t) * (size_t)n);
  o.kp16 = take(sizeof(uint32_t) * e);
  o.angle = take(sizeof(float) * e);
  o.desc = take(sizeof(orbx_descriptor) * e);
  o.compact = off;
  o.kp = take(sizeof(orbx_keypoint) * e);
  o.lkp = take(sizeof(orbx_keypoint) * e);
  o.resp = take(sizeof(float) * e);
  o.level = take(sizeof(int32_t) * e);
  o.total = off;
  return o;
}

}  // namespace

// what a captured launch sequence depends on (run_batch)
struct OrbxGraphKey {
  const uint8_t* d_frames;
  size_t frame_stride;
  int n, w, h, row_stride, early, plan_serial, block;  // early: switches (early exit, fusion, two passes)
  bool operator==(const OrbxGraphKey& o) const {
    return d_frames == o.d_frames && frame_stride == o.frame_stride && n == o.n && w == o.w && h == o.h &&
           row_stride == o.row_stride && early == o.early && plan_serial == o.plan_serial && block == o.block;
  }
};

struct orbx_ctx {
  orbx_params p{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // geometry for the current frame size, and for the largest size (capacity)
  OrbxPlan plan{};
  OrbxPlan plan_max{};
  OrbxTileMap tm_blur{};  // 5x5 /273 variant (LDS tile kernel)
  OrbxBandMap bm_fast{};
  unsigned long long* d_row_stat = nullptr;
  // per-workgroup tile descriptor tables (see OrbxTileDesc)
  OrbxTileDesc* d_tiles_fast = nullptr;  // one frame, band-major
  size_t tiles_fast_capacity = 0;
  int fast_tiles_count = 0;
  OrbxTileDesc* d_tiles_blur = nullptr;
  OrbxTileDesc* d_tiles_pyr2 = nullptr;
  OrbxTileDesc* d_tiles_pyrblur = nullptr;  // fused pyramid + blur strips
  int pyrblur_tiles_count = 0;
  // the same strips cut into short row bands: few frames per call (the reference's one-frame call shape)
  // fill the chip only with many short waves, where a large batch wants few tall ones
  OrbxTileDesc* d_tiles_pyrblur_small = nullptr;
  int pyrblur_small_count = 0;
  // top-rows-first pipeline: the strips of the first pass and of the second one
  OrbxTileDesc* d_tiles_pyrblur_top = nullptr;
  OrbxTileDesc* d_tiles_pyrblur_rest = nullptr;
  int pyrblur_top_count = 0, pyrblur_rest_count = 0;
  OrbxTopLevels top_levels{};  // the levels the second pass may skip
  size_t tiles_frame_capacity = 0, tiles_small_capacity = 0;
  int blur_tiles_count = 0, pyr2_tiles_count = 0;
  DevBuf s_tiles;  // stage-API tables
  std::vector<OrbxResizeTap> h_taps;
  int plan_w = 0, plan_h = 0;

  // pools (sized for max_batch frames of max_width x max_height)
  uint8_t* d_in = nullptr;  // staged host frames, tight pitch
  uint8_t* d_pyr = nullptr;
  uint8_t* d_pyr_blur = nullptr;
  unsigned long long* d_mask = nullptr;
  orbx_keypoint* d_cand = nullptr;
  int32_t* d_cand_count = nullptr;
  int32_t* d_cand_total = nullptr;
  float* d_resp = nullptr;
  // captured launch sequences of the most recent batch shapes (run_batch), round-robin replacement
  static constexpr int kGraphs = 16;  // (input, result block, lane) triples: 8 resident inputs over 4 blocks x 2 lanes
  hipGraphExec_t g_exec[kGraphs] = {};
  OrbxGraphKey g_key[kGraphs] = {};
  int g_next = 0;
  int plan_serial = 0;  // bumped whenever set_plan rebuilds the plan / tables
  uint32_t* d_lcand = nullptr;  // spread selection: packed candidates, their responses, counts
  float* d_lresp = nullptr;
  int32_t* d_lcount = nullptr;
  OrbxResizeTap* d_taps = nullptr;
  size_t taps_capacity = 0;
  float* d_gauss = nullptr;
  // Result blocks.  A ring of kBlocks, used in turn by consecutive batches, each with a pinned host
  // mirror: the D2H copy of batch i (orbx_batch_prefetch, on its own copy stream) overlaps the
  // kernels of batch i+1, which write the other block.  d_out / h_out / out_layout / last_n always
  // describe the block of the most recent batch.
  // (a ring of kBlocks blocks: with two, the copy of batch i -- about as long as a step at 256 frames per batch --
  // had to finish before batch i + 2 could start; with four it overlaps two following batches)
  static constexpr int kBlocks = 4;
  uint8_t* d_outb[kBlocks] = {};
  uint8_t* h_outb[kBlocks] = {};
  uint8_t* h_outb_dev[kBlocks] = {};   // the device-visible addresses of the pinned mirrors
  int host_results = 0;                // orbx_set_host_results: the describe kernel writes the compact record to the mirror
  bool host_written[kBlocks] = {};     // ... and did so for the batch in this block
  OutLayout layoutb[kBlocks] = {};
  int nb[kBlocks] = {};    // frames in the block (0: never written)
  int capb[kBlocks] = {1, 1, 1, 1};  // slots per frame the block was written with
  bool copy_pending[kBlocks] = {};  // an asynchronous D2H of the block has been enqueued (ev_copied)
  bool copy_compact[kBlocks] = {};  // ... of its compact prefix only (orbx_batch_prefetch_compact)
  int blk = 0;
  int next_lane = 1;  // pipelined mode: the lane of the next batch (alternates)
  hipStream_t cstream = nullptr;
  hipEvent_t ev_done[kBlocks] = {}, ev_copied[kBlocks] = {};
  uint8_t* d_out = nullptr;
  uint8_t* h_out = nullptr;  // pinned mirror
  OutLayout out_layout{};
  int out_cap = 0;  // slot capacity of the pool (plan_max.out_cap)
  // Pipelined batches (orbx_set_pipelined_batches): two LANES, each with its own stream and its own working pools
  // (pyramids, mask, statistics, candidates); consecutive device-resident batches alternate between them -- batch
  // k uses lane k & 1 = its result block -- so the kernels of one batch overlap the tails and the nearly empty
  // launches of the other.  lane_pool[0] / lane_stream[0] are the context's own pools / stream; the d_* members
  // above always point at the pools of the most recent batch's lane.
  struct LanePool {
    uint8_t *d_pyr = nullptr, *d_pyr_blur = nullptr;
    unsigned long long *d_mask = nullptr, *d_row_stat = nullptr;
    orbx_keypoint* d_cand = nullptr;
    int32_t *d_cand_count = nullptr, *d_cand_total = nullptr, *d_lcount = nullptr;
    float *d_resp = nullptr, *d_lresp = nullptr;
    uint32_t* d_lcand = nullptr;
  };
  LanePool lane_pool[2];
  hipStream_t lane_stream[2] = {nullptr, nullptr};
  // Stream order is the only ordering inside a lane.  A batch that comes to a lane's pools, or to a result block,
  // on ANOTHER stream than their previous user (a caller's stream, the other lane) first makes its stream wait for
  // that user's event: ev_pool[k] / pool_stream[k] for the pools of lane k, ev_done[b] / blk_stream[b] for block b.
  hipEvent_t ev_pool[2] = {nullptr, nullptr};
  hipStream_t pool_stream[2] = {nullptr, nullptr};
  hipStream_t blk_stream[kBlocks] = {};
  bool pipelined = false;
  int last_n = 0;
  bool last_two_pass = false;  // the last batch built its pyramid top rows first (enqueue_batch)
  hipStream_t last_stream = nullptr;

  // stage-API scratch (grown on demand; never touched by the batched path)
  DevBuf s_img_a, s_img_b, s_f32, s_u16, s_mask, s_kps, s_f32b, s_desc, s_i32, s_kern;
  DevBuf m_q, m_t, m_idx, m_dist, m_match, m_cnt;  // matcher (stage API and batch)
  // Lucas-Kanade tracker: two image pyramids (ping-pong: the `next` of one call is the
  // `prev` of the following one), the derivative pyramid of the current `prev`, point buffers
  DevBuf lk_img[2], lk_deriv, lk_io;  // lk_io: prev points | next points | err | status, one block
  void* lk_host = nullptr;            // pinned mirror of lk_io (one H2D + one D2H per call)
  size_t lk_host_bytes = 0;
  int lk_w = 0, lk_h = 0, lk_top = -1, lk_win = 0, lk_last = -1;  // lk_last: buffer holding the last `next`
  int match_pairs = 0;

  int timing = 0;  // 0 off, 1 all stages, 2 blur + fast only
  int fast_early = 1;
  int blur_impl = 2;  // ORBX_BLUR_IMPL, read at creation (launch_blur_auto)
  int fast_impl = 4;  // 4: streaming kernel (orbx_fast4.hip, the default), 3: LDS tile kernel (orbx_fast.hip); ORBX_FAST_IMPL, read at creation
  int fuse = 1;  // pyramid + blur in one kernel when blur runs on every level (orbx_set_fused_pyramid_blur)
  // Top-rows-first pipeline (enqueue_batch): 0 never, 1 whenever eligible, 2 adaptive -- the second pass
  // counts the (frame, level)s it skipped / had to produce (d_feedback, running totals, written to the pinned
  // h_feedback by the last kernel of every two-pass batch and read WITHOUT waiting at the start of later ones); while
  // fewer than a quarter are skipped the batches run in one pass, and every 128th one probes again.
  int top_mode = 2;
  bool top_on = true;          // the adaptive verdict
  int top_single_batches = 0;  // one-pass batches since the verdict turned negative
  uint32_t* d_feedback = nullptr;
  volatile uint32_t* h_feedback = nullptr;
  uint32_t feedback_seen[2] = {0, 0};
  // adaptive first pass (adapt_tile_rows): rows each level needed to fill its cap -- maximum of the current and of
  // the previous observation window --, the tile-row heights chosen from them (0: the default), bookkeeping
  uint32_t need_cur[ORBX_MAX_LEVELS] = {}, need_prev[ORBX_MAX_LEVELS] = {};
  int tile_h_pref[ORBX_MAX_LEVELS] = {};
  int need_batches = 0, need_window = 2, retiles = 0, learn_w = 0, learn_h = 0;
  bool prefs_applied = false;  // the current tile tables were built with tile_h_pref
  // ring of event sets: one per timed batched call, so that several calls can be
  // in flight before their stage times are read (no host sync between steps)
  // (slots ORBX_NUM_STAGE_TIMES + 1, + 2: the boundaries inside the top-rows-first pipeline)
  hipEvent_t evr[ORBX_EVENT_SETS][ORBX_NUM_STAGE_TIMES + 3] = {};
  int ev_mode[ORBX_EVENT_SETS] = {};
  bool ev_split[ORBX_EVENT_SETS] = {};
  long long ev_calls = 0;  // timed batched calls so far
  hipEvent_t ev[2] = {};   // orbx_bench_stage
};

namespace {

// Every entry point that takes a context runs on the context's device, whatever the caller's
// current device is (another context's, torch.cuda.set_device, ...), and leaves the caller's
// current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const orbx_ctx* c) {
    if (!c) return;
    enter(c->device);
  }
  explicit DeviceGuard(int dev) { enter(dev); }
  void enter(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(orbx_ctx* c, int status, const std::string& msg) {
  if (c)
    c->err = msg;
  else
    g_create_error = msg;
  return status;
}

#define HIPCHK(c, expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return fail((c), ORBX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)

int ensure(orbx_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return ORBX_OK;
  if (b.p) {  // nothing in flight may still use the old allocation: the context's stream, both lanes, a caller's stream
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (hipStream_t ls : c->lane_stream)
      if (ls && ls != c->stream) HIPCHK(c, hipStreamSynchronize(ls));
    if (c->last_stream && c->last_stream != c->stream) HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  bytes = align_up_sz(std::max<size_t>(bytes, 256), 256);
  HIPCHK(c, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return ORBX_OK;
}

// ---- geometry (src/orb.cpp:62, :95, :117-118) ------------------------------

float level_scale(float sf, int l) { return (float)std::pow((double)sf, (double)l); }

void level_size(int w0, int h0, float sf, int l, int* wl, int* hl) {
  if (l == 0) {
    *wl = w0;
    *hl = h0;
    return;
  }
  const float scale = level_scale(sf, l);
  *wl = (int)std::round((double)((float)w0 / scale));
  *hl = (int)std::round((double)((float)h0 / scale));
}

int level_quota(int nfeatures, float sf, int nlevels, int l) {
  // int * ((float - float) / (int - double)) * double, truncated to int
  const float inv = 1 / sf;
  const float num = 1 - inv;
  const double den = 1 - std::pow((double)inv, (double)nlevels);
  return (int)(nfeatures * ((double)num / den) * std::pow((double)inv, (double)l));
}

void make_tilemap(const OrbxPlan& plan, int tw, int th, bool use_pitch, OrbxTileMap* tm) {
  int acc = 0;
  for (int l = 0; l < plan.nlevels; l++) {
    const int wcols = use_pitch ? plan.L[l].pitch : plan.L[l].w;
    const int tx = (wcols + tw - 1) / tw, ty = (plan.L[l].h + th - 1) / th;
    tm->begin[l] = acc;
    tm->tiles_x[l] = tx;
    acc += tx * ty;
  }
  for (int l = plan.nlevels; l <= ORBX_MAX_LEVELS; l++) tm->begin[l] = acc;
}

// band-major order of the FAST tiles (levels shrink with the level index, so the
// levels that have a tile row b are always a prefix of the level list)
// strips: the units of the streaming kernel of the whole path (orbx_fast4.hip: a wave per 64-dword strip and tile
// row) instead of the 128-pixel tiles of the LDS tile kernel (stage operators)
int make_bandmap(const OrbxPlan& plan, int nms_radius, OrbxBandMap* bm, std::string* why, bool strips = false,
                 const int* pref_h = nullptr) {
  std::memset(bm, 0, sizeof(*bm));
  const int tw = ORBX_FAST3_TW, th = orbx_fast3_tile_h(nms_radius);
  int nb = 0;
  for (int l = 0; l < plan.nlevels; l++) {
    bm->tiles_x[l] = strips ? orbx_fast4_strips(plan.L[l].w, nms_radius) : (plan.L[l].w + tw - 1) / tw;
    bm->tiles_y[l] = (plan.L[l].h + th - 1) / th;
    bm->tile_h[l] = (plan.L[l].h + bm->tiles_y[l] - 1) / bm->tiles_y[l];  // balanced tile rows
    // pref_h[l] > 0: SHORTER tile rows for this level (the adaptive first pass of the top-rows-first pipeline:
    // adapt_tile_rows) -- never more tile rows than ORBX_MAX_BANDS or than the level above has (band-major order)
    if (pref_h && pref_h[l] > 0 && pref_h[l] < bm->tile_h[l]) {
      int hh = std::max(pref_h[l], ORBX_MIN_TILE_H);
      const int most = l > 0 ? std::min(bm->tiles_y[l - 1], ORBX_MAX_BANDS) : ORBX_MAX_BANDS;
      while ((plan.L[l].h + hh - 1) / hh > most) hh++;
      if (hh < bm->tile_h[l]) {
        bm->tile_h[l] = hh;
        bm->tiles_y[l] = (plan.L[l].h + hh - 1) / hh;
      }
    }
    bm->xprefix[l + 1] = bm->xprefix[l] + bm->tiles_x[l];
    if (l > 0 && bm->tiles_y[l] > bm->tiles_y[l - 1]) {
      *why = "pyramid levels must not grow with the level index";
      return ORBX_ERR_UNSUPPORTED;
    }
    nb = std::max(nb, bm->tiles_y[l]);
  }
  if (nb > ORBX_MAX_BANDS) {
    *why = "image taller than ORBX_MAX_BANDS FAST tile rows";
    return ORBX_ERR_UNSUPPORTED;
  }
  bm->nbands = nb;
  int acc = 0;
  for (int b = 0; b < nb; b++) {
    bm->band_begin[b] = acc;
    for (int l = 0; l < plan.nlevels; l++)
      if (bm->tiles_y[l] > b) acc += bm->tiles_x[l];
  }
  for (int b = nb; b <= ORBX_MAX_BANDS; b++) bm->band_begin[b] = acc;
  return ORBX_OK;
}

// FAST tiles of ONE frame in band-major order (the kernel's grid is frames x tiles with
// the frame index dispatched fastest, so tile row b of every frame runs before tile row
// b+1 of any frame).  Tile rows >= first_band only; a workgroup owns `strip` tiles of a row.
void build_fast_tiles(const OrbxPlan& plan, const OrbxBandMap& bm, int first_band, int strip,
                      std::vector<OrbxTileDesc>* out) {
  out->clear();
  for (int b = first_band; b < bm.nbands; b++)
    for (int l = 0; l < plan.nlevels; l++) {
      if (bm.tiles_y[l] <= b) continue;
      const OrbxLevel& L = plan.L[l];
      for (int tx = 0; tx < bm.tiles_x[l]; tx += strip) {
        OrbxTileDesc d{};
        d.l = l;
        d.tx = tx;
        d.ty = b;
        d.f = bm.tile_h[l];
        d.w = L.w;
        d.h = L.h;
        d.pitch = L.pitch;
        d.u0 = L.cap;
        d.u1 = L.mask_wpr;
        d.u2 = bm.tiles_x[l];
        d.stat_index = (uint32_t)(l * ORBX_MAX_BANDS);
        d.img_off = (uint64_t)L.img_off;
        d.mask_off = (uint64_t)L.mask_off;
        out->push_back(d);
      }
    }
}

// tiles of ONE frame, level-major, for the blur / pyramid kernels (blockIdx.y = frame)
void build_frame_tiles(const OrbxPlan& plan, int tw, int th, bool pyramid_fields, std::vector<OrbxTileDesc>* out) {
  out->clear();
  for (int l = 0; l < plan.nlevels; l++) {
    const OrbxLevel& L = plan.L[l];
    // pyramid tiles: 8 rows per wave where a lane keeps only 4 registers per row in flight
    // (level 0 copy, 8-byte-window levels), else 4
    const int rpw = pyramid_fields ? ((l == 0 || L.win8 == 1) ? 8 : 4) : 0;
    if (pyramid_fields) th = 4 * rpw;
    const int ntx = (L.pitch + tw - 1) / tw, nty = (L.h + th - 1) / th;
    for (int ty = 0; ty < nty; ty++)
      for (int tx = 0; tx < ntx; tx++) {
        OrbxTileDesc d{};
        d.l = l;
        d.tx = tx;
        d.ty = ty;
        d.f = rpw;
        d.w = L.w;
        d.h = L.h;
        d.pitch = L.pitch;
        if (pyramid_fields) {
          d.u0 = L.xtab_off;
          d.u1 = L.ytab_off;
          d.u2 = L.win8 == 1;  // (k_pyramid2 knows the one-window mode only)
        }
        d.img_off = (uint64_t)L.img_off;
        out->push_back(d);
      }
  }
}

// strips of the streaming blur for ONE frame: per level ceil(pitch / 256) strips x balanced row bands of
// at most ORBX_BLUR3_RH rows (one wave each; the 4 warm-up rows of the vertical pass are per band)
void build_blur_tiles(const OrbxPlan& plan, std::vector<OrbxTileDesc>* out) {
  out->clear();
  for (int l = 0; l < plan.nlevels; l++) {
    const OrbxLevel& L = plan.L[l];
    const int ntx = (L.pitch + ORBX_BLUR3_TW - 1) / ORBX_BLUR3_TW;  // the padding bytes are (re)written as zeros
    const int nb = (L.h + ORBX_BLUR3_RH - 1) / ORBX_BLUR3_RH, rows = (L.h + nb - 1) / nb;
    for (int b = 0; b < nb; b++)
      for (int tx = 0; tx < ntx; tx++) {
        OrbxTileDesc d{};
        d.l = l;
        d.tx = tx;
        d.ty = b * rows;
        d.f = std::min(rows, L.h - b * rows);
        d.w = L.w;
        d.h = L.h;
        d.pitch = L.pitch;
        d.img_off = (uint64_t)L.img_off;
        if (d.f > 0) out->push_back(d);
      }
  }
}

// units of k_blur4 for ONE frame: per level ceil(pitch / 256) strips x waves of FOUR row bands each (f = rows per
// band: a level's height spread over the fewest waves whose bands stay within ORBX_BLUR4_RH rows)
void build_blur4_tiles(const OrbxPlan& plan, std::vector<OrbxTileDesc>* out) {
  out->clear();
  for (int l = 0; l < plan.nlevels; l++) {
    const OrbxLevel& L = plan.L[l];
    const int ntx = (L.pitch + ORBX_BLUR4_TW - 1) / ORBX_BLUR4_TW;  // the padding bytes are (re)written as zeros
    const int nwv = (L.h + 4 * ORBX_BLUR4_RH - 1) / (4 * ORBX_BLUR4_RH), rows = (L.h + 4 * nwv - 1) / (4 * nwv);
    for (int wv = 0; wv < nwv; wv++)
      for (int tx = 0; tx < ntx; tx++) {
        OrbxTileDesc d{};
        d.l = l;
        d.tx = tx;
        d.ty = wv * 4 * rows;
        d.f = rows;
        d.w = L.w;
        d.h = L.h;
        d.pitch = L.pitch;
        d.img_off = (uint64_t)L.img_off;
        if (d.ty < L.h) out->push_back(d);
      }
  }
}

// strips of the fused pyramid + blur kernel for ONE frame: 248-px strips (the halo dwords are
// computed by lanes 0 / 63) x balanced row bands, with the level's resize-table fields.
// part 0: every row.  Top-rows-first pipeline (enqueue_batch): part 1 = the rows the FAST tiles of the
// first `top_rows` tile rows and the descriptors of their keypoints can read -- rows below
// top_rows * tile_h + ORBX_TOP_MARGIN -- and part 2 = the rest, whose strips carry what the kernel's skip
// test needs (stat_index, mask_off = tile rows of the first pass << 32 | cap).
#define ORBX_TOP_MARGIN 21  // a descriptor reaches DESC_R = 20 rows below its keypoint (orbx_kernels.hip); Harris, FAST less
int pyrblur_first_pass_rows(const OrbxPlan& plan, const OrbxBandMap& bm, int l, int top_rows) {
  if (top_rows <= 0 || bm.tiles_y[l] <= top_rows) return plan.L[l].h;
  return std::min(plan.L[l].h, top_rows * bm.tile_h[l] + ORBX_TOP_MARGIN);
}
void build_pyrblur_tiles(const OrbxPlan& plan, int max_rows, std::vector<OrbxTileDesc>* out, bool heavy_first = false,
                         int part = 0, const OrbxBandMap* bm = nullptr, int top_rows = 0) {
  out->clear();
  bool have_reporter = false;
  for (int l = 0; l < plan.nlevels; l++) {
    const OrbxLevel& L = plan.L[l];
    // dwords that hold image pixels: 62 per strip, one more in the first and in the last strip (their
    // outer neighbour is a reflection, not another strip's dword).  The padding dwords beyond are not
    // written by this kernel: they are zeroed when the plan is set.
    const int dw = (L.w + 3) / 4;
    const int ntx = dw <= 64 ? 1 : (dw - 2 + 61) / 62;
    const int split = part == 0 ? L.h : pyrblur_first_pass_rows(plan, *bm, l, top_rows);
    const int r0 = part == 2 ? split : 0, r1 = part == 1 ? split : L.h;
    if (r1 <= r0) continue;
    const int nb = (r1 - r0 + max_rows - 1) / max_rows, rows = (r1 - r0 + nb - 1) / nb;
    for (int b = 0; b < nb; b++)
      for (int tx = 0; tx < ntx; tx++) {
        OrbxTileDesc d{};
        d.l = l;
        d.tx = tx;
        d.ty = r0 + b * rows;
        d.f = std::min(rows, r1 - d.ty);
        d.w = L.w;
        d.h = L.h;
        d.pitch = L.pitch;
        d.u0 = L.xtab_off;
        d.u1 = L.ytab_off;
        d.u2 = L.win8;
        d.pad = (uint32_t)ntx;
        d.img_off = (uint64_t)L.img_off;
        if (part == 2) {
          d.stat_index = (uint32_t)(l * ORBX_MAX_BANDS);
          d.mask_off = ((uint64_t)(uint32_t)std::min(top_rows, bm->tiles_y[l]) << 32) | (uint32_t)L.cap;
          if (b == 0 && tx == 0 && !have_reporter) {  // this strip's wave reports the verdicts of all the frame's levels
            d.mask_off |= 1ull << 62;
            have_reporter = true;
          }
        }
        if (d.f > 0) out->push_back(d);
      }
  }
  if (heavy_first) {
    // estimated instructions per strip row: level 0 copies, the 8-byte-window levels resize, the others gather
    auto cost = [](const OrbxTileDesc& d) { return (d.f + 4) * (d.l == 0 ? 35 : d.u2 == 1 ? 80 : 90); };
    std::stable_sort(out->begin(), out->end(),
                     [&](const OrbxTileDesc& a, const OrbxTileDesc& b) { return cost(a) > cost(b); });
  }
}

// ORBX_PYR_GROUP=g: frames per dispatch group of the fused pyramid + blur kernel (0: frame-major grid)
int pyr_group_env() {
  static const int v = [] {
    const char* e = getenv("ORBX_PYR_GROUP");
    return e ? atoi(e) : 32;
  }();
  return v;
}

// ORBX_TOP_ROWS=k: FAST tile rows of the first pass of the top-rows-first pipeline (0: one pass)
int top_rows_env() {
  static const int v = [] {
    const char* e = getenv("ORBX_TOP_ROWS");
    return e ? atoi(e) : 2;
  }();
  return v < 0 ? 0 : v;
}

// Which FAST + NMS kernel the whole path runs: the register-streaming kernel (orbx_fast4.hip; the default) or, with
// ORBX_FAST_IMPL=3, the LDS tile kernel (orbx_fast.hip; always the stage operators').  Same results.  With every tile
// working they take the same time (+-2 %: 6.5 % fewer vector instructions against 14 instead of 24 waves per CU);
// in production -- the short tile rows of the adaptive first pass, most units exiting early -- the streaming kernel
// has less to do per unit (no tile fill, no barriers): same-box A/B 519 k vs 499 k frames/s (tools/ab_fast2.sh).
// Read when a context is created (A/B timing in one process).
int fast_impl_env() {
  const char* e = getenv("ORBX_FAST_IMPL");
  return e && atoi(e) == 3 ? 3 : 4;
}

int build_plan(const orbx_params& p, int w0, int h0, OrbxPlan* plan, std::string* why, int fast_impl = 3) {
  std::memset(plan, 0, sizeof(*plan));
  plan->nlevels = p.nlevels;
  plan->w0 = w0;
  plan->h0 = h0;
  size_t img_off = 0, mask_off = 0;
  int cand_off = 0, out_cap = 0, xt = 0;
  for (int l = 0; l < p.nlevels; l++) {
    OrbxLevel& L = plan->L[l];
    level_size(w0, h0, p.scale_factor, l, &L.w, &L.h);
    if (L.w < 8 || L.h < 8) {
      *why = "pyramid level " + std::to_string(l) + " is smaller than 8x8 (" + std::to_string(L.w) + "x" +
             std::to_string(L.h) + ")";
      return ORBX_ERR_UNSUPPORTED;
    }
    L.pitch = align_up(L.w, 64);
    L.img_off = (int32_t)img_off;
    img_off = align_up_sz(img_off + (size_t)L.pitch * L.h, 256);
    // the whole path's FAST kernel writes its survivor masks in strip layout (orbx_fast4.hip)
    if (fast_impl == 4) {
      L.mask_strip_px = 4 * orbx_fast4_strip_lanes(p.nms_window / 2);
      L.mask_wpr = 4 * orbx_fast4_strips(L.w, p.nms_window / 2);
    } else {
      L.mask_wpr = (L.w + 63) / 64;
    }
    L.mask_off = (int32_t)mask_off;
    mask_off += (size_t)L.mask_wpr * L.h;
    int quota;
    if (p.select_mode == ORBX_SELECT_ROWMAJOR && p.nlevels == 1)
      quota = p.nfeatures;  // OrientedFASTCPU::detect cap (src/orb_cpu.cpp:110)
    else
      quota = level_quota(p.nfeatures, p.scale_factor, p.nlevels, l);
    if (quota < 0) quota = 0;
    L.quota = quota;
    L.cap = p.select_mode == ORBX_SELECT_HARRIS ? 2 * quota : quota;  // src/orb.cpp:63
    if (L.cap > ORBX_MAX_SELECT) {
      *why = "per-level FAST cap " + std::to_string(L.cap) + " exceeds ORBX_MAX_SELECT";
      return ORBX_ERR_UNSUPPORTED;
    }
    L.cand_off = cand_off;
    cand_off += L.cap;
    L.out_off = out_cap;
    out_cap += quota;
    L.scale = level_scale(p.scale_factor, l);
    // x table first (padded to a multiple of 4 entries = 32 bytes so that a thread's
    // four taps are two aligned 16-byte loads), then the y table
    L.xtab_off = xt;
    L.ytab_off = xt + (l == 0 ? 0 : align_up(L.w, 4));
    xt += (l == 0 ? 0 : align_up(L.w, 4) + align_up(L.h, 4));
    if (img_off > 0x7fffffffull) {
      *why = "pyramid frame exceeds 2 GiB";
      return ORBX_ERR_UNSUPPORTED;
    }
  }
  plan->frame_bytes = (int32_t)img_off;
  plan->mask_words = (int32_t)mask_off;
  // (the selection output lives in the candidate pools, at the result blocks' stride: in the row-major mode, where a
  // level's cap is its quota, the rounded-up slot count below is the larger of the two)
  plan->cand_total = std::max(cand_off, (out_cap + 15) & ~15);
  // slots per frame of the result blocks: the sum of the quotas, rounded up to 16 -- a describe workgroup's 16 slots
  // are then whole 64-byte lines of every section (fewer, full-line writes when the record goes to the host mirror)
  plan->out_cap = (out_cap + 15) & ~15;
  return ORBX_OK;
}

// 8-bit bilinear coefficient tables: OpenCV 4.x generic 8UC1 INTER_LINEAR path
// (imgproc/src/resize.cpp: scale = 1/((double)dst/src); fx = (float)((dx+0.5)*
// scale-0.5); sx = floor(fx); clamp with fx=0; 11-bit coefficients by cvRound).
// OpenCV is not part of this image: PARITY UNPINNED (DESIGN.md "Pyramid").
void make_taps(OrbxPlan& plan, std::vector<OrbxResizeTap>* taps) {
  size_t total = 0;
  for (int l = 1; l < plan.nlevels; l++) total += (size_t)align_up(plan.L[l].w, 4) + align_up(plan.L[l].h, 4);
  taps->assign(total ? total : 1, OrbxResizeTap{0, 0, 0});
  for (int l = 1; l < plan.nlevels; l++) {
    const OrbxLevel& L = plan.L[l];
    const double scale_x = 1. / ((double)L.w / plan.w0), scale_y = 1. / ((double)L.h / plan.h0);
    for (int dx = 0; dx < L.w; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)std::floor(fx);
      fx -= (float)sx;
      if (sx < 0) {
        fx = 0;
        sx = 0;
      }
      if (sx >= plan.w0 - 1) {
        fx = 0;
        sx = plan.w0 - 1;
      }
      OrbxResizeTap& t = (*taps)[L.xtab_off + dx];
      t.ofs = sx;
      t.c0 = (int16_t)std::lrintf((1.f - fx) * 2048.f);
      t.c1 = (int16_t)std::lrintf(fx * 2048.f);
      if (sx == plan.w0 - 1) {
        // src[w0-1]*c0 (+ src[w0-1]*0) == src[w0-2]*0 + src[w0-1]*c0: same sum, but the
        // kernel may now always fetch the pair (ofs, ofs+1) with one 16-bit load
        t.ofs = plan.w0 - 2;
        t.c1 = t.c0;
        t.c0 = 0;
      }
    }
    // can k_pyramid2 fetch the pairs of outputs 4g..4g+3 with one 8-byte window?
    {
      bool ok = plan.w0 >= 8;
      for (int dx = 0; dx + 3 < L.w && ok; dx += 4)
        ok = (*taps)[L.xtab_off + dx + 3].ofs + 1 - (*taps)[L.xtab_off + dx].ofs <= 7;
      // (a trailing partial group only adds padding taps with ofs 0, which the kernel never uses)
      for (int dx = L.w & ~3; dx < L.w && ok; dx++)
        ok = (*taps)[L.xtab_off + dx].ofs + 1 - (*taps)[L.xtab_off + (L.w & ~3)].ofs <= 7;
      plan.L[l].win8 = ok ? 1 : 0;
      if (!ok && plan.w0 >= 8) {
        // ... or stage the source rows of a strip through LDS (k_pyrblur): every 256-pixel strip of the level must
        // read its pairs from at most ORBX_PYR_STAGE_BYTES of a source row (scales up to ~3.2; orbx_internal.h).  The span starts at
        // the first pair, moved down so that its dwords END with the source row -- the buffer descriptor of the
        // frame returns zero for a dword that straddles the frame's last byte, which the dword holding the last
        // bytes of the last source row would do at any other alignment; where that is impossible (first strip: the
        // span cannot start before the row) the strip must not reach that dword.
        bool ok3 = true;
        for (int x0 = 0; x0 < L.w && ok3; x0 += ORBX_PYRBLUR_TW) {
          const int ofs_first = (*taps)[L.xtab_off + x0].ofs, ofs_last = (*taps)[L.xtab_off + std::min(x0 + 255, L.w - 1)].ofs;
          const int want = ofs_first - ((ofs_first - plan.w0) & 3), span0 = std::max(want, 0);
          ok3 = ofs_last + 2 - span0 <= ORBX_PYR_STAGE_BYTES;
          if (want < 0 && (plan.w0 & 3) && ofs_last + 2 > (plan.w0 & ~3)) ok3 = false;
        }
        if (ok3) plan.L[l].win8 = 3;
      }
    }
    for (int dy = 0; dy < L.h; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = (int)std::floor(fy);
      fy -= (float)sy;
      OrbxResizeTap& t = (*taps)[L.ytab_off + dy];
      t.ofs = sy;  // rows are clamped in the kernel, weights kept (OpenCV clips the row index only)
      t.c0 = (int16_t)std::lrintf((1.f - fy) * 2048.f);
      t.c1 = (int16_t)std::lrintf(fy * 2048.f);
    }
  }
}

// createGaussianKernel (src/GaussianBlur.cpp:7-37), host side like the reference
int gaussian_kernel(int K, float sigma, float* kernel) {
  if (K <= 0 || (K % 2) == 0 || !kernel) return ORBX_ERR_INVALID_ARG;
  if (sigma <= 0.0f) sigma = 0.3f * ((K - 1) * 0.5f) + 0.8f;
  const int half = K / 2;
  float sum = 0.0f;
  for (int y = -half; y <= half; ++y)
    for (int x = -half; x <= half; ++x) {
      const float value = std::exp(-(float)(x * x + y * y) / (2 * sigma * sigma));
      kernel[(y + half) * K + (x + half)] = value;
      sum += value;
    }
  for (int i = 0; i < K * K; ++i) kernel[i] /= sum;
  return ORBX_OK;
}

int validate_params(const orbx_params& p, std::string* why) {
  auto bad = [&](const char* m) {
    *why = m;
    return (int)ORBX_ERR_INVALID_ARG;
  };
  if (p.nfeatures < 0) return bad("nfeatures < 0");
  if (!(p.scale_factor > 1.0f) || !(p.scale_factor <= 4.0f)) return bad("scale_factor must be in (1, 4]");
  if (p.nlevels < 1 || p.nlevels > ORBX_MAX_LEVELS) return bad("nlevels must be in [1, 16]");
  if (p.n < 1 || p.n > 16) return bad("n must be in [1, 16]");
  if (p.threshold < 0 || p.threshold > 255) return bad("threshold must be in [0, 255]");
  if (p.nms_window < 0 || p.nms_window / 2 > 3) return bad("nms_window must be in [0, 7]");
  if (p.patch_size < 1 || p.patch_size / 2 > 20) return bad("patch_size must be in [1, 41]");
  if (p.harris_window < 1 || (p.harris_window % 2) == 0 || p.harris_window > 15)
    return bad("harris_window must be odd, in [1, 15]");
  if (p.select_mode != ORBX_SELECT_HARRIS && p.select_mode != ORBX_SELECT_ROWMAJOR) return bad("select_mode");
  if (p.blur_levels < 0 || p.blur_levels > 2) return bad("blur_levels");
  if (p.blur_kind < 0 || p.blur_kind > 1) return bad("blur_kind");
  if (p.max_width < 8 || p.max_height < 8 || p.max_width > 16384 || p.max_height > 16384)
    return bad("max_width/max_height must be in [8, 16384]");
  if (p.max_batch < 1 || p.max_batch > 65535) return bad("max_batch must be in [1, 65535]");
  return ORBX_OK;
}

void blur_tiles_for_impl(int impl, const OrbxPlan& plan, std::vector<OrbxTileDesc>* out);

// the working pools of lane k become the context's current ones
void use_lane(orbx_ctx* c, int k) {
  const orbx_ctx::LanePool& L = c->lane_pool[k];
  c->d_pyr = L.d_pyr;
  c->d_pyr_blur = L.d_pyr_blur;
  c->d_mask = L.d_mask;
  c->d_row_stat = L.d_row_stat;
  c->d_cand = L.d_cand;
  c->d_cand_count = L.d_cand_count;
  c->d_cand_total = L.d_cand_total;
  c->d_resp = L.d_resp;
  c->d_lcand = L.d_lcand;
  c->d_lresp = L.d_lresp;
  c->d_lcount = L.d_lcount;
}
// everything either lane has in flight has finished
hipError_t lanes_sync(orbx_ctx* c) {
  for (hipStream_t s : c->lane_stream)
    if (s) {
      const hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

bool fused_pyrblur(const orbx_ctx* c);
bool fast_early_on(const orbx_ctx* c);
int top_rows_env();
bool tile_prefs_apply(const orbx_ctx* c) {
  return c->top_mode == 2 && top_rows_env() > 0 && fast_early_on(c) && fused_pyrblur(c);
}

int set_plan(orbx_ctx* c, int w, int h) {
  if (w == c->plan_w && h == c->plan_h) return ORBX_OK;
  if (w < 8 || h < 8 || w > c->p.max_width || h > c->p.max_height)
    return fail(c, ORBX_ERR_INVALID_ARG, "frame size outside [8, max_width] x [8, max_height]");
  OrbxPlan plan;
  std::string why;
  int st = build_plan(c->p, w, h, &plan, &why, c->fast_impl);
  if (st != ORBX_OK) return fail(c, st, why);
  make_taps(plan, &c->h_taps);
  if (c->h_taps.size() > c->taps_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "resize table exceeds pool");
  // the table may still be in use by an in-flight batch of the previous size
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, lanes_sync(c));
  if (c->last_stream && c->last_stream != c->stream) HIPCHK(c, hipStreamSynchronize(c->last_stream));
  HIPCHK(c, hipMemcpy(c->d_taps, c->h_taps.data(), c->h_taps.size() * sizeof(OrbxResizeTap),
                      hipMemcpyHostToDevice));
  c->plan = plan;
  // the fused pyramid + blur kernel writes only the dwords that hold image pixels; consumers rely on the
  // padding bytes of a level being zero (BRIEF's zero-extension), and another frame size re-uses the pool
  if (c->d_pyr_blur) {  // (on the context's stream, and waited for: batches may run on a caller's stream)
    HIPCHK(c, hipMemsetAsync(c->d_pyr_blur, 0, (size_t)c->p.max_batch * (size_t)c->plan_max.frame_bytes, c->stream));
    const orbx_ctx::LanePool& other = c->lane_pool[c->d_pyr_blur == c->lane_pool[0].d_pyr_blur ? 1 : 0];
    if (other.d_pyr_blur)  // the other lane's pool as well
      HIPCHK(c, hipMemsetAsync(other.d_pyr_blur, 0, (size_t)c->p.max_batch * (size_t)c->plan_max.frame_bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  make_tilemap(plan, ORBX_BLUR_TW, ORBX_BLUR_TH, true, &c->tm_blur);
  // (the adaptive first pass's shorter tile rows only where the top-rows-first pipeline can run at all: with the
  // early exit or the fused kernel switched off every tile works, and the default rows have the smaller halo share)
  c->prefs_applied = tile_prefs_apply(c);
  if ((st = make_bandmap(plan, c->p.nms_window / 2, &c->bm_fast, &why, c->fast_impl == 4,
                         c->prefs_applied ? c->tile_h_pref : nullptr)) != ORBX_OK)
    return fail(c, st, why);
  {
    std::vector<OrbxTileDesc> t;
    blur_tiles_for_impl(c->blur_impl, plan, &t);
    if (t.size() > c->tiles_frame_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "blur tile table exceeds pool");
    HIPCHK(c, hipMemcpy(c->d_tiles_blur, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->blur_tiles_count = (int)t.size();
    build_pyrblur_tiles(plan, ORBX_PYRBLUR_RH, &t, pyr_group_env() > 0);
    if (t.size() > c->tiles_frame_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "strip table exceeds pool");
    HIPCHK(c, hipMemcpy(c->d_tiles_pyrblur, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->pyrblur_tiles_count = (int)t.size();
    build_pyrblur_tiles(plan, ORBX_PYRBLUR_RH, &t, pyr_group_env() > 0, 1, &c->bm_fast, top_rows_env());
    if (t.size() > c->tiles_frame_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "strip table exceeds pool");
    HIPCHK(c, hipMemcpy(c->d_tiles_pyrblur_top, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->pyrblur_top_count = (int)t.size();
    build_pyrblur_tiles(plan, ORBX_PYRBLUR_RH, &t, pyr_group_env() > 0, 2, &c->bm_fast, top_rows_env());
    if (t.size() > c->tiles_frame_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "strip table exceeds pool");
    if (!t.empty())
      HIPCHK(c, hipMemcpy(c->d_tiles_pyrblur_rest, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->pyrblur_rest_count = (int)t.size();
    c->top_levels = OrbxTopLevels{};
    for (int l = 0; l < plan.nlevels; l++)
      if (pyrblur_first_pass_rows(plan, c->bm_fast, l, top_rows_env()) < plan.L[l].h) {
        const int i = c->top_levels.n++;
        c->top_levels.stat_index[i] = l * ORBX_MAX_BANDS;
        c->top_levels.rows[i] = std::min(top_rows_env(), c->bm_fast.tiles_y[l]);
        c->top_levels.cap[i] = plan.L[l].cap;
      }
    build_pyrblur_tiles(plan, ORBX_PYRBLUR_RH_SMALL, &t);
    if (t.size() > c->tiles_small_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "strip table exceeds pool");
    HIPCHK(c, hipMemcpy(c->d_tiles_pyrblur_small, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->pyrblur_small_count = (int)t.size();
    build_frame_tiles(plan, ORBX_PYR2_TW, ORBX_PYR2_TH, true, &t);
    if (t.size() > c->tiles_frame_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "pyramid tile table exceeds pool");
    HIPCHK(c, hipMemcpy(c->d_tiles_pyr2, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
    c->pyr2_tiles_count = (int)t.size();
  }
  {
    std::vector<OrbxTileDesc> t;
    build_fast_tiles(plan, c->bm_fast, 0, 1, &t);
    if (t.size() > c->tiles_fast_capacity) return fail(c, ORBX_ERR_UNSUPPORTED, "FAST tile table exceeds pool");
    c->fast_tiles_count = (int)t.size();
    if (!t.empty())
      HIPCHK(c, hipMemcpy(c->d_tiles_fast, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
  }
  c->plan_serial++;
  c->plan_w = w;
  c->plan_h = h;
  return ORBX_OK;
}

hipError_t launch_pyramid_auto(orbx_ctx* c, hipStream_t s, int n, const uint8_t* d_in, int in_stride,
                               size_t in_frame_stride) {
  return orbx_launch_pyramid2(s, c->d_tiles_pyr2, c->pyr2_tiles_count, c->plan.frame_bytes, c->plan.w0, c->plan.h0, n,
                              d_in, in_stride, in_frame_stride, c->d_taps, c->d_pyr);
}


bool blur_enabled(const orbx_ctx* c) { return c->p.blur_levels != ORBX_BLUR_NONE; }
// ORBX_FUSE=0: separate pyramid and blur kernels (same results; A/B timing and per-kernel profiles)
bool fused_pyrblur(const orbx_ctx* c) {
  static const int env = [] {
    const char* e = getenv("ORBX_FUSE");
    return e ? atoi(e) : 1;
  }();
  return env && c->fuse && c->p.blur_levels == ORBX_BLUR_ALL && c->p.blur_kind == ORBX_BLUR_SEP16;
}
const uint8_t* final_pyr(const orbx_ctx* c);

// FAST + NMS of the batched path.  Tiles that provably cannot contribute to the
// first `cap` row-major survivors exit early (see decode_band in the kernels);
// ORBX_FAST_EARLY=0 disables that (every tile does the full work).
int fast_early_env() {  // ORBX_FAST_EARLY=0: every tile does the full work (results are identical)
  static const int v = [] {
    const char* e = getenv("ORBX_FAST_EARLY");
    return e ? atoi(e) : 1;
  }();
  return v;
}
bool fast_early_on(const orbx_ctx* c) { return fast_early_env() && c->fast_early; }

// FAST + NMS over tiles [first, first + count) of the context's band-major table
hipError_t launch_fast_tiles(orbx_ctx* c, hipStream_t s, int first, int count, int n, OrbxFastParams fp,
                             unsigned long long* stat, int chunk_scale = 1) {
  if (c->fast_impl == 4)
    return orbx_launch_fast4(s, c->d_tiles_fast + first, count, n, final_pyr(c), c->plan.frame_bytes, c->plan.mask_words, fp,
                             c->d_mask, stat);
  return orbx_launch_fast_nms(s, c->d_tiles_fast + first, count, n, final_pyr(c), c->plan.frame_bytes, c->plan.mask_words,
                              fp, c->d_mask, nullptr, stat, chunk_scale);
}

hipError_t launch_fast_whole(orbx_ctx* c, hipStream_t s, int n, OrbxFastParams fp, bool stats_zeroed = false) {
  unsigned long long* stat = fast_early_on(c) ? c->d_row_stat : nullptr;
  if (stat && !stats_zeroed) {
    hipError_t e = hipMemsetAsync(stat, 0, (size_t)n * ORBX_FAST_STAT_WORDS * 8, s);
    if (e != hipSuccess) return e;
  }
  // (a two-launch variant -- tile row 0 first, the rest in strips of several tiles per
  // workgroup -- was measured slower: the kernel boundary costs more than the cheaper exits save)
  return launch_fast_tiles(c, s, 0, c->fast_tiles_count, n, fp, stat);
}

// separable kind -> register-streaming kernel; /273 kind -> LDS tile kernel.
// ORBX_BLUR_IMPL (read when a context is created): 2 (default) k_blur3, 4 pixels per lane; 3: k_blur4, 16 pixels per
// lane (measured 9 % slower: the blur's vertical pass dominates its instruction count, and 94 registers leave 5
// waves per SIMD); 1: the first-generation LDS tile kernel.  The strip table is built for the kernel that reads it.
int blur_impl_env() {
  const char* e = getenv("ORBX_BLUR_IMPL");
  const int v = e ? atoi(e) : 2;
  return v >= 1 && v <= 3 ? v : 2;
}
void blur_tiles_for_impl(int impl, const OrbxPlan& plan, std::vector<OrbxTileDesc>* out) {
  if (impl == 3)
    build_blur4_tiles(plan, out);
  else
    build_blur_tiles(plan, out);
}
hipError_t launch_blur_auto(int impl, hipStream_t s, const OrbxPlan& P, const OrbxTileMap& tm1, const OrbxTileDesc* tiles2,
                            int ntiles2, int n, const uint8_t* src, uint8_t* dst, int first_level, int kind) {
  if (kind == ORBX_BLUR_SEP16 && impl == 3)
    return orbx_launch_blur4(s, tiles2, ntiles2, P.frame_bytes, n, src, dst, first_level);
  if (kind == ORBX_BLUR_SEP16 && impl != 1)
    return orbx_launch_blur3(s, tiles2, ntiles2, P.frame_bytes, n, src, dst, first_level);
  return orbx_launch_blur(s, P, tm1, n, src, dst, first_level, kind);
}
const uint8_t* final_pyr(const orbx_ctx* c) { return blur_enabled(c) ? c->d_pyr_blur : c->d_pyr; }

// Top-rows-first pipeline: wanted for the next batch?  (Eligibility -- fused kernel, early exit on, large
// batch -- is checked where the launches are made.)
bool top_rows_wanted(const orbx_ctx* c) {
  if (top_rows_env() <= 0 || c->top_mode == 0) return false;
  return c->top_mode == 1 || c->top_on;
}
// adaptive mode: look at the totals the second passes have reported so far (no waiting: whatever has arrived)
void top_rows_update(orbx_ctx* c) {
  if (c->top_mode != 2 || !c->h_feedback) return;
  // ONE 64-bit load of the pinned pair (the device writes it as one 8-byte store of two lanes' dwords at best: a torn
  // or stale pair, or a low half that has wrapped, only misleads this heuristic for one batch -- results never
  // depend on it)
  const uint64_t both = *reinterpret_cast<const volatile uint64_t*>(c->h_feedback);
  const uint32_t sk = (uint32_t)both, pr = (uint32_t)(both >> 32);
  const uint32_t dsk = sk - c->feedback_seen[0], dpr = pr - c->feedback_seen[1];
  if (dsk + dpr > 0) {
    c->feedback_seen[0] = sk;
    c->feedback_seen[1] = pr;
    const bool pays = 4ull * dsk >= (unsigned long long)(dsk + dpr);  // a quarter of the levels skipped
    if (!pays && c->top_on) c->top_single_batches = 0;
    c->top_on = pays;
  }
  if (!c->top_on && ++c->top_single_batches >= 128) {  // probe again
    c->top_on = true;
    c->top_single_batches = 0;
  }
}

// Adaptive first pass.  The top-rows-first pipeline produces and searches the first ORBX_TOP_ROWS FAST tile rows of
// every level before anything else; with the default tile rows (<= 47 rows, balanced) that is ~90 rows per level,
// while a level's cap is typically full after 40-75 rows on the benchmark stream -- and everything a first-pass tile
// row holds beyond that row is work nobody reads.  The selection kernel reports, per level, the row in which the
// cap filled (maximum over the frames of a batch; the whole level if it never did); from the maximum over the last
// two observation windows, plus a margin, this picks SHORTER tile rows for the level, so that the first pass ends
// just below that row.  Only the partition of the work changes, never a result; a stream whose caps fill lower
// gets taller tile rows back (at most the default), and a frame whose cap is not full after the first pass is
// finished by the second one as always.  Returns true if the tile tables must be rebuilt (the caller forces
// set_plan, which waits for the batches in flight: it happens a few times per stream, not per batch).
bool adapt_tile_rows(orbx_ctx* c) {
  if (!tile_prefs_apply(c) || !c->h_feedback || c->plan_w <= 0) return false;
  const int top = top_rows_env(), nl = c->plan.nlevels;
  bool any = false;
  for (int l = 0; l < nl; l++) {
    const uint32_t v = c->h_feedback[2 + l];  // (whatever has arrived; a stale or torn word only misleads the heuristic)
    if (v) {
      c->need_cur[l] = std::max(c->need_cur[l], std::min<uint32_t>(v, (uint32_t)c->plan.L[l].h));
      any = true;
    }
  }
  if (!any || ++c->need_batches < c->need_window) return false;
  // end of an observation window (2, 4, 8, 16, then every 32 batches)
  c->need_batches = 0;
  c->need_window = std::min(2 * c->need_window, 32);
  bool change = false, grow = false;
  int want[ORBX_MAX_LEVELS] = {};
  for (int l = 0; l < nl; l++) {
    const uint32_t need = std::max(c->need_cur[l], c->need_prev[l]);
    c->need_prev[l] = c->need_cur[l];
    c->need_cur[l] = 0;
    if (need == 0) {
      want[l] = c->tile_h_pref[l];
      continue;
    }
    const int rows = (int)need + std::max(4, (int)need / 10);  // margin: the next frames' caps may fill a little lower
    // the default tile rows of the level (make_bandmap: <= dflt rows, balanced)
    const int dflt = orbx_fast3_tile_h(c->p.nms_window / 2), lh = c->plan.L[l].h;
    const int bal = (lh + (lh + dflt - 1) / dflt - 1) / ((lh + dflt - 1) / dflt);
    int hh = (rows + top - 1) / top;
    // the streaming FAST kernel walks whole groups of 7 centre rows (tile rows + 2 x NMS radius): a height that is
    // one or two rows into a new group gives them back where at least 3 rows of margin remain (measured: level 0 at
    // 40 instead of 41 rows, FAST -4.6 %, pyramid -2.9 %; rounding every level to a group boundary costs more than
    // it saves)
    if (c->fast_impl == 4) {
      const int over = (hh + 2 * (c->p.nms_window / 2)) % 7;
      if ((over == 1 || over == 2) && top * (hh - over) >= (int)need + 3) hh -= over;
    }
    if (hh >= bal || lh <= top * hh) hh = 0;  // the default rows do
    const int eff = hh ? hh : bal, cur = c->bm_fast.tile_h[l], asked = c->tile_h_pref[l] ? c->tile_h_pref[l] : bal;
    // (a level that neither must grow nor gains three rows keeps its height when another level makes the tables change)
    want[l] = ((eff > cur && (int)need > top * cur) || eff + 2 < std::min(cur, asked)) ? hh : c->tile_h_pref[l];
    // the first pass has become too short for this stream -- caps fill BELOW it (the margin is used up): follow
    if (eff > cur && (int)need > top * cur) grow = true;
    // shrink only for a gain of three rows or more (`cur` may be taller than what was asked for: a level never has
    // more tile rows than the level above)
    else if (eff + 2 < std::min(cur, asked)) change = true;
  }
  if (!grow && !change) return false;
  if (!grow && c->retiles >= 4 && c->need_window < 32) return false;  // (settle first)
  for (int l = 0; l < nl; l++) c->tile_h_pref[l] = want[l];
  c->retiles++;
  return true;
}

// the launches of the whole path for n frames already on the device (the plan is set)
int enqueue_batch(orbx_ctx* c, const uint8_t* d_frames, int n, int row_stride, size_t frame_stride, hipStream_t s) {
  const OrbxPlan& P = c->plan;
  const int tm = c->timing;
  hipEvent_t* evs = c->evr[c->ev_calls % ORBX_EVENT_SETS];
  // event slots: 0 start | 1 pyramid | 2 blur | 3 fast | 4 compact | 5 harris | 6 select | 7 describe
  auto mark = [&](int slot, bool roofline_edge) -> hipError_t {
    if (tm == 1 || (tm == 2 && roofline_edge)) return hipEventRecord(evs[slot], s);
    return hipSuccess;
  };
  // tile-row statistics of the FAST early exit: zeroed up front so that the events around the FAST stage bracket
  // the kernel alone -- except in the top-rows-first pipeline (decided below; the same test here), whose first
  // pyramid pass clears them itself: a memset node less per batch
  const bool small0 = (long long)n * c->pyrblur_tiles_count < 4096;
  const bool two_pass0 = fused_pyrblur(c) && !small0 && fast_early_on(c) && top_rows_wanted(c) && c->pyrblur_rest_count > 0 &&
                         c->bm_fast.nbands > top_rows_env();
  if (!two_pass0) HIPCHK(c, hipMemsetAsync(c->d_row_stat, 0, (size_t)n * ORBX_FAST_STAT_WORDS * 8, s));
  HIPCHK(c, mark(0, false));
  OrbxFastParams fp{c->p.threshold, c->p.n, c->p.nms_window / 2};
  bool two_pass = false;
  if (fused_pyrblur(c)) {
    // blur on every level: pyramid and blur in one pass, the un-blurred pyramid is never materialised
    // (the event slots then read: pyramid = 0, blur = the fused kernel)
    HIPCHK(c, mark(1, true));
    // (a wave per strip: below ~4096 waves the chip is far from full and the short-band table wins)
    const bool small = (long long)n * c->pyrblur_tiles_count < 4096;
    // Top rows first.  The FAST early exit rests on the row-major cap (src/orb_cpu.cpp:108-110, src/orb.cpp:63):
    // once the top tile rows of a level hold `cap` survivors, nothing below them is ever looked at -- not by
    // FAST (its tiles exit), not by the selection (it stops at the first cap survivors), not by Harris or
    // the descriptors (their keypoints lie in those top rows).  So the pyramid is produced in two passes:
    //   1. the rows the first ORBX_TOP_ROWS FAST tile rows (and the descriptors of their keypoints) can read,
    //   2. FAST on those tile rows,
    //   3. the remaining rows -- a strip whose (frame, level) already has its cap survivors is skipped,
    //   4. FAST on the remaining tile rows (their tiles exit the same way).
    // What a skipped strip leaves in the pool (rows of an earlier batch) is never read.  Results are
    // identical either way (tests/test_gpu_parity.py, tests/test_batch64_parity.py).
    two_pass = !small && fast_early_on(c) && top_rows_wanted(c) && c->pyrblur_rest_count > 0 &&
               c->bm_fast.nbands > top_rows_env();
    if (!two_pass) {
      HIPCHK(c, orbx_launch_pyrblur(s, small ? c->d_tiles_pyrblur_small : c->d_tiles_pyrblur,
                                    small ? c->pyrblur_small_count : c->pyrblur_tiles_count, P.frame_bytes, P.w0, P.h0, n,
                                    d_frames, row_stride, frame_stride, c->d_taps, c->d_pyr_blur,
                                    small ? 0 : pyr_group_env()));
    } else {
      const int first_tiles = c->bm_fast.band_begin[top_rows_env()];
      HIPCHK(c, orbx_launch_pyrblur(s, c->d_tiles_pyrblur_top, c->pyrblur_top_count, P.frame_bytes, P.w0, P.h0, n, d_frames,
                                    row_stride, frame_stride, c->d_taps, c->d_pyr_blur, pyr_group_env(), nullptr, c->d_feedback, nullptr,
                                    c->d_row_stat));
      HIPCHK(c, mark(ORBX_NUM_STAGE_TIMES + 1, true));
      HIPCHK(c, launch_fast_tiles(c, s, 0, first_tiles, n, fp, c->d_row_stat));
      HIPCHK(c, mark(ORBX_NUM_STAGE_TIMES + 2, true));
      HIPCHK(c, orbx_launch_pyrblur(s, c->d_tiles_pyrblur_rest, c->pyrblur_rest_count, P.frame_bytes, P.w0, P.h0, n,
                                    d_frames, row_stride, frame_stride, c->d_taps, c->d_pyr_blur, pyr_group_env(),
                                    c->d_row_stat, c->d_feedback, &c->top_levels));
      HIPCHK(c, mark(2, true));
      HIPCHK(c, launch_fast_tiles(c, s, first_tiles, c->fast_tiles_count - first_tiles, n, fp, c->d_row_stat, 4));
    }
  } else {
    HIPCHK(c, launch_pyramid_auto(c, s, n, d_frames, row_stride, frame_stride));
    HIPCHK(c, mark(1, true));
    if (blur_enabled(c))
      HIPCHK(c, launch_blur_auto(c->blur_impl, s, P, c->tm_blur, c->d_tiles_blur, c->blur_tiles_count, n, c->d_pyr,
                                 c->d_pyr_blur, c->p.blur_levels == ORBX_BLUR_UPPER ? 1 : 0, c->p.blur_kind));
  }
  if (!two_pass) {
    HIPCHK(c, mark(2, true));
    HIPCHK(c, launch_fast_whole(c, s, n, fp, true));
  }
  c->ev_split[c->ev_calls % ORBX_EVENT_SETS] = two_pass;
  c->last_two_pass = two_pass;
  HIPCHK(c, mark(3, true));
  HIPCHK(c, mark(4, false));  // (compaction, Harris and selection are one kernel: its time is the "select" slot)
  HIPCHK(c, mark(5, false));
  // result block sections are laid out for (n, pool slot capacity)
  c->out_layout = make_out_layout(n, P.out_cap > 0 ? P.out_cap : 1);
  const OutLayout& o = c->out_layout;
  static const int spread = [] {  // ORBX_SELECT_SPREAD=0/1 forces the fused / the three-kernel selection (A/B timing)
    const char* e = getenv("ORBX_SELECT_SPREAD");
    return e ? atoi(e) : -1;
  }();
  HIPCHK(c, orbx_launch_level_select_auto(s, P, n, c->p.select_mode, spread, c->d_mask, final_pyr(c), c->d_gauss,
                                          c->p.harris_window, c->p.harris_k, c->d_lcand, c->d_lcount, c->d_lresp,
                                          c->d_cand, c->d_resp, c->d_cand_count, two_pass ? c->d_feedback + 2 : nullptr));
  HIPCHK(c, mark(6, false));
  if (P.out_cap <= 0)  // nfeatures too small for any quota: no describe launch, so the counts are zeroed here
    HIPCHK(c, hipMemsetAsync(c->d_out + o.counts, 0, sizeof(int32_t) * (size_t)n, s));
  // orbx_set_host_results: the kernel also writes the compact record into the pinned mirror of the block
  OrbxHostRecord hr{};
  if (c->host_results && P.out_cap > 0) {
    uint8_t* hd = nullptr;
    for (int i = 0; i < orbx_ctx::kBlocks; i++)
      if (c->h_out == c->h_outb[i]) hd = c->h_outb_dev[i];
    if (hd) hr = OrbxHostRecord{(int32_t*)(hd + o.counts), (uint32_t*)(hd + o.kp16), (float*)(hd + o.angle), (orbx_descriptor*)(hd + o.desc)};
  }
  HIPCHK(c, orbx_launch_describe(s, P, n, final_pyr(c), c->p.patch_size, c->d_cand_count, c->d_cand, c->d_resp,
                                 (int32_t*)(c->d_out + o.counts), (orbx_keypoint*)(c->d_out + o.lkp),
                                 (float*)(c->d_out + o.resp), (int32_t*)(c->d_out + o.level),
                                 (orbx_keypoint*)(c->d_out + o.kp), (uint32_t*)(c->d_out + o.kp16),
                                 (float*)(c->d_out + o.angle),
                                 (orbx_descriptor*)(c->d_out + o.desc), two_pass ? c->d_feedback : nullptr,
                                 two_pass ? const_cast<uint32_t*>(c->h_feedback) : nullptr, &hr));
  HIPCHK(c, mark(7, false));
  return ORBX_OK;
}

// The launch sequence of a batch depends only on (input pointer and strides, n, plan, switches): it
// is captured once into a hipGraph and replayed with one hipGraphLaunch per batch (7 enqueues ->
// 1; matters most for the one-frame-per-call shape, which is launch-bound).  Stage timing needs
// event records between the kernels, so it takes the plain path.  ORBX_GRAPH=0 disables.
void drop_graph(orbx_ctx* c, int i) {
  if (c->g_exec[i]) (void)hipGraphExecDestroy(c->g_exec[i]);
  c->g_exec[i] = nullptr;
}

int run_batch(orbx_ctx* c, const uint8_t* d_frames, int n, int w, int h, int row_stride, size_t frame_stride,
              hipStream_t s, bool may_pipeline = false) {
  // Pipelined mode: a device-resident batch on the context's stream goes to the lane of its result block -- own
  // pools, own stream, so nothing of the other lane's batch in flight is touched.  (The plan's tables are shared:
  // set_plan waits for both lanes before it changes them.)
  // adaptive first pass (adapt_tile_rows): what was learned belongs to one frame size; new tile-row heights make
  // set_plan below rebuild the tables (it waits for the batches in flight; this batch then runs unpipelined)
  if (w != c->learn_w || h != c->learn_h) {
    c->learn_w = w;
    c->learn_h = h;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) c->need_cur[l] = c->need_prev[l] = 0, c->tile_h_pref[l] = 0;
    for (int i = 2; i < ORBX_FEEDBACK_WORDS; i++) c->h_feedback[i] = 0;
    c->need_batches = 0;
    c->need_window = 2;
    c->retiles = 0;
  } else if (w == c->plan_w && h == c->plan_h) {
    bool any_pref = false;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) any_pref |= c->tile_h_pref[l] != 0;
    // (a switch that takes the top-rows-first pipeline away, or brings it back, since the tables were built)
    if (adapt_tile_rows(c) || (any_pref && tile_prefs_apply(c) != c->prefs_applied)) c->plan_w = 0;
  }
  const bool lanes = may_pipeline && c->pipelined && s == c->stream && w == c->plan_w && h == c->plan_h;
  const int lane = lanes ? c->next_lane : 0;
  if (lanes) {
    use_lane(c, lane);
    s = c->lane_stream[lane];
  } else {
    if (c->lane_stream[1]) HIPCHK(c, lanes_sync(c));
    use_lane(c, 0);
    // the pools (pyramids, mask, result block) are reused by every batch: a batch still in
    // flight on a DIFFERENT stream must have finished before this one may touch them
    if (c->last_stream && c->last_stream != s) HIPCHK(c, hipStreamSynchronize(c->last_stream));
  }
  int st = set_plan(c, w, h);
  if (st != ORBX_OK) return st;
  top_rows_update(c);
  // this batch writes the other result block; if that block is still the source of an
  // asynchronous D2H copy (orbx_batch_prefetch two batches ago), the kernels wait for the copy
  const int blk = (c->blk + 1) % orbx_ctx::kBlocks;
  if (c->copy_pending[blk]) HIPCHK(c, hipStreamWaitEvent(s, c->ev_copied[blk], 0));
  // the previous users of this lane's pools and of this result block, if they ran on another stream (a batch on a
  // caller's stream between pipelined batches, or the other way round): device-side waits, no host stall
  if (c->pool_stream[lane] && c->pool_stream[lane] != s) HIPCHK(c, hipStreamWaitEvent(s, c->ev_pool[lane], 0));
  if (c->blk_stream[blk] && c->blk_stream[blk] != s) HIPCHK(c, hipStreamWaitEvent(s, c->ev_done[blk], 0));
  // The block becomes "the last batch" only once its launches are enqueued: after a failed call orbx_batch_fetch
  // must not hand out what an older batch left in it.
  c->nb[blk] = 0;
  c->copy_pending[blk] = false;
  c->copy_compact[blk] = false;
  uint8_t* const prev_d_out = c->d_out;
  uint8_t* const prev_h_out = c->h_out;
  const OutLayout prev_layout = c->out_layout;
  c->d_out = c->d_outb[blk];  // (enqueue_batch writes through c->d_out)
  c->h_out = c->h_outb[blk];
  auto fail_restore = [&](int status) {
    c->d_out = prev_d_out;
    c->h_out = prev_h_out;
    c->out_layout = prev_layout;
    return status;
  };
  static const int use_graph = [] {
    const char* e = getenv("ORBX_GRAPH");
    return e ? atoi(e) : 1;
  }();
  const int tm = c->timing;
  if (use_graph && tm == 0) {
    const OrbxGraphKey key{d_frames, frame_stride, n, w, h, row_stride, (fast_early_on(c) ? 1 : 0) | (fused_pyrblur(c) ? 2 : 0) | (top_rows_wanted(c) ? 4 : 0) | (lanes ? 8 : 0) | (lane << 4) | (c->host_results ? 64 : 0),
                           c->plan_serial, blk};
    int gi = -1;
    for (int i = 0; i < orbx_ctx::kGraphs; i++)
      if (c->g_exec[i] && key == c->g_key[i]) gi = i;
    if (gi < 0) {
      gi = c->g_next;
      c->g_next = (c->g_next + 1) % orbx_ctx::kGraphs;
      drop_graph(c, gi);
      hipGraph_t g = nullptr;
      const hipError_t be = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (be != hipSuccess) return fail_restore(fail(c, ORBX_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(be)));
      st = enqueue_batch(c, d_frames, n, row_stride, frame_stride, s);
      const hipError_t ee = hipStreamEndCapture(s, &g);
      if (st != ORBX_OK) {
        if (g) (void)hipGraphDestroy(g);
        return fail_restore(st);
      }
      if (ee != hipSuccess) return fail_restore(fail(c, ORBX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee)));
      const hipError_t ie = hipGraphInstantiate(&c->g_exec[gi], g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ie != hipSuccess) {
        c->g_exec[gi] = nullptr;
        return fail_restore(fail(c, ORBX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)));
      }
      c->g_key[gi] = key;
    }
    const hipError_t le = hipGraphLaunch(c->g_exec[gi], s);
    if (le != hipSuccess) return fail_restore(fail(c, ORBX_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(le)));
  } else {
    if ((st = enqueue_batch(c, d_frames, n, row_stride, frame_stride, s)) != ORBX_OK) return fail_restore(st);
  }
  c->blk = blk;
  if (lanes) c->next_lane ^= 1;
  c->out_layout = make_out_layout(n, c->plan.out_cap > 0 ? c->plan.out_cap : 1);
  c->last_n = n;
  c->last_stream = s;
  c->layoutb[blk] = c->out_layout;
  c->nb[blk] = n;
  c->capb[blk] = c->plan.out_cap > 0 ? c->plan.out_cap : 1;
  c->copy_pending[blk] = false;
  c->host_written[blk] = c->host_results && c->plan.out_cap > 0;
  HIPCHK(c, hipEventRecord(c->ev_done[blk], s));
  if (c->host_written[blk]) {
    // orbx_set_host_results: the compact record is in the pinned mirror when the batch ends -- the block counts as
    // compact-copied from the start (the copy stream is not involved; orbx_batch_prefetch_compact has nothing to do)
    HIPCHK(c, hipEventRecord(c->ev_copied[blk], s));
    c->copy_pending[blk] = true;
    c->copy_compact[blk] = true;
  }
  HIPCHK(c, hipEventRecord(c->ev_pool[lane], s));
  c->pool_stream[lane] = s;
  c->blk_stream[blk] = s;
  if (tm != 0) {
    c->ev_mode[c->ev_calls % ORBX_EVENT_SETS] = tm;
    c->ev_calls++;
  }
  return ORBX_OK;
}

int check_image(orbx_ctx* c, const void* img, int w, int h, int stride) {
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!img) return fail(c, ORBX_ERR_INVALID_ARG, "image is NULL");
  if (w < 8 || h < 8 || w > c->p.max_width || h > c->p.max_height)
    return fail(c, ORBX_ERR_INVALID_ARG, "image size outside [8, max_width] x [8, max_height]");
  if (stride < w) return fail(c, ORBX_ERR_INVALID_ARG, "stride < width");
  return ORBX_OK;
}

// single-level plan over a scratch image, for the stage-level operators
OrbxPlan flat_plan(int w, int h, int cap) {
  OrbxPlan P;
  std::memset(&P, 0, sizeof(P));
  P.nlevels = 1;
  P.w0 = w;
  P.h0 = h;
  OrbxLevel& L = P.L[0];
  L.w = w;
  L.h = h;
  L.pitch = align_up(w, 64);
  L.mask_wpr = (w + 63) / 64;
  L.cap = cap;
  L.quota = cap;
  L.scale = 1.0f;
  P.frame_bytes = (int32_t)align_up_sz((size_t)L.pitch * h, 256);
  P.mask_words = L.mask_wpr * h;
  P.cand_total = cap;
  P.out_cap = cap;
  return P;
}

// upload a host image into a zero-padded, 64-aligned-pitch scratch image
int upload_flat(orbx_ctx* c, DevBuf& b, const uint8_t* img, int w, int h, int stride, int* pitch) {
  const int p = align_up(w, 64);
  int st = ensure(c, b, (size_t)p * h + 256);
  if (st != ORBX_OK) return st;
  HIPCHK(c, hipMemsetAsync(b.p, 0, (size_t)p * h, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(b.p, p, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  *pitch = p;
  return ORBX_OK;
}

}  // namespace

// ===========================================================================
extern "C" {

int orbx_params_default_gpu(orbx_params* p) {
  if (!p) return ORBX_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->nfeatures = 500;  // include/orb.hpp:36
  p->scale_factor = 1.2f;
  p->nlevels = 8;
  p->threshold = 20;  // include/orb.hpp:12
  p->n = 9;
  p->nms_window = 3;
  p->patch_size = 31;
  p->harris_window = 7;  // src/orb.cpp:65
  p->harris_k = 0.04f;
  p->select_mode = ORBX_SELECT_HARRIS;
  p->blur_levels = ORBX_BLUR_NONE;
  p->blur_kind = ORBX_BLUR_SEP16;
  p->max_width = 1920;
  p->max_height = 1080;
  p->max_batch = 1;
  p->device = -1;
  return ORBX_OK;
}

int orbx_params_default_cpu(orbx_params* p) {
  int st = orbx_params_default_gpu(p);
  if (st != ORBX_OK) return st;
  p->nfeatures = 3000;  // include/orb_cpu.hpp:6
  p->threshold = 50;
  p->n = 9;
  p->nms_window = 3;
  p->patch_size = 9;
  p->nlevels = 1;  // ORBCPU::detectAndCompute ignores the pyramid (src/orb_cpu.cpp:271-276)
  p->select_mode = ORBX_SELECT_ROWMAJOR;
  return ORBX_OK;
}

const char* orbx_status_string(int status) {
  switch (status) {
    case ORBX_OK:
      return "ok";
    case ORBX_ERR_INVALID_ARG:
      return "invalid argument";
    case ORBX_ERR_CAPACITY:
      return "output capacity exceeded";
    case ORBX_ERR_HIP:
      return "HIP runtime error";
    case ORBX_ERR_NO_DEVICE:
      return "no usable gfx950 device";
    case ORBX_ERR_UNSUPPORTED:
      return "unsupported parameter combination";
    default:
      return "unknown status";
  }
}

const char* orbx_version(void) { return "liborbx 0.1.0 gfx950"; }

const char* orbx_last_error_string(const orbx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

namespace {
// the second lane's pools and stream (nothing of it may be in flight)
void free_lane1(orbx_ctx* c) {
  orbx_ctx::LanePool& L = c->lane_pool[1];
  void* lb[] = {L.d_pyr, L.d_pyr_blur, L.d_mask, L.d_row_stat, L.d_cand, L.d_cand_count, L.d_cand_total, L.d_resp,
                L.d_lcand, L.d_lresp, L.d_lcount};
  for (void* b : lb)
    if (b) (void)hipFree(b);
  L = orbx_ctx::LanePool{};
  if (c->lane_stream[1]) (void)hipStreamDestroy(c->lane_stream[1]);
  c->lane_stream[1] = nullptr;
  c->pool_stream[1] = nullptr;
}
}  // namespace

void orbx_destroy(orbx_ctx* c) {
  DeviceGuard _dg(c);
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->lane_stream[1]) (void)hipStreamSynchronize(c->lane_stream[1]);
  use_lane(c, 0);  // (the list below frees the context's own pools)
  free_lane1(c);   // the second lane of the pipelined mode (also what a failed enable left behind)
  for (int i = 0; i < orbx_ctx::kGraphs; i++) drop_graph(c, i);
  void* bufs[] = {c->d_in,   c->d_pyr,  c->d_pyr_blur, c->d_mask, c->d_cand, c->d_cand_count, c->d_cand_total,
                  c->d_resp, c->d_taps, c->d_gauss,    c->d_row_stat, c->d_tiles_fast, c->d_tiles_blur, c->d_tiles_pyr2, c->d_tiles_pyrblur, c->d_tiles_pyrblur_small,
                  c->d_tiles_pyrblur_top, c->d_tiles_pyrblur_rest, c->d_feedback, c->d_lcand, c->d_lresp, c->d_lcount};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->h_feedback) (void)hipHostFree(const_cast<uint32_t*>(c->h_feedback));
  for (int i = 0; i < orbx_ctx::kBlocks; i++) {
    if (c->h_outb[i]) (void)hipHostFree(c->h_outb[i]);
    if (c->d_outb[i]) (void)hipFree(c->d_outb[i]);
    if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
  }
  for (auto& e : c->ev_pool)
    if (e) (void)hipEventDestroy(e);
  if (c->cstream) {
    (void)hipStreamSynchronize(c->cstream);
    (void)hipStreamDestroy(c->cstream);
  }
  DevBuf* sb[] = {&c->s_img_a, &c->s_img_b, &c->s_f32,  &c->s_u16, &c->s_mask, &c->s_kps,   &c->s_f32b, &c->s_desc,
                  &c->s_i32,   &c->s_kern,  &c->s_tiles, &c->m_q,    &c->m_t,   &c->m_idx,  &c->m_dist,  &c->m_match, &c->m_cnt,
                  &c->lk_img[0], &c->lk_img[1], &c->lk_deriv, &c->lk_io};
  if (c->lk_host) (void)hipHostFree(c->lk_host);
  for (DevBuf* b : sb)
    if (b->p) (void)hipFree(b->p);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& set : c->evr)
    for (auto& e : set)
      if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int orbx_create(const orbx_params* p, orbx_ctx** out) {
  if (!p || !out) return fail(nullptr, ORBX_ERR_INVALID_ARG, "params/out is NULL");
  *out = nullptr;
  std::string why;
  int st = validate_params(*p, &why);
  if (st != ORBX_OK) return fail(nullptr, st, why);

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, ORBX_ERR_NO_DEVICE, "hipGetDeviceCount found no device (liborbx has no CPU fallback)");
  int dev = p->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  }
  if (dev >= ndev) return fail(nullptr, ORBX_ERR_INVALID_ARG, "device ordinal out of range");
  DeviceGuard dg(dev);  // the caller's current device is restored on every return path
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != dev) return fail(nullptr, ORBX_ERR_NO_DEVICE, "hipSetDevice failed");
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return fail(nullptr, ORBX_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, ORBX_ERR_NO_DEVICE,
                std::string("device is ") + prop.gcnArchName + ", liborbx is built for gfx950 only");

  orbx_ctx* c = new (std::nothrow) orbx_ctx();
  if (!c) return fail(nullptr, ORBX_ERR_HIP, "out of host memory");
  c->p = *p;
  c->device = dev;

  c->fast_impl = fast_impl_env();
  c->blur_impl = blur_impl_env();
  st = build_plan(c->p, p->max_width, p->max_height, &c->plan_max, &why, c->fast_impl);
  if (st != ORBX_OK) {
    delete c;
    return fail(nullptr, st, why);
  }
  const OrbxPlan& M = c->plan_max;
  const size_t B = (size_t)p->max_batch;

#define CREATE_CHK(expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
      orbx_destroy(c);                                                                       \
      return fail(nullptr, ORBX_ERR_HIP, m);                                                 \
    }                                                                                        \
  } while (0)

  CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto& e : c->ev) CREATE_CHK(hipEventCreate(&e));
  for (auto& set : c->evr)
    for (auto& e : set) CREATE_CHK(hipEventCreate(&e));
  CREATE_CHK(hipMalloc((void**)&c->d_in, B * (size_t)p->max_width * p->max_height + 256));
  CREATE_CHK(hipMalloc((void**)&c->d_pyr, B * (size_t)M.frame_bytes + 256));
  if (p->blur_levels != ORBX_BLUR_NONE)
    CREATE_CHK(hipMalloc((void**)&c->d_pyr_blur, B * (size_t)M.frame_bytes + 256));
  CREATE_CHK(hipMalloc((void**)&c->d_mask, B * (size_t)M.mask_words * 8 + 256));
  CREATE_CHK(hipMalloc((void**)&c->d_row_stat, B * ORBX_FAST_STAT_WORDS * 8));
  CREATE_CHK(hipMalloc((void**)&c->d_feedback, ORBX_FEEDBACK_WORDS * 4));
  CREATE_CHK(hipMemset(c->d_feedback, 0, ORBX_FEEDBACK_WORDS * 4));
  CREATE_CHK(hipHostMalloc((void**)&c->h_feedback, ORBX_FEEDBACK_WORDS * 4, hipHostMallocDefault));
  for (int i = 0; i < ORBX_FEEDBACK_WORDS; i++) c->h_feedback[i] = 0;
  {
    OrbxBandMap bmm;  // (sized for the shortest tile rows the adaptive first pass may choose)
    int min_pref[ORBX_MAX_LEVELS];
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) min_pref[l] = ORBX_MIN_TILE_H;
    if ((st = make_bandmap(M, p->nms_window / 2, &bmm, &why, c->fast_impl == 4, min_pref)) != ORBX_OK) {
      orbx_destroy(c);
      return fail(nullptr, st, why);
    }
    c->tiles_fast_capacity = (size_t)bmm.band_begin[bmm.nbands];
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_fast, std::max<size_t>(c->tiles_fast_capacity, 1) * sizeof(OrbxTileDesc)));
    std::vector<OrbxTileDesc> t1, t2;
    blur_tiles_for_impl(c->blur_impl, M, &t1);
    build_frame_tiles(M, ORBX_PYR2_TW, ORBX_PYR2_TH, true, &t2);
    std::vector<OrbxTileDesc> t3, t4;
    build_pyrblur_tiles(M, ORBX_PYRBLUR_RH, &t3);
    build_pyrblur_tiles(M, ORBX_PYRBLUR_RH_SMALL, &t4);
    c->tiles_small_capacity = t4.size() + 64;
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_pyrblur_small, c->tiles_small_capacity * sizeof(OrbxTileDesc)));
    c->tiles_frame_capacity = std::max(std::max(t1.size(), t2.size()), t3.size()) + 256;  // (+ the extra bands of a split table)
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_pyrblur, c->tiles_frame_capacity * sizeof(OrbxTileDesc)));
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_pyrblur_top, c->tiles_frame_capacity * sizeof(OrbxTileDesc)));
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_pyrblur_rest, c->tiles_frame_capacity * sizeof(OrbxTileDesc)));
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_blur, c->tiles_frame_capacity * sizeof(OrbxTileDesc)));
    CREATE_CHK(hipMalloc((void**)&c->d_tiles_pyr2, c->tiles_frame_capacity * sizeof(OrbxTileDesc)));
  }
  CREATE_CHK(hipMalloc((void**)&c->d_cand, B * (size_t)std::max(M.cand_total, 1) * sizeof(orbx_keypoint)));
  CREATE_CHK(hipMalloc((void**)&c->d_cand_count, B * ORBX_MAX_LEVELS * sizeof(int32_t)));
  CREATE_CHK(hipMalloc((void**)&c->d_cand_total, B * ORBX_MAX_LEVELS * sizeof(int32_t)));
  CREATE_CHK(hipMalloc((void**)&c->d_resp, B * (size_t)std::max(M.cand_total, 1) * sizeof(float)));
  CREATE_CHK(hipMalloc((void**)&c->d_lcand, B * (size_t)std::max(M.cand_total, 1) * sizeof(uint32_t)));
  CREATE_CHK(hipMalloc((void**)&c->d_lresp, B * (size_t)std::max(M.cand_total, 1) * sizeof(float)));
  CREATE_CHK(hipMalloc((void**)&c->d_lcount, B * ORBX_MAX_LEVELS * sizeof(int32_t)));
  {
    size_t taps = 1;
    for (int l = 1; l < M.nlevels; l++) taps += (size_t)align_up(M.L[l].w, 4) + align_up(M.L[l].h, 4);
    // level sizes of smaller frames never exceed those of the largest frame
    c->taps_capacity = taps + 16;
    CREATE_CHK(hipMalloc((void**)&c->d_taps, c->taps_capacity * sizeof(OrbxResizeTap)));
  }
  {
    const int K = p->harris_window;
    std::vector<float> g((size_t)K * K);
    gaussian_kernel(K, -1.0f, g.data());
    CREATE_CHK(hipMalloc((void**)&c->d_gauss, g.size() * sizeof(float)));
    CREATE_CHK(hipMemcpy(c->d_gauss, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  c->out_cap = std::max(M.out_cap, 1);
  {
    const OutLayout o = make_out_layout((int)B, c->out_cap);
    {
      // The copy stream gets the HIGHEST priority -- not for the copies' sake: streams of one priority share a few
      // hardware queues, and the copy of a batch is enqueued behind a wait for the batch's end.  In a queue shared
      // with the other lane's stream that wait holds back the other lane's kernels for as long as the batch runs
      // (measured: the rate with the results on the host then falls from 0.99 to 0.8 of the rate without copies,
      // depending on which streams the process happens to have created); another priority is another queue.
      int least = 0, greatest = 0;
      CREATE_CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      CREATE_CHK(hipStreamCreateWithPriority(&c->cstream, hipStreamNonBlocking, greatest));
    }
    for (int i = 0; i < orbx_ctx::kBlocks; i++) {
      CREATE_CHK(hipMalloc((void**)&c->d_outb[i], o.total));
      CREATE_CHK(hipHostMalloc((void**)&c->h_outb[i], o.total, hipHostMallocDefault));
      CREATE_CHK(hipHostGetDevicePointer((void**)&c->h_outb_dev[i], c->h_outb[i], 0));
      CREATE_CHK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
      CREATE_CHK(hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
    }
    for (auto& e : c->ev_pool) CREATE_CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->d_out = c->d_outb[0];
    c->h_out = c->h_outb[0];
  }
#undef CREATE_CHK
  {  // lane 0 of the pipelined mode = the context's own pools and stream
    orbx_ctx::LanePool& L = c->lane_pool[0];
    L.d_pyr = c->d_pyr;
    L.d_pyr_blur = c->d_pyr_blur;
    L.d_mask = c->d_mask;
    L.d_row_stat = c->d_row_stat;
    L.d_cand = c->d_cand;
    L.d_cand_count = c->d_cand_count;
    L.d_cand_total = c->d_cand_total;
    L.d_resp = c->d_resp;
    L.d_lcand = c->d_lcand;
    L.d_lresp = c->d_lresp;
    L.d_lcount = c->d_lcount;
    c->lane_stream[0] = c->stream;
  }
  *out = c;
  return ORBX_OK;
}

int orbx_get_plan(orbx_ctx* c, int width, int height, int32_t* level_w, int32_t* level_h, int32_t* quota,
                  int32_t* fast_cap, float* level_scale_out, int32_t* out_capacity) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  OrbxPlan plan;
  std::string why;
  int st = build_plan(c->p, width, height, &plan, &why, c->fast_impl);
  if (st != ORBX_OK) return fail(c, st, why);
  for (int l = 0; l < plan.nlevels; l++) {
    if (level_w) level_w[l] = plan.L[l].w;
    if (level_h) level_h[l] = plan.L[l].h;
    if (quota) quota[l] = plan.L[l].quota;
    if (fast_cap) fast_cap[l] = plan.L[l].cap;
    if (level_scale_out) level_scale_out[l] = plan.L[l].scale;
  }
  if (out_capacity) *out_capacity = plan.out_cap;
  return ORBX_OK;
}

int orbx_detect_and_compute_batch_device(orbx_ctx* c, const void* d_frames, int n, int width, int height,
                                         int row_stride, size_t frame_stride, void* stream) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!d_frames) return fail(c, ORBX_ERR_INVALID_ARG, "d_frames is NULL");
  if (n < 1 || n > c->p.max_batch) return fail(c, ORBX_ERR_INVALID_ARG, "n outside [1, max_batch]");
  if (row_stride < width) return fail(c, ORBX_ERR_INVALID_ARG, "row_stride < width");
  if (frame_stride < (size_t)row_stride * (size_t)(height - 1) + (size_t)width)
    return fail(c, ORBX_ERR_INVALID_ARG, "frame_stride smaller than a frame");
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  return run_batch(c, (const uint8_t*)d_frames, n, width, height, row_stride, frame_stride, s, true);
}

int orbx_detect_and_compute_batch_host(orbx_ctx* c, const uint8_t* frames, int n, int width, int height,
                                       int row_stride, size_t frame_stride) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (n < 1 || n > c->p.max_batch) return fail(c, ORBX_ERR_INVALID_ARG, "n outside [1, max_batch]");
  int st = check_image(c, frames, width, height, row_stride);
  if (st != ORBX_OK) return st;
  if (n > 1 && frame_stride < (size_t)row_stride * (size_t)(height - 1) + (size_t)width)
    return fail(c, ORBX_ERR_INVALID_ARG, "frame_stride smaller than a frame");
  const size_t tight = (size_t)width * height;
  if (row_stride == width && (n == 1 || frame_stride == tight)) {
    HIPCHK(c, hipMemcpyAsync(c->d_in, frames, tight * n, hipMemcpyHostToDevice, c->stream));
  } else {
    for (int i = 0; i < n; i++)
      HIPCHK(c, hipMemcpy2DAsync(c->d_in + tight * i, width, frames + frame_stride * i, row_stride, width, height,
                                 hipMemcpyHostToDevice, c->stream));
  }
  return run_batch(c, c->d_in, n, width, height, width, tight, c->stream);
}

int orbx_wait(orbx_ctx* c) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  HIPCHK(c, hipStreamSynchronize(c->last_stream ? c->last_stream : c->stream));
  HIPCHK(c, lanes_sync(c));
  return ORBX_OK;
}

int orbx_set_host_results(orbx_ctx* c, int enable) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  c->host_results = enable ? 1 : 0;  // (takes effect with the next batched call: part of the launch sequence's key)
  return ORBX_OK;
}

int orbx_set_pipelined_batches(orbx_ctx* c, int enable) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (c->last_stream) HIPCHK(c, hipStreamSynchronize(c->last_stream));
  HIPCHK(c, lanes_sync(c));
  if (enable && !c->lane_stream[1]) {  // the second lane: a stream and a second set of working pools
    const orbx_params* p = &c->p;
    const OrbxPlan& M = c->plan_max;
    const size_t B = (size_t)p->max_batch;
    orbx_ctx::LanePool& L = c->lane_pool[1];
    const size_t nc = (size_t)std::max(M.cand_total, 1);
    hipError_t e = hipMalloc((void**)&L.d_pyr, B * (size_t)M.frame_bytes + 256);
    if (e == hipSuccess && p->blur_levels != ORBX_BLUR_NONE) {
      e = hipMalloc((void**)&L.d_pyr_blur, B * (size_t)M.frame_bytes + 256);
      if (e == hipSuccess) e = hipMemset(L.d_pyr_blur, 0, B * (size_t)M.frame_bytes);  // (the padding bytes of a level stay zero)
    }
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_mask, B * (size_t)M.mask_words * 8 + 256);
    if (e == hipSuccess) e = hipMemset(L.d_mask, 0, B * (size_t)M.mask_words * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_row_stat, B * ORBX_FAST_STAT_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_cand, B * nc * sizeof(orbx_keypoint));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_cand_count, B * ORBX_MAX_LEVELS * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_cand_total, B * ORBX_MAX_LEVELS * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_resp, B * nc * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_lcand, B * nc * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_lresp, B * nc * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&L.d_lcount, B * ORBX_MAX_LEVELS * sizeof(int32_t));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->lane_stream[1], hipStreamNonBlocking);
    if (e != hipSuccess) {  // (at batch 512 the second pool set is GBs: running out of memory is the realistic failure)
      free_lane1(c);
      c->pipelined = false;
      return fail(c, ORBX_ERR_HIP, std::string("second lane of the pipelined mode: ") + hipGetErrorString(e));
    }
  }
  c->pipelined = enable != 0;
  return ORBX_OK;
}

int orbx_set_fast_early_exit(orbx_ctx* c, int enable) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  c->fast_early = enable != 0;
  return ORBX_OK;
}

int orbx_set_fused_pyramid_blur(orbx_ctx* c, int enable) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  c->fuse = enable != 0;
  return ORBX_OK;
}

int orbx_set_top_rows_first(orbx_ctx* c, int mode) {
  DeviceGuard _dg(c);
  if (!c || mode < 0 || mode > 2) return ORBX_ERR_INVALID_ARG;
  c->top_mode = mode;
  c->top_on = true;
  c->top_single_batches = 0;
  return ORBX_OK;
}

int orbx_fast_tile_counts(orbx_ctx* c, long long* worked, long long* total) {
  DeviceGuard _dg(c);
  if (!c || !worked || !total) return ORBX_ERR_INVALID_ARG;
  if (c->plan_w == 0 || c->last_n < 1) return fail(c, ORBX_ERR_INVALID_ARG, "run a batch first");
  const int n = c->last_n;
  std::vector<unsigned long long> h((size_t)n * ORBX_FAST_STAT_WORDS);
  HIPCHK(c, hipStreamSynchronize(c->last_stream));
  HIPCHK(c, hipMemcpy(h.data(), c->d_row_stat, h.size() * 8, hipMemcpyDeviceToHost));
  long long w = 0;
  for (int f = 0; f < n; f++)
    for (int l = 0; l < c->plan.nlevels; l++)
      for (int b = 0; b < c->bm_fast.tiles_y[l]; b++)
        w += (long long)(h[(size_t)f * ORBX_FAST_STAT_WORDS + (size_t)l * ORBX_MAX_BANDS + b] >> 32);
  *total = (long long)c->bm_fast.band_begin[c->bm_fast.nbands] * n;
  *worked = fast_early_on(c) ? w : *total;  // (no statistics are kept when the early exit is off)
  return ORBX_OK;
}

int orbx_pyramid_pixel_counts(orbx_ctx* c, long long* produced, long long* total) {
  DeviceGuard _dg(c);
  if (!c || !produced || !total) return ORBX_ERR_INVALID_ARG;
  if (c->plan_w == 0 || c->last_n < 1) return fail(c, ORBX_ERR_INVALID_ARG, "run a batch first");
  const int n = c->last_n, top = top_rows_env();
  long long per_frame = 0;
  for (int l = 0; l < c->plan.nlevels; l++) per_frame += (long long)c->plan.L[l].w * c->plan.L[l].h;
  *total = per_frame * n;
  *produced = *total;
  if (!c->last_two_pass) return ORBX_OK;
  std::vector<unsigned long long> h((size_t)n * ORBX_FAST_STAT_WORDS);
  HIPCHK(c, hipStreamSynchronize(c->last_stream));
  HIPCHK(c, hipMemcpy(h.data(), c->d_row_stat, h.size() * 8, hipMemcpyDeviceToHost));
  long long done = 0;
  for (int f = 0; f < n; f++)
    for (int l = 0; l < c->plan.nlevels; l++) {
      const OrbxLevel& L = c->plan.L[l];
      const int first = pyrblur_first_pass_rows(c->plan, c->bm_fast, l, top);
      long long surv = 0;  // the kernel's test: survivors of the first-pass tile rows
      for (int b = 0; b < std::min(top, c->bm_fast.tiles_y[l]); b++)
        surv += (long long)(uint32_t)h[(size_t)f * ORBX_FAST_STAT_WORDS + (size_t)l * ORBX_MAX_BANDS + b];
      done += (long long)L.w * (first < L.h && surv >= L.cap ? first : L.h);
    }
  *produced = done;
  return ORBX_OK;
}

int orbx_enable_stage_timing(orbx_ctx* c, int enable) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  c->timing = enable < 0 || enable > 2 ? 1 : enable;
  return ORBX_OK;
}

int orbx_stage_times_history(orbx_ctx* c, int back, float* ms) {
  DeviceGuard _dg(c);
  if (!c || !ms) return ORBX_ERR_INVALID_ARG;
  if (back < 0 || back >= ORBX_EVENT_SETS || back >= c->ev_calls)
    return fail(c, ORBX_ERR_INVALID_ARG, "no timed batched call that far back");
  const long long call = c->ev_calls - 1 - back;
  hipEvent_t* evs = c->evr[call % ORBX_EVENT_SETS];
  const int mode = c->ev_mode[call % ORBX_EVENT_SETS];
  std::memset(ms, 0, sizeof(float) * ORBX_NUM_STAGE_TIMES);
  if (mode == 1) {
    for (int i = 0; i < ORBX_NUM_STAGE_TIMES - 1; i++) HIPCHK(c, hipEventElapsedTime(&ms[i], evs[i], evs[i + 1]));
    HIPCHK(c, hipEventElapsedTime(&ms[ORBX_NUM_STAGE_TIMES - 1], evs[0], evs[ORBX_NUM_STAGE_TIMES - 1]));
  } else {  // blur and fast+nms only
    HIPCHK(c, hipEventElapsedTime(&ms[1], evs[1], evs[2]));
    HIPCHK(c, hipEventElapsedTime(&ms[2], evs[2], evs[3]));
  }
  if (c->ev_split[call % ORBX_EVENT_SETS]) {
    // top-rows-first pipeline: events 1 | pyramid+blur (top) | N+1 | FAST (top) | N+2 | pyramid+blur (rest) | 2 | FAST (rest) | 3
    float a = 0, b = 0, d = 0, e = 0;
    HIPCHK(c, hipEventElapsedTime(&a, evs[1], evs[ORBX_NUM_STAGE_TIMES + 1]));
    HIPCHK(c, hipEventElapsedTime(&b, evs[ORBX_NUM_STAGE_TIMES + 1], evs[ORBX_NUM_STAGE_TIMES + 2]));
    HIPCHK(c, hipEventElapsedTime(&d, evs[ORBX_NUM_STAGE_TIMES + 2], evs[2]));
    HIPCHK(c, hipEventElapsedTime(&e, evs[2], evs[3]));
    ms[1] = a + d;
    ms[2] = b + e;
  }
  return ORBX_OK;
}

int orbx_last_stage_times(orbx_ctx* c, float* ms) { return orbx_stage_times_history(c, 0, ms); }

int orbx_batch_results_device(orbx_ctx* c, orbx_batch_view* v) {
  DeviceGuard _dg(c);
  if (!c || !v) return ORBX_ERR_INVALID_ARG;
  if (c->last_n <= 0) return fail(c, ORBX_ERR_INVALID_ARG, "no batch has been run");
  const OutLayout& o = c->out_layout;
  v->counts = (const int32_t*)(c->d_out + o.counts);
  v->keypoints = (const orbx_keypoint*)(c->d_out + o.kp);
  v->keypoints16 = (const uint32_t*)(c->d_out + o.kp16);
  v->level_kps = (const orbx_keypoint*)(c->d_out + o.lkp);
  v->orientations = (const float*)(c->d_out + o.angle);
  v->responses = (const float*)(c->d_out + o.resp);
  v->levels = (const int32_t*)(c->d_out + o.level);
  v->descriptors = (const orbx_descriptor*)(c->d_out + o.desc);
  v->slot_capacity = c->plan.out_cap > 0 ? c->plan.out_cap : 1;
  v->n = c->last_n;
  return ORBX_OK;
}

namespace {
// frames [first, first+n) of result block `b` -> host arrays.  If an asynchronous copy of the block
// is pending (orbx_batch_prefetch) only its event is waited for; otherwise ONE blocking D2H of the block.
int fetch_block(orbx_ctx* c, int b, int first, int n, int32_t* counts, orbx_keypoint* keypoints, float* orientations,
                orbx_descriptor* descriptors, float* responses, int32_t* levels, orbx_keypoint* level_kps,
                int capacity) {
  if (!counts) return fail(c, ORBX_ERR_INVALID_ARG, "counts is NULL");
  if (first < 0 || n < 1 || first + n > c->nb[b]) return fail(c, ORBX_ERR_INVALID_ARG, "frame range outside batch");
  if (capacity < 0) return fail(c, ORBX_ERR_INVALID_ARG, "capacity < 0");
  const OutLayout& o = c->layoutb[b];
  const int cap = c->capb[b];
  const uint8_t* h = c->h_outb[b];
  if (c->copy_pending[b] && c->copy_compact[b] && (responses || levels || level_kps)) {
    // only the compact prefix is on its way: fetch the other sections now (blocking)
    HIPCHK(c, hipEventSynchronize(c->ev_copied[b]));
    HIPCHK(c, hipMemcpyAsync(c->h_outb[b] + o.compact, c->d_outb[b] + o.compact, o.total - o.compact, hipMemcpyDeviceToHost,
                             c->cstream));
    HIPCHK(c, hipStreamSynchronize(c->cstream));
    c->copy_compact[b] = false;
  } else if (c->copy_pending[b]) {
    HIPCHK(c, hipEventSynchronize(c->ev_copied[b]));
  } else if (b == c->blk) {
    // blocking fetch of the last batch: the copy goes behind the batch on ITS stream (no hop to the copy
    // stream: the synchronous one-frame call is latency-bound)
    hipStream_t s = c->last_stream ? c->last_stream : c->stream;
    HIPCHK(c, hipMemcpyAsync(c->h_outb[b], c->d_outb[b], o.total, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  } else {
    HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_done[b], 0));
    HIPCHK(c, hipMemcpyAsync(c->h_outb[b], c->d_outb[b], o.total, hipMemcpyDeviceToHost, c->cstream));
    HIPCHK(c, hipStreamSynchronize(c->cstream));
  }
  const int32_t* hc = (const int32_t*)(h + o.counts);
  bool truncated = false;
  for (int i = 0; i < n; i++) {
    const int f = first + i;
    const int cnt = hc[f];
    counts[i] = cnt;
    const int m = std::min(cnt, capacity);
    if (cnt > capacity) truncated = true;
    const size_t so = (size_t)f * cap, dst = (size_t)i * capacity;
    if (keypoints) {  // (a compact copy holds them packed only: x | y << 16)
      const uint32_t* k16 = (const uint32_t*)(h + o.kp16) + so;
      for (int j = 0; j < m; j++) keypoints[dst + j] = orbx_keypoint{(int32_t)(k16[j] & 0xffffu), (int32_t)(k16[j] >> 16)};
    }
    if (level_kps) std::memcpy(level_kps + dst, (const orbx_keypoint*)(h + o.lkp) + so, sizeof(orbx_keypoint) * m);
    if (orientations) std::memcpy(orientations + dst, (const float*)(h + o.angle) + so, sizeof(float) * m);
    if (responses) std::memcpy(responses + dst, (const float*)(h + o.resp) + so, sizeof(float) * m);
    if (levels) std::memcpy(levels + dst, (const int32_t*)(h + o.level) + so, sizeof(int32_t) * m);
    if (descriptors)
      std::memcpy(descriptors + dst, (const orbx_descriptor*)(h + o.desc) + so, sizeof(orbx_descriptor) * m);
  }
  if (truncated) return fail(c, ORBX_ERR_CAPACITY, "capacity smaller than keypoint count");
  return ORBX_OK;
}
}  // namespace

int orbx_batch_fetch(orbx_ctx* c, int first, int n, int32_t* counts, orbx_keypoint* keypoints,
                     float* orientations, orbx_descriptor* descriptors, float* responses, int32_t* levels,
                     orbx_keypoint* level_kps, int capacity) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  return fetch_block(c, c->blk, first, n, counts, keypoints, orientations, descriptors, responses, levels, level_kps,
                     capacity);
}

int orbx_batch_results_host(orbx_ctx* c, int previous, orbx_batch_view* v) {
  DeviceGuard _dg(c);
  if (!c || !v) return ORBX_ERR_INVALID_ARG;
  if (previous < 0 || previous >= orbx_ctx::kBlocks) return fail(c, ORBX_ERR_INVALID_ARG, "previous must be 0..3 (a ring of four result blocks)");
  const int b = (c->blk + orbx_ctx::kBlocks - previous) % orbx_ctx::kBlocks;
  if (c->nb[b] <= 0) return fail(c, ORBX_ERR_INVALID_ARG, previous ? "there is no batch that far back" : "no batch has been run");
  const OutLayout& o = c->layoutb[b];
  if (c->copy_pending[b]) {
    HIPCHK(c, hipEventSynchronize(c->ev_copied[b]));
  } else {
    HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_done[b], 0));
    HIPCHK(c, hipMemcpyAsync(c->h_outb[b], c->d_outb[b], o.total, hipMemcpyDeviceToHost, c->cstream));
    HIPCHK(c, hipEventRecord(c->ev_copied[b], c->cstream));
    c->copy_pending[b] = true;
    HIPCHK(c, hipEventSynchronize(c->ev_copied[b]));
  }
  const uint8_t* h = c->h_outb[b];
  const bool compact = c->copy_compact[b];  // (those sections of the mirror were not copied: NULL in the view)
  v->counts = (const int32_t*)(h + o.counts);
  v->keypoints = compact ? nullptr : (const orbx_keypoint*)(h + o.kp);
  v->keypoints16 = (const uint32_t*)(h + o.kp16);
  v->level_kps = compact ? nullptr : (const orbx_keypoint*)(h + o.lkp);
  v->orientations = (const float*)(h + o.angle);
  v->responses = compact ? nullptr : (const float*)(h + o.resp);
  v->levels = compact ? nullptr : (const int32_t*)(h + o.level);
  v->descriptors = (const orbx_descriptor*)(h + o.desc);
  v->slot_capacity = c->capb[b];
  v->n = c->nb[b];
  return ORBX_OK;
}

namespace {
int prefetch_block(orbx_ctx* c, bool compact) {
  const int b = c->blk;
  if (c->nb[b] <= 0) return fail(c, ORBX_ERR_INVALID_ARG, "no batch has been run");
  const OutLayout& o = c->layoutb[b];
  if (c->copy_pending[b]) {
    if (compact || !c->copy_compact[b]) return ORBX_OK;
    // a compact copy is on its way and the whole block is wanted after all: the other sections follow it
    HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_done[b], 0));  // (host results: the compact part never was on this stream)
    HIPCHK(c, hipMemcpyAsync(c->h_outb[b] + o.compact, c->d_outb[b] + o.compact, o.total - o.compact, hipMemcpyDeviceToHost,
                             c->cstream));
    HIPCHK(c, hipEventRecord(c->ev_copied[b], c->cstream));
    c->copy_compact[b] = false;
    return ORBX_OK;
  }
  // (a block the describe kernel has written its compact record into -- orbx_set_host_results -- never gets here:
  // it is compact-pending from the moment the batch is enqueued)
  HIPCHK(c, hipStreamWaitEvent(c->cstream, c->ev_done[b], 0));
  HIPCHK(c, hipMemcpyAsync(c->h_outb[b], c->d_outb[b], compact ? o.compact : o.total, hipMemcpyDeviceToHost, c->cstream));
  HIPCHK(c, hipEventRecord(c->ev_copied[b], c->cstream));
  c->copy_pending[b] = true;
  c->copy_compact[b] = compact;
  return ORBX_OK;
}
}  // namespace

int orbx_batch_prefetch(orbx_ctx* c) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  return prefetch_block(c, false);
}

int orbx_batch_prefetch_compact(orbx_ctx* c) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  return prefetch_block(c, true);
}

int orbx_batch_fetch_previous(orbx_ctx* c, int first, int n, int32_t* counts, orbx_keypoint* keypoints,
                              float* orientations, orbx_descriptor* descriptors, float* responses, int32_t* levels,
                              orbx_keypoint* level_kps, int capacity) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  const int b = (c->blk + orbx_ctx::kBlocks - 1) % orbx_ctx::kBlocks;
  if (c->nb[b] <= 0) return fail(c, ORBX_ERR_INVALID_ARG, "there is no batch before the last one");
  return fetch_block(c, b, first, n, counts, keypoints, orientations, descriptors, responses, levels, level_kps,
                     capacity);
}

int orbx_detect_and_compute(orbx_ctx* c, const uint8_t* image, int width, int height, int stride,
                            orbx_keypoint* keypoints, float* orientations, orbx_descriptor* descriptors,
                            float* responses, int32_t* levels, orbx_keypoint* level_kps, int capacity,
                            int* count) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!count) return fail(c, ORBX_ERR_INVALID_ARG, "count is NULL");
  int st = orbx_detect_and_compute_batch_host(c, image, 1, width, height, stride, (size_t)stride * height);
  if (st != ORBX_OK) return st;
  int32_t cnt = 0;
  st = orbx_batch_fetch(c, 0, 1, &cnt, keypoints, orientations, descriptors, responses, levels, level_kps,
                        capacity);
  *count = cnt;
  return st;
}

int orbx_bench_stage(orbx_ctx* c, int n_frames, int stage, int reps, float* avg_ms) {
  DeviceGuard _dg(c);
  if (!c || !avg_ms) return ORBX_ERR_INVALID_ARG;
  if (c->plan_w == 0) return fail(c, ORBX_ERR_INVALID_ARG, "run a batch first (no pyramid built)");
  if (n_frames < 1 || n_frames > c->last_n || reps < 1) return fail(c, ORBX_ERR_INVALID_ARG, "n_frames/reps");
  const OrbxPlan& P = c->plan;
  hipStream_t s = c->stream;
  OrbxFastParams fp{c->p.threshold, c->p.n, c->p.nms_window / 2};
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, lanes_sync(c));  // (the last batch may have run on the other lane's stream)
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  for (int i = 0; i < reps; i++) {
    switch (stage) {
      case ORBX_STAGE_BLUR:
        if (!blur_enabled(c)) return fail(c, ORBX_ERR_INVALID_ARG, "blur is disabled in this context");
        HIPCHK(c, launch_blur_auto(c->blur_impl, s, P, c->tm_blur, c->d_tiles_blur, c->blur_tiles_count, n_frames, c->d_pyr,
                                   c->d_pyr_blur,
                                   c->p.blur_levels == ORBX_BLUR_UPPER ? 1 : 0, c->p.blur_kind));
        break;
      case ORBX_STAGE_FAST:
        HIPCHK(c, launch_fast_whole(c, s, n_frames, fp));
        break;
      case ORBX_STAGE_COMPACT:
        HIPCHK(c, orbx_launch_compact(s, P, n_frames, c->d_mask, c->d_cand, c->d_cand_count, c->d_cand_total, 0));
        break;
      default:
        return fail(c, ORBX_ERR_INVALID_ARG, "stage not benchmarkable in isolation");
    }
  }
  HIPCHK(c, hipEventRecord(c->ev[1], s));
  HIPCHK(c, hipEventSynchronize(c->ev[1]));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  *avg_ms = ms / reps;
  return ORBX_OK;
}

// ---- stage-level operators --------------------------------------------------

int orbx_fast_score(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, int threshold, int n,
                    float* scores) {
  DeviceGuard _dg(c);
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (!scores || n < 1 || n > 16 || threshold < 0 || threshold > 255)
    return fail(c, ORBX_ERR_INVALID_ARG, "scores NULL or n/threshold out of range");
  int pitch;
  st = upload_flat(c, c->s_img_a, image, width, height, stride, &pitch);
  if (st != ORBX_OK) return st;
  OrbxPlan P = flat_plan(width, height, 0);
  OrbxBandMap bm;
  std::string why;
  if ((st = make_bandmap(P, 0, &bm, &why)) != ORBX_OK) return fail(c, st, why);
  std::vector<OrbxTileDesc> t;
  build_fast_tiles(P, bm, 0, 1, &t);
  if ((st = ensure(c, c->s_tiles, t.size() * sizeof(OrbxTileDesc))) != ORBX_OK) return st;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->s_tiles.p, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
  const size_t npx = (size_t)width * height;
  if ((st = ensure(c, c->s_u16, npx * 2)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_mask, (size_t)P.mask_words * 8)) != ORBX_OK) return st;
  OrbxFastParams fp{threshold, n, 0};
  HIPCHK(c, orbx_launch_fast_nms(c->stream, (const OrbxTileDesc*)c->s_tiles.p, (int)t.size(), 1,
                                 (const uint8_t*)c->s_img_a.p, P.frame_bytes, P.mask_words, fp,
                                 (unsigned long long*)c->s_mask.p, (uint16_t*)c->s_u16.p, nullptr));
  std::vector<uint16_t> h(npx);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->s_u16.p, npx * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < npx; i++) scores[i] = (float)h[i];
  return ORBX_OK;
}

static int compact_and_fetch(orbx_ctx* c, const OrbxPlan& P, int nfeatures, orbx_keypoint* keypoints, int* count,
                             int* total) {
  int st;
  if ((st = ensure(c, c->s_kps, sizeof(orbx_keypoint) * (size_t)std::max(nfeatures, 1))) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_i32, 64)) != ORBX_OK) return st;
  int32_t* d_cnt = (int32_t*)c->s_i32.p;
  HIPCHK(c, orbx_launch_compact(c->stream, P, 1, (const unsigned long long*)c->s_mask.p, (orbx_keypoint*)c->s_kps.p,
                                d_cnt, d_cnt + 1, 1));
  int32_t h[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h[0] > 0)
    HIPCHK(c, hipMemcpy(keypoints, c->s_kps.p, sizeof(orbx_keypoint) * (size_t)h[0], hipMemcpyDeviceToHost));
  *count = h[0];
  if (total) *total = h[1];
  return ORBX_OK;
}

int orbx_fast(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, int threshold, int n,
              int nms_window, int nfeatures, orbx_keypoint* keypoints, int* count, int* total) {
  DeviceGuard _dg(c);
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (!count || (!keypoints && nfeatures > 0) || nfeatures < 0 || n < 1 || n > 16 || threshold < 0 ||
      threshold > 255 || nms_window < 0 || nms_window / 2 > 3)
    return fail(c, ORBX_ERR_INVALID_ARG, "bad Fast() arguments");
  int pitch;
  st = upload_flat(c, c->s_img_a, image, width, height, stride, &pitch);
  if (st != ORBX_OK) return st;
  OrbxPlan P = flat_plan(width, height, nfeatures);
  OrbxBandMap bm;
  std::string why;
  if ((st = make_bandmap(P, nms_window / 2, &bm, &why)) != ORBX_OK) return fail(c, st, why);
  std::vector<OrbxTileDesc> t;
  build_fast_tiles(P, bm, 0, 1, &t);
  if ((st = ensure(c, c->s_tiles, t.size() * sizeof(OrbxTileDesc))) != ORBX_OK) return st;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->s_tiles.p, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
  if ((st = ensure(c, c->s_mask, (size_t)P.mask_words * 8)) != ORBX_OK) return st;
  OrbxFastParams fp{threshold, n, nms_window / 2};
  // stage operator: exact totals are part of the contract -> no early exit
  HIPCHK(c, orbx_launch_fast_nms(c->stream, (const OrbxTileDesc*)c->s_tiles.p, (int)t.size(), 1,
                                 (const uint8_t*)c->s_img_a.p, P.frame_bytes, P.mask_words, fp,
                                 (unsigned long long*)c->s_mask.p, nullptr, nullptr));
  return compact_and_fetch(c, P, nfeatures, keypoints, count, total);
}

int orbx_nms(orbx_ctx* c, const float* scores, int width, int height, int nms_window, int nfeatures,
             float threshold, orbx_keypoint* keypoints, int* count, int* total) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!scores || !count || (!keypoints && nfeatures > 0) || nfeatures < 0 || width < 1 || height < 1 ||
      nms_window < 0 || nms_window / 2 > 3)
    return fail(c, ORBX_ERR_INVALID_ARG, "bad NMS() arguments");
  int st;
  const size_t npx = (size_t)width * height;
  if ((st = ensure(c, c->s_f32, npx * 4)) != ORBX_OK) return st;
  OrbxPlan P = flat_plan(width, height, nfeatures);
  if ((st = ensure(c, c->s_mask, (size_t)P.mask_words * 8)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->s_f32.p, scores, npx * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, orbx_launch_nms_f32(c->stream, (const float*)c->s_f32.p, width, height, nms_window / 2, threshold,
                                (unsigned long long*)c->s_mask.p, P.L[0].mask_wpr));
  return compact_and_fetch(c, P, nfeatures, keypoints, count, total);
}

static int describe_stage(orbx_ctx* c, const uint8_t* image, int width, int height, int stride,
                          const orbx_keypoint* keypoints, int nkp, int patch_size, const float* angles_in,
                          float* angles_out, orbx_descriptor* desc_out) {
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (nkp < 0 || (nkp > 0 && !keypoints)) return fail(c, ORBX_ERR_INVALID_ARG, "keypoints NULL / nkp < 0");
  if (patch_size < 1 || patch_size / 2 > 20) return fail(c, ORBX_ERR_INVALID_ARG, "patch_size must be in [1, 41]");
  if (nkp == 0) return ORBX_OK;
  for (int i = 0; i < nkp; i++)
    if (keypoints[i].x < 0 || keypoints[i].y < 0 || keypoints[i].x >= width || keypoints[i].y >= height)
      return fail(c, ORBX_ERR_INVALID_ARG, "keypoint outside the image");
  if (angles_in)
    for (int i = 0; i < nkp; i++)
      if (!(std::fabs(angles_in[i]) < 100.0f))
        return fail(c, ORBX_ERR_INVALID_ARG, "orientation must be finite and |angle| < 100 rad");
  int pitch;
  st = upload_flat(c, c->s_img_a, image, width, height, stride, &pitch);
  if (st != ORBX_OK) return st;
  if ((st = ensure(c, c->s_kps, sizeof(orbx_keypoint) * (size_t)nkp)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_f32b, sizeof(float) * (size_t)nkp)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_desc, sizeof(orbx_descriptor) * (size_t)nkp)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->s_kps.p, keypoints, sizeof(orbx_keypoint) * (size_t)nkp, hipMemcpyHostToDevice,
                           c->stream));
  if (angles_in)
    HIPCHK(c, hipMemcpyAsync(c->s_f32b.p, angles_in, sizeof(float) * (size_t)nkp, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, orbx_launch_describe_flat(c->stream, (const uint8_t*)c->s_img_a.p, width, height, pitch,
                                      (const orbx_keypoint*)c->s_kps.p, nkp, patch_size, angles_in != nullptr,
                                      desc_out != nullptr, (float*)c->s_f32b.p, (orbx_descriptor*)c->s_desc.p));
  if (angles_out)
    HIPCHK(c, hipMemcpyAsync(angles_out, c->s_f32b.p, sizeof(float) * (size_t)nkp, hipMemcpyDeviceToHost, c->stream));
  if (desc_out)
    HIPCHK(c, hipMemcpyAsync(desc_out, c->s_desc.p, sizeof(orbx_descriptor) * (size_t)nkp, hipMemcpyDeviceToHost,
                             c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ORBX_OK;
}

int orbx_orientations(orbx_ctx* c, const uint8_t* image, int width, int height, int stride,
                      const orbx_keypoint* keypoints, int nkp, int patch_size, float* orientations) {
  DeviceGuard _dg(c);
  if (c && nkp > 0 && !orientations) return fail(c, ORBX_ERR_INVALID_ARG, "orientations is NULL");
  return describe_stage(c, image, width, height, stride, keypoints, nkp, patch_size, nullptr, orientations, nullptr);
}

int orbx_brief(orbx_ctx* c, const uint8_t* image, int width, int height, int stride,
               const orbx_keypoint* keypoints, const float* orientations, int nkp, orbx_descriptor* descriptors) {
  DeviceGuard _dg(c);
  if (c && nkp > 0 && (!orientations || !descriptors))
    return fail(c, ORBX_ERR_INVALID_ARG, "orientations/descriptors is NULL");
  return describe_stage(c, image, width, height, stride, keypoints, nkp, 31, orientations, nullptr, descriptors);
}

int orbx_harris(orbx_ctx* c, const uint8_t* image, int width, int height, int stride,
                const orbx_keypoint* keypoints, int nkp, int window, float k, float* responses) {
  DeviceGuard _dg(c);
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (nkp < 0 || (nkp > 0 && (!keypoints || !responses)) || window < 1 || (window % 2) == 0 || window > 15)
    return fail(c, ORBX_ERR_INVALID_ARG, "bad HarrisScore() arguments");
  if (nkp == 0) return ORBX_OK;
  for (int i = 0; i < nkp; i++)
    if (keypoints[i].x < 0 || keypoints[i].y < 0 || keypoints[i].x >= width || keypoints[i].y >= height)
      return fail(c, ORBX_ERR_INVALID_ARG, "keypoint outside the image");
  int pitch;
  st = upload_flat(c, c->s_img_a, image, width, height, stride, &pitch);
  if (st != ORBX_OK) return st;
  std::vector<float> g((size_t)window * window);
  gaussian_kernel(window, -1.0f, g.data());
  if ((st = ensure(c, c->s_kern, g.size() * 4)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_kps, sizeof(orbx_keypoint) * (size_t)nkp)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_f32b, sizeof(float) * (size_t)nkp)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->s_kern.p, g.data(), g.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->s_kps.p, keypoints, sizeof(orbx_keypoint) * (size_t)nkp, hipMemcpyHostToDevice,
                           c->stream));
  HIPCHK(c, orbx_launch_harris_flat(c->stream, (const uint8_t*)c->s_img_a.p, width, height, pitch,
                                    (const orbx_keypoint*)c->s_kps.p, nkp, (const float*)c->s_kern.p, window, k,
                                    (float*)c->s_f32b.p));
  HIPCHK(c, hipMemcpyAsync(responses, c->s_f32b.p, sizeof(float) * (size_t)nkp, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ORBX_OK;
}

static int blur_stage(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, uint8_t* dst,
                      int dst_stride, int kind) {
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (!dst || dst_stride < width) return fail(c, ORBX_ERR_INVALID_ARG, "dst NULL or dst_stride < width");
  int pitch;
  st = upload_flat(c, c->s_img_a, image, width, height, stride, &pitch);
  if (st != ORBX_OK) return st;
  OrbxPlan P = flat_plan(width, height, 0);
  if ((st = ensure(c, c->s_img_b, (size_t)P.frame_bytes + 256)) != ORBX_OK) return st;
  OrbxTileMap tm;
  make_tilemap(P, ORBX_BLUR_TW, ORBX_BLUR_TH, true, &tm);
  std::vector<OrbxTileDesc> t;
  blur_tiles_for_impl(c->blur_impl, P, &t);
  if ((st = ensure(c, c->s_tiles, t.size() * sizeof(OrbxTileDesc))) != ORBX_OK) return st;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->s_tiles.p, t.data(), t.size() * sizeof(OrbxTileDesc), hipMemcpyHostToDevice));
  HIPCHK(c, launch_blur_auto(c->blur_impl, c->stream, P, tm, (const OrbxTileDesc*)c->s_tiles.p, (int)t.size(), 1,
                             (const uint8_t*)c->s_img_a.p, (uint8_t*)c->s_img_b.p, 0, kind));
  HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, c->s_img_b.p, pitch, width, height, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ORBX_OK;
}

int orbx_blur5_sep(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, uint8_t* dst,
                   int dst_stride) {
  DeviceGuard _dg(c);
  return blur_stage(c, image, width, height, stride, dst, dst_stride, ORBX_BLUR_SEP16);
}

int orbx_blur5_273(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, uint8_t* dst,
                   int dst_stride) {
  DeviceGuard _dg(c);
  return blur_stage(c, image, width, height, stride, dst, dst_stride, ORBX_BLUR_K273);
}

static int conv_stage(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, const float* kernel,
                      int K, int reflect_pad, uint8_t* dst) {
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!image || !kernel || !dst || width < 1 || height < 1 || stride < width || K < 1 || (K % 2) == 0 || K > 31)
    return fail(c, ORBX_ERR_INVALID_ARG, "bad conv2d() arguments (kernel_size must be odd, <= 31)");
  const int wo = reflect_pad ? width : width - K + 1, ho = reflect_pad ? height : height - K + 1;
  if (wo < 1 || ho < 1) return fail(c, ORBX_ERR_INVALID_ARG, "image smaller than the kernel");
  if (reflect_pad && (width < K / 2 + 1 || height < K / 2 + 1))
    return fail(c, ORBX_ERR_INVALID_ARG, "image too small for REFLECT_101 padding");
  int st, pitch;
  const int p = align_up(width, 64);
  if ((st = ensure(c, c->s_img_a, (size_t)p * height + 256)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpy2DAsync(c->s_img_a.p, p, image, stride, width, height, hipMemcpyHostToDevice, c->stream));
  pitch = p;
  if ((st = ensure(c, c->s_img_b, (size_t)wo * ho + 256)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_kern, (size_t)K * K * 4)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->s_kern.p, kernel, (size_t)K * K * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, orbx_launch_conv2d(c->stream, (const uint8_t*)c->s_img_a.p, width, height, pitch,
                               (const float*)c->s_kern.p, K, reflect_pad, (uint8_t*)c->s_img_b.p, wo));
  HIPCHK(c, hipMemcpyAsync(dst, c->s_img_b.p, (size_t)wo * ho, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ORBX_OK;
}

int orbx_conv2d(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, const float* kernel,
                int kernel_size, uint8_t* dst) {
  DeviceGuard _dg(c);
  return conv_stage(c, image, width, height, stride, kernel, kernel_size, 0, dst);
}

int orbx_gaussian_kernel(int kernel_size, float sigma, float* kernel) {
  return gaussian_kernel(kernel_size, sigma, kernel);
}

int orbx_gaussian_blur_conv(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, int kernel_size,
                            uint8_t* dst) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (kernel_size < 1 || (kernel_size % 2) == 0 || kernel_size > 31)
    return fail(c, ORBX_ERR_INVALID_ARG, "kernel_size must be odd and <= 31 (src/GaussianBlur.cpp:8-11)");
  std::vector<float> g((size_t)kernel_size * kernel_size);
  gaussian_kernel(kernel_size, -1.0f, g.data());
  return conv_stage(c, image, width, height, stride, g.data(), kernel_size, 1, dst);
}

int orbx_sobel(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, int dir, uint8_t* dst) {
  DeviceGuard _dg(c);
  static const float SX[9] = {-1.f, 0.f, 1.f, -2.f, 0.f, 2.f, -1.f, 0.f, 1.f};   // src/Sobel.cpp:6-10
  static const float SY[9] = {-1.f, -2.f, -1.f, 0.f, 0.f, 0.f, 1.f, 2.f, 1.f};   // src/Sobel.cpp:12-16
  return conv_stage(c, image, width, height, stride, dir == 0 ? SX : SY, 3, 1, dst);
}

int orbx_build_pyramid_level(orbx_ctx* c, const uint8_t* image, int width, int height, int stride, int level,
                             uint8_t* dst, int* level_w, int* level_h) {
  DeviceGuard _dg(c);
  int st = check_image(c, image, width, height, stride);
  if (st != ORBX_OK) return st;
  if (level < 0 || level >= c->p.nlevels || !dst) return fail(c, ORBX_ERR_INVALID_ARG, "level out of range / dst NULL");
  if ((st = set_plan(c, width, height)) != ORBX_OK) return st;
  const OrbxPlan& P = c->plan;
  HIPCHK(c, lanes_sync(c));
  HIPCHK(c, hipMemcpy2DAsync(c->d_in, width, image, stride, width, height, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_pyramid_auto(c, c->stream, 1, c->d_in, width, (size_t)width * height));
  if (blur_enabled(c))
    HIPCHK(c, launch_blur_auto(c->blur_impl, c->stream, P, c->tm_blur, c->d_tiles_blur, c->blur_tiles_count, 1, c->d_pyr,
                               c->d_pyr_blur,
                               c->p.blur_levels == ORBX_BLUR_UPPER ? 1 : 0, c->p.blur_kind));
  const OrbxLevel& L = P.L[level];
  HIPCHK(c, hipMemcpy2DAsync(dst, L.w, final_pyr(c) + L.img_off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost,
                             c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (level_w) *level_w = L.w;
  if (level_h) *level_h = L.h;
  return ORBX_OK;
}

int orbx_select_top(orbx_ctx* c, const float* responses, int n, int keep, int32_t* indices, int* kept) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (n < 0 || keep < 0 || (n > 0 && (!responses || !indices)) || !kept)
    return fail(c, ORBX_ERR_INVALID_ARG, "bad select_top arguments");
  const int m = std::min(n, keep);
  *kept = m;
  if (m == 0) return ORBX_OK;
  int st;
  if ((st = ensure(c, c->s_f32b, sizeof(float) * (size_t)n)) != ORBX_OK) return st;
  if ((st = ensure(c, c->s_i32, sizeof(int32_t) * (size_t)n)) != ORBX_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->s_f32b.p, responses, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, orbx_launch_select_flat(c->stream, (const float*)c->s_f32b.p, n, keep, (int32_t*)c->s_i32.p));
  HIPCHK(c, hipMemcpyAsync(indices, c->s_i32.p, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ORBX_OK;
}

// ---- descriptor matching (next row) -----------------------------------------

static int knn_host(orbx_ctx* c, const orbx_descriptor* query, int nq, const orbx_descriptor* train, int nt,
                    double ratio, std::vector<int32_t>* idx, std::vector<int32_t>* dist,
                    std::vector<int32_t>* match) {
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (nq < 0 || nt < 0 || (nq > 0 && !query) || (nt > 0 && !train))
    return fail(c, ORBX_ERR_INVALID_ARG, "bad matcher arguments");
  idx->assign((size_t)2 * nq, -1);
  dist->assign((size_t)2 * nq, -1);
  match->assign((size_t)nq, -1);
  if (nq == 0) return ORBX_OK;
  int st;
  if ((st = ensure(c, c->m_q, sizeof(orbx_descriptor) * (size_t)nq)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_t, sizeof(orbx_descriptor) * (size_t)std::max(nt, 1))) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_idx, sizeof(int32_t) * 2 * (size_t)nq)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_dist, sizeof(int32_t) * 2 * (size_t)nq)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_match, sizeof(int32_t) * (size_t)nq)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_cnt, 64)) != ORBX_OK) return st;
  const int32_t cnt[2] = {nq, nt};
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->m_cnt.p, cnt, sizeof(cnt), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->m_q.p, query, sizeof(orbx_descriptor) * (size_t)nq, hipMemcpyHostToDevice, s));
  if (nt > 0) HIPCHK(c, hipMemcpyAsync(c->m_t.p, train, sizeof(orbx_descriptor) * (size_t)nt, hipMemcpyHostToDevice, s));
  HIPCHK(c, orbx_launch_knn2(s, 1, nq, (const orbx_descriptor*)c->m_q.p, (const int32_t*)c->m_cnt.p, 0,
                             (const orbx_descriptor*)c->m_t.p, (const int32_t*)c->m_cnt.p + 1, 0, ratio,
                             (int32_t*)c->m_idx.p, (int32_t*)c->m_dist.p, (int32_t*)c->m_match.p, 0));
  HIPCHK(c, hipMemcpyAsync(idx->data(), c->m_idx.p, sizeof(int32_t) * 2 * (size_t)nq, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(dist->data(), c->m_dist.p, sizeof(int32_t) * 2 * (size_t)nq, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(match->data(), c->m_match.p, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  c->match_pairs = 0;  // the scratch no longer holds a batch's matches
  return ORBX_OK;
}

static int compact_matches(orbx_ctx* c, const int32_t* match, const int32_t* dist2, int nq, int32_t* query_idx,
                           int32_t* train_idx, int32_t* dist1, int capacity, int* count) {
  int n = 0;
  for (int i = 0; i < nq; i++)
    if (match[i] >= 0) {
      if (n < capacity) {
        query_idx[n] = i;
        train_idx[n] = match[i];
        if (dist1) dist1[n] = dist2[2 * i];
      }
      n++;
    }
  *count = n;
  return n > capacity ? fail(c, ORBX_ERR_CAPACITY, "capacity smaller than match count") : (int)ORBX_OK;
}

int orbx_knn2(orbx_ctx* c, const orbx_descriptor* query, int nq, const orbx_descriptor* train, int nt,
              int32_t* idx, int32_t* dist) {
  DeviceGuard _dg(c);
  if (c && nq > 0 && (!idx || !dist)) return fail(c, ORBX_ERR_INVALID_ARG, "idx/dist is NULL");
  std::vector<int32_t> vi, vd, vm;
  int st = knn_host(c, query, nq, train, nt, 0.8, &vi, &vd, &vm);
  if (st != ORBX_OK) return st;
  if (nq > 0) {
    std::memcpy(idx, vi.data(), sizeof(int32_t) * 2 * (size_t)nq);
    std::memcpy(dist, vd.data(), sizeof(int32_t) * 2 * (size_t)nq);
  }
  return ORBX_OK;
}

int orbx_match_ratio(orbx_ctx* c, const orbx_descriptor* query, int nq, const orbx_descriptor* train, int nt,
                     double ratio, int32_t* query_idx, int32_t* train_idx, int32_t* dist1, int capacity, int* count) {
  DeviceGuard _dg(c);
  if (c && (!count || capacity < 0 || (capacity > 0 && (!query_idx || !train_idx))))
    return fail(c, ORBX_ERR_INVALID_ARG, "bad match output arguments");
  std::vector<int32_t> vi, vd, vm;
  int st = knn_host(c, query, nq, train, nt, ratio, &vi, &vd, &vm);
  if (st != ORBX_OK) return st;
  return compact_matches(c, vm.data(), vd.data(), nq, query_idx, train_idx, dist1, capacity, count);
}

int orbx_batch_match_consecutive(orbx_ctx* c, double ratio) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (c->last_n < 2) return fail(c, ORBX_ERR_INVALID_ARG, "needs a batch of at least two frames");
  // the match buffers are ONE set per context: a match of the other lane's batch may still be writing them
  if (c->lane_stream[1]) HIPCHK(c, lanes_sync(c));
  const int n = c->last_n, cap = c->plan.out_cap > 0 ? c->plan.out_cap : 1;
  const size_t e = (size_t)(n - 1) * cap;
  int st;
  if ((st = ensure(c, c->m_idx, sizeof(int32_t) * 2 * e)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_dist, sizeof(int32_t) * 2 * e)) != ORBX_OK) return st;
  if ((st = ensure(c, c->m_match, sizeof(int32_t) * e)) != ORBX_OK) return st;
  const OutLayout& o = c->out_layout;
  const int32_t* counts = (const int32_t*)(c->d_out + o.counts);
  const orbx_descriptor* desc = (const orbx_descriptor*)(c->d_out + o.desc);
  hipStream_t s = c->last_stream ? c->last_stream : c->stream;
  // pair p: query = frame p, train = frame p+1 (same arrays, shifted by one slot block)
  HIPCHK(c, orbx_launch_knn2(s, n - 1, cap, desc, counts, (size_t)cap, desc + cap, counts + 1, (size_t)cap, ratio,
                             (int32_t*)c->m_idx.p, (int32_t*)c->m_dist.p, (int32_t*)c->m_match.p, (size_t)cap));
  c->match_pairs = n - 1;
  return ORBX_OK;
}

int orbx_batch_match_fetch(orbx_ctx* c, int pair, int32_t* query_idx, int32_t* train_idx, int32_t* dist1,
                           int capacity, int* count) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!count || capacity < 0 || (capacity > 0 && (!query_idx || !train_idx)))
    return fail(c, ORBX_ERR_INVALID_ARG, "bad match output arguments");
  if (pair < 0 || pair >= c->match_pairs) return fail(c, ORBX_ERR_INVALID_ARG, "pair outside the last matched batch");
  const int cap = c->plan.out_cap > 0 ? c->plan.out_cap : 1;
  hipStream_t s = c->last_stream ? c->last_stream : c->stream;
  int32_t nq = 0;
  std::vector<int32_t> vm((size_t)cap), vd((size_t)2 * cap);
  HIPCHK(c, hipMemcpyAsync(&nq, (const int32_t*)(c->d_out + c->out_layout.counts) + pair, sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(vm.data(), (const int32_t*)c->m_match.p + (size_t)pair * cap, sizeof(int32_t) * cap,
                           hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(vd.data(), (const int32_t*)c->m_dist.p + (size_t)2 * pair * cap, sizeof(int32_t) * 2 * cap,
                           hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  return compact_matches(c, vm.data(), vd.data(), nq, query_idx, train_idx, dist1, capacity, count);
}

}  // extern "C"

// ---- pyramidal Lucas-Kanade tracking (src/feature_tracking.cpp:166-193) ------------------

namespace {
struct LkGeom {
  int top = 0;
  int w[ORBX_LK_MAX_LEVELS], h[ORBX_LK_MAX_LEVELS], pitch[ORBX_LK_MAX_LEVELS];
  size_t img_off[ORBX_LK_MAX_LEVELS], der_off[ORBX_LK_MAX_LEVELS];
  size_t img_bytes = 0, der_bytes = 0;
};
// buildOpticalFlowPyramid: halve ((w+1)/2) until a level would not be larger than the window
LkGeom lk_geometry(int w, int h, int win, int max_level) {
  LkGeom g;
  for (int l = 0; l <= max_level; l++) {
    const int lw = l == 0 ? w : (g.w[l - 1] + 1) / 2, lh = l == 0 ? h : (g.h[l - 1] + 1) / 2;
    if (l > 0 && (lw <= win || lh <= win)) break;
    g.w[l] = lw;
    g.h[l] = lh;
    g.pitch[l] = lw;  // tight: a host image with stride == width goes up in ONE contiguous copy
    g.img_off[l] = g.img_bytes;
    g.img_bytes += align_up_sz((size_t)g.pitch[l] * lh, 256);
    g.der_off[l] = g.der_bytes;
    g.der_bytes += align_up_sz((size_t)lw * lh * 4, 256);
    g.top = l;
  }
  return g;
}
OrbxLkPyr lk_pyr(const LkGeom& g, const uint8_t* img, const uint8_t* deriv) {
  OrbxLkPyr P;
  std::memset(&P, 0, sizeof(P));
  P.top = g.top;
  for (int l = 0; l <= g.top; l++) {
    P.L[l].img = img + g.img_off[l];
    P.L[l].deriv = deriv ? reinterpret_cast<const int16_t*>(deriv + g.der_off[l]) : nullptr;
    P.L[l].w = g.w[l];
    P.L[l].h = g.h[l];
    P.L[l].pitch = g.pitch[l];
  }
  return P;
}
// host image -> level 0, then pyrDown level by level
int lk_upload(orbx_ctx* c, const LkGeom& g, DevBuf& b, const uint8_t* img, int stride) {
  int st = ensure(c, b, g.img_bytes + 256);
  if (st != ORBX_OK) return st;
  uint8_t* base = (uint8_t*)b.p;
  if (stride == g.w[0])
    HIPCHK(c, hipMemcpyAsync(base, img, (size_t)g.w[0] * g.h[0], hipMemcpyHostToDevice, c->stream));
  else  // (row-by-row in the runtime: slow, but only for padded host images)
    HIPCHK(c, hipMemcpy2DAsync(base, g.pitch[0], img, stride, g.w[0], g.h[0], hipMemcpyHostToDevice, c->stream));
  for (int l = 1; l <= g.top; l++)
    HIPCHK(c, orbx_launch_lk_pyrdown(c->stream, base + g.img_off[l - 1], g.w[l - 1], g.h[l - 1], g.pitch[l - 1],
                                     base + g.img_off[l], g.w[l], g.h[l], g.pitch[l]));
  return ORBX_OK;
}
}  // namespace

extern "C" {

int orbx_lk_track(orbx_ctx* c, const uint8_t* prev, int prev_stride, const uint8_t* next, int next_stride, int width,
                  int height, const float* prev_pts_xy, int n, float* next_pts_xy, uint8_t* status, float* err,
                  int win_size, int max_level, int max_iters, double epsilon) {
  DeviceGuard _dg(c);
  if (!c) return ORBX_ERR_INVALID_ARG;
  if (!next || n < 0 || (n > 0 && (!prev_pts_xy || !next_pts_xy || !status)))
    return fail(c, ORBX_ERR_INVALID_ARG, "next image / point arrays are NULL");
  if (width < 1 || height < 1 || next_stride < width || (prev && prev_stride < width))
    return fail(c, ORBX_ERR_INVALID_ARG, "bad image size or stride");
  if (win_size < 3 || win_size > 31 || max_level < 0 || max_level >= ORBX_LK_MAX_LEVELS)
    return fail(c, ORBX_ERR_INVALID_ARG, "win_size must be in [3, 31], max_level in [0, 7]");
  // TermCriteria sanitising of calcOpticalFlowPyrLK
  max_iters = std::min(std::max(max_iters, 0), 100);
  epsilon = std::min(std::max(epsilon, 0.0), 10.0);
  const LkGeom g = lk_geometry(width, height, win_size, max_level);
  int st;
  int ip;  // buffer holding the `prev` pyramid
  if (prev) {
    ip = c->lk_last == 0 ? 1 : 0;
    if ((st = lk_upload(c, g, c->lk_img[ip], prev, prev_stride)) != ORBX_OK) return st;
  } else {
    // the previous call's `next` image is this call's `prev` (img1 = img2.clone(), feature_tracking.cpp:112)
    if (c->lk_last < 0 || c->lk_w != width || c->lk_h != height || c->lk_top != g.top || c->lk_win != win_size)
      return fail(c, ORBX_ERR_INVALID_ARG, "prev == NULL needs a previous orbx_lk_track call of the same geometry");
    ip = c->lk_last;
  }
  const int in = 1 - ip;
  c->lk_last = -1;  // invalid until this call has succeeded
  if ((st = lk_upload(c, g, c->lk_img[in], next, next_stride)) != ORBX_OK) return st;
  if ((st = ensure(c, c->lk_deriv, g.der_bytes + 256)) != ORBX_OK) return st;
  const uint8_t* pimg = (const uint8_t*)c->lk_img[ip].p;
  for (int l = 0; l <= g.top; l++)
    HIPCHK(c, orbx_launch_lk_scharr(c->stream, pimg + g.img_off[l], g.w[l], g.h[l], g.pitch[l],
                                    reinterpret_cast<int16_t*>((uint8_t*)c->lk_deriv.p + g.der_off[l])));
  uint8_t* hio = nullptr;
  const size_t o_out = sizeof(float) * 2 * (size_t)n, o_err = 2 * o_out, o_st = o_err + sizeof(float) * (size_t)n;
  const size_t io_bytes = o_st + (size_t)n;
  if (n > 0) {
    if ((st = ensure(c, c->lk_io, io_bytes)) != ORBX_OK) return st;
    if (c->lk_host_bytes < io_bytes) {  // pinned staging: small pageable copies cost ~15 us each
      if (c->lk_host) (void)hipHostFree(c->lk_host);
      c->lk_host = nullptr;
      c->lk_host_bytes = 0;
      HIPCHK(c, hipHostMalloc(&c->lk_host, align_up_sz(io_bytes, 4096), hipHostMallocDefault));
      c->lk_host_bytes = align_up_sz(io_bytes, 4096);
    }
    hio = (uint8_t*)c->lk_host;
    uint8_t* dio = (uint8_t*)c->lk_io.p;
    std::memcpy(hio, prev_pts_xy, o_out);
    HIPCHK(c, hipMemcpyAsync(dio, hio, o_out, hipMemcpyHostToDevice, c->stream));
    const OrbxLkPyr P = lk_pyr(g, pimg, (const uint8_t*)c->lk_deriv.p);
    const OrbxLkPyr N = lk_pyr(g, (const uint8_t*)c->lk_img[in].p, nullptr);
    HIPCHK(c, orbx_launch_lk_track(c->stream, P, N, n, (const float*)dio, (float*)(dio + o_out), dio + o_st,
                                   (float*)(dio + o_err), win_size, max_iters, epsilon * epsilon));
    HIPCHK(c, hipMemcpyAsync(hio + o_out, dio + o_out, io_bytes - o_out, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n > 0) {
    std::memcpy(next_pts_xy, hio + o_out, o_out);
    std::memcpy(status, hio + o_st, (size_t)n);
    if (err) std::memcpy(err, hio + o_err, sizeof(float) * (size_t)n);
  }
  c->lk_w = width;
  c->lk_h = height;
  c->lk_top = g.top;
  c->lk_win = win_size;
  c->lk_last = in;
  return ORBX_OK;
}

int orbx_lk_pyramid_levels(int width, int height, int win_size, int max_level) {
  if (width < 1 || height < 1 || win_size < 3 || win_size > 31 || max_level < 0 || max_level >= ORBX_LK_MAX_LEVELS)
    return -1;
  return lk_geometry(width, height, win_size, max_level).top + 1;
}

}  // extern "C"
