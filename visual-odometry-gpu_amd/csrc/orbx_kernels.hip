// orbx_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the ORB front-end.
//
// Design notes (see DESIGN.md §6 for the full picture and the measurements):
//  * 64-lane wavefronts everywhere.  Cross-lane work uses DPP (wave_shr/shl for
//    the blur's left/right neighbours, row scans and reductions, the Harris sum
//    hand-off) and 64-bit ballots (4 ballots == one 256-bit BRIEF descriptor).
//  * Pixels are processed 4 per lane as aligned dwords; the 8-bit stencil math
//    runs on packed 16-bit lanes (v_perm_b32 + v_pk_*_u16).  No MFMA: nothing
//    here is a dense contraction.  Every kernel turned out to be bound by VALU
//    issue (one wave64 instruction per SIMD per 4 clocks), so the designs
//    minimise instructions per pixel first, memory traffic second.
//  * Row-coalesced aligned loads: straight into registers where a stencil can
//    stream (blur, pyramid), into an LDS tile with halo where it needs random
//    access (FAST ring gathers, BRIEF patches).
//  * All levels of all frames of a batch are covered by ONE launch per stage;
//    each workgroup reads its tile from a 64-byte OrbxTileDesc record.
//  * Keypoint order is deterministic (row-major), never atomics-ordered.
//  * Integer arithmetic wherever the reference's float arithmetic is exact, so
//    results are bit-identical to the CPU oracle; the float parts (angles,
//    rotation, Harris) use orbx_math.h and -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "orbx_internal.h"
#include "orbx_math.h"
#include "orbx_wave.h"

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "pattern_31.inc"
};

// ---------------------------------------------------------------------------
// helpers

typedef uint16_t __attribute__((aligned(1))) u16_unaligned;

// (level, tile_x, tile_y) of this workgroup
__device__ __forceinline__ void decode_tile(const OrbxTileMap& tm, int nlevels, int& l, int& tx, int& ty) {
  const int t = blockIdx.x;
  l = 0;
#pragma unroll 1
  for (int i = 1; i < nlevels; i++)
    if (t >= tm.begin[i]) l = i;
  const int local = t - tm.begin[l];
  ty = local / tm.tiles_x[l];
  tx = local - ty * tm.tiles_x[l];
}

// BORDER_REFLECT_101 for p in [-len+1, 2*len-2], clamped otherwise
// (src/cuda/GaussianBlur1D.cu:27-32)
__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - p - 2;
  p = p < 0 ? 0 : p;
  return p >= len ? len - 1 : p;
}

// ---------------------------------------------------------------------------
// 1. pyramid: level 0 copy + fixed-point bilinear resize of every level >= 1
//    straight from level 0 (src/orb.cpp:111-120).  (A first version with one
//    thread per 4 pixels was latency-bound: 416k short waves, each with two
//    dependent memory round trips, tap table -> pixel gathers; 206 us per batch.)
//    Here a wave owns 256 x 4 output pixels: the row index
//     is wave-uniform, so the y tap and the two source-row bases are scalar
//     loads / SGPR addresses; the four x taps of a lane are loaded once and
//     reused for four rows; all 32 pixel-pair gathers of a lane are issued
//     before the first is consumed.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
struct __attribute__((packed, aligned(1))) uint2_unaligned {
  uint32_t x, y;
  __device__ operator uint2() const { return make_uint2(x, y); }
};

// ROWS output rows of a lane's four pixels.  WIN8: levels with scale <= 2, where the four
// source pairs of a lane lie inside one 8-byte window (host-verified per level), so ONE
// unaligned 8-byte load per source row replaces four 2-byte gathers and v_perm picks each pair
// out of it (only 4 registers per output row are in flight, which is what lets a wave own 8
// rows); else four 2-byte pair gathers per source row.
template <int ROWS, bool WIN8>
__device__ __forceinline__ void pyr_rows(const uint8_t* __restrict__ src, int in_stride, int w0, int h0,
                                         const OrbxResizeTap* __restrict__ ytaps, int yb, int lh,
                                         const uint32_t (&ofs)[4], const uint32_t (&cc)[4],
                                         __amdgpu_buffer_rsrc_t rout, uint32_t voff_st, uint32_t vmask, int pitch) {
  constexpr int NW = WIN8 ? 2 : 4;  // dwords in flight per source row
  uint32_t q0[ROWS][NW], q1[ROWS][NW];
  int b0[ROWS], b1[ROWS];
  uint32_t sel[4];
  uint32_t base = 0;
  if (WIN8) {
    // the window start is clamped so that it never reads past the source row
    base = min(ofs[0], (uint32_t)(w0 - 8));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t sb = ofs[k] - base;               // 0..6
      sel[k] = 0x0c000c00u | ((sb + 1) << 16) | sb;  // (src[ofs], src[ofs+1]) as two u16 lanes
    }
  }
  // every gather of the wave is in flight before the first is used
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int y = min(yb + r, lh - 1);         // rows past the level repeat the last one (not stored)
    const OrbxResizeTap ty_ = ytaps[y];        // wave-uniform -> scalar load
    const int sy0 = min(max(ty_.ofs, 0), h0 - 1), sy1 = min(max(ty_.ofs + 1, 0), h0 - 1);
    const uint8_t* S0 = src + (size_t)sy0 * in_stride;
    const uint8_t* S1 = src + (size_t)sy1 * in_stride;
    b0[r] = ty_.c0;
    b1[r] = ty_.c1;
    if (WIN8) {
      const uint2 a = *reinterpret_cast<const uint2_unaligned*>(S0 + base);
      const uint2 b = *reinterpret_cast<const uint2_unaligned*>(S1 + base);
      q0[r][0] = a.x;
      q0[r][1] = a.y;
      q1[r][0] = b.x;
      q1[r][1] = b.y;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // ofs <= w0-2 always (host table): one unaligned 16-bit load fetches src[ofs], src[ofs+1]
        q0[r][k] = *reinterpret_cast<const u16_unaligned*>(S0 + ofs[k]);
        q1[r][k] = *reinterpret_cast<const u16_unaligned*>(S1 + ofs[k]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    uint32_t out = 0;
    const uint32_t bs0 = ((uint32_t)b0[r] << 12) & 0xffffffu, bs1 = ((uint32_t)b1[r] << 12) & 0xffffffu;  // b <= 2048
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t p0, p1;  // the pixel pair as two u16 lanes
      if (WIN8) {
        p0 = __builtin_amdgcn_perm(q0[r][1], q0[r][0], sel[k]);
        p1 = __builtin_amdgcn_perm(q1[r][1], q1[r][0], sel[k]);
      } else {
        p0 = __builtin_amdgcn_perm(q0[r][k], q0[r][k], 0x0c010c00u);
        p1 = __builtin_amdgcn_perm(q1[r][k], q1[r][k], 0x0c010c00u);
      }
      // horizontal pass: src[ofs] * c0 + src[ofs+1] * c1 is one v_dot2_u32_u16 of the pixel
      // pair with the tap's packed (c0, c1)
      const us2_t cw = __builtin_bit_cast(us2_t, cc[k]);
      const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, p0), cw, 0u, false);
      const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, p1), cw, 0u, false);
      // vertical pass: (b * (r >> 4)) >> 16 == ((r & ~15) * (b << 12)) >> 32 with both factors
      // below 2^24: one v_and + one full-rate v_mul_hi_u32_u24 per term; the sum is <= 1022
      const uint32_t t0 = (uint32_t)(((u64)(r0 & 0xfffff0u) * (u64)bs0) >> 32);
      const uint32_t t1 = (uint32_t)(((u64)(r1 & 0xfffff0u) * (u64)bs1) >> 32);
      out |= ((t0 + t1 + 2u) >> 2) << (8 * k);
    }
    const int y = yb + r;
    if (y < lh) __builtin_amdgcn_raw_buffer_store_b32(out & vmask, rout, voff_st, y * pitch, 0);
  }
}

__global__ __launch_bounds__(256) void k_pyramid2(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int frame_bytes, int w0,
                                                  int h0, const uint8_t* __restrict__ in, int in_stride,
                                                  size_t in_frame_stride, const OrbxResizeTap* __restrict__ taps,
                                                  uint8_t* __restrict__ pyr) {
  if ((int)blockIdx.x >= n_tiles) return;  // (grid.x is padded to an odd number: orbx_launch_pyramid2)
  const OrbxTileDesc d = tiles[blockIdx.x];  // one scalar load instead of decoding through the plan
  struct {
    int w, h, pitch, xtab_off, ytab_off, win8;
  } L = {d.w, d.h, d.pitch, d.u0, d.u1, d.u2};
  const int l = d.l, tx = d.tx, ty = d.ty;
  const int rpw = d.f;  // rows per wave of this level's tiles: 8 (level 0, 8-byte-window levels) or 4
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint8_t* src = in + (size_t)f * in_frame_stride;
  const int x = tx * 256 + lane * 4;
  const int yb = (ty * 4 + wave) * rpw;
  if (yb >= L.h) return;  // whole wave
  const int pitch = L.pitch, w = L.w;
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(pyr + (size_t)f * frame_bytes + d.img_off, 0, pitch * L.h, 0x00020000);
  const uint32_t voff_st = x < pitch ? (uint32_t)x : 0xffffffffu;
  const int nvalid = w - x;
  const uint32_t vmask = nvalid >= 4 ? 0xffffffffu : nvalid <= 0 ? 0u : ((1u << (8 * nvalid)) - 1u);

  if (l == 0) {  // level 0 = the input frame (src/orb.cpp:112), re-pitched
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int y = yb + r;
      if (r < rpw && y < L.h) {
        const uint8_t* row = src + (size_t)y * in_stride;
        uint32_t v = 0;
        if (nvalid >= 4 || (nvalid > 0 && y + 1 < L.h)) {
          // (a partial last dword of a row above the last one runs into the next row of the
          // same frame: readable, and vmask drops those bytes)
          v = *reinterpret_cast<const u32_unaligned*>(row + x);
        } else if (nvalid > 0) {  // last dword of the last row: never read past the frame
          for (int k = 0; k < nvalid; k++) v |= (uint32_t)row[x + k] << (8 * k);
        }
        __builtin_amdgcn_raw_buffer_store_b32(v & vmask, rout, voff_st, y * pitch, 0);
      }
    }
    return;
  }

  // x taps of this lane's four pixels (zero taps for lanes right of the image)
  uint32_t ofs[4] = {0, 0, 0, 0}, cc[4] = {0, 0, 0, 0};
  if (nvalid > 0) {
    const uint4* tp = reinterpret_cast<const uint4*>(taps + L.xtab_off + x);
    const uint4 t01 = tp[0], t23 = tp[1];
    ofs[0] = t01.x; ofs[1] = t01.z; ofs[2] = t23.x; ofs[3] = t23.z;
    cc[0] = t01.y; cc[1] = t01.w; cc[2] = t23.y; cc[3] = t23.w;
  }
  const OrbxResizeTap* ytaps = taps + L.ytab_off;
  if (L.win8) {
    if (rpw == 8)
      pyr_rows<8, true>(src, in_stride, w0, h0, ytaps, yb, L.h, ofs, cc, rout, voff_st, vmask, pitch);
    else
      pyr_rows<4, true>(src, in_stride, w0, h0, ytaps, yb, L.h, ofs, cc, rout, voff_st, vmask, pitch);
  } else {
    pyr_rows<4, false>(src, in_stride, w0, h0, ytaps, yb, L.h, ofs, cc, rout, voff_st, vmask, pitch);
  }
}

// ---------------------------------------------------------------------------
// 2. 5x5 Gaussian blur, REFLECT_101.
//    kind 0: separable [1 4 6 4 1]/16 twice then round-half-even
//            == rne(sum_ij w_i w_j p / 256)   (src/cuda/GaussianBlur1D.cu:34-163;
//            every float intermediate there is an exact dyadic, so integer
//            arithmetic reproduces it bit for bit)
//    kind 1: 5x5 /273 kernel, rne(S/273)      (src/cuda/GaussianBlur.cu:21-130;
//            S/273 is never within float error of a .5 tie because 273 is odd)
//    Levels below first_level are copied unchanged.
#define BLUR_SROWS (ORBX_BLUR_TH + 4)
#define BLUR_SPITCH 72  // bytes: x0-4 .. x0+67

__device__ __forceinline__ void blur_load_tile(uint8_t* s_src, const uint8_t* img, const OrbxLevel& L, int x0,
                                               int y0) {
  uint32_t* s32 = reinterpret_cast<uint32_t*>(s_src);
  for (int i = threadIdx.x; i < BLUR_SROWS * (BLUR_SPITCH / 4); i += 256) {
    const int row = i / (BLUR_SPITCH / 4), c = i - row * (BLUR_SPITCH / 4);
    const int gy = reflect101(y0 - 2 + row, L.h);
    const int gx = x0 - 4 + 4 * c;
    uint32_t v = 0;
    if (gx >= 0 && gx + 4 <= L.pitch) v = *reinterpret_cast<const uint32_t*>(img + (size_t)gy * L.pitch + gx);
    s32[i] = v;
  }
  __syncthreads();
  // horizontal REFLECT_101 fix-up of the two halo columns on image borders
  if (threadIdx.x < BLUR_SROWS) {
    uint8_t* row = s_src + threadIdx.x * BLUR_SPITCH;
    if (x0 == 0) {
      row[3] = row[5];  // x=-1 <- x=1
      row[2] = row[6];  // x=-2 <- x=2
    }
    const int rem = L.w - x0;  // valid columns in this tile
    if (rem <= ORBX_BLUR_TW + 1) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int x = L.w + k;  // needs reflect
        const int col = x - x0 + 4;
        if (col < BLUR_SPITCH) row[col] = row[(2 * L.w - x - 2) - x0 + 4];
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_blur(OrbxPlan plan, OrbxTileMap tm, const uint8_t* __restrict__ src,
                                              uint8_t* __restrict__ dst, int first_level, int kind) {
  __shared__ __attribute__((aligned(16))) uint8_t s_src[BLUR_SROWS * BLUR_SPITCH];
  __shared__ __attribute__((aligned(16))) uint16_t s_h[BLUR_SROWS * ORBX_BLUR_TW];
  int l, tx, ty;
  decode_tile(tm, plan.nlevels, l, tx, ty);
  const OrbxLevel& L = plan.L[l];
  const int f = blockIdx.y;
  const uint8_t* img = src + (size_t)f * plan.frame_bytes + L.img_off;
  uint8_t* out = dst + (size_t)f * plan.frame_bytes + L.img_off;
  const int x0 = tx * ORBX_BLUR_TW, y0 = ty * ORBX_BLUR_TH;
  const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;
  const int gy = y0 + r, gx = x0 + c4;

  if (l < first_level) {  // pass-through copy
    if (gy < L.h && gx < L.pitch)
      *reinterpret_cast<uint32_t*>(out + (size_t)gy * L.pitch + gx) =
          *reinterpret_cast<const uint32_t*>(img + (size_t)gy * L.pitch + gx);
    return;
  }
  blur_load_tile(s_src, img, L, x0, y0);

  uint32_t packed = 0;
  if (kind == 0) {
    // horizontal pass -> u16 (<= 4080)
    for (int i = threadIdx.x; i < BLUR_SROWS * ORBX_BLUR_TW; i += 256) {
      const int row = i >> 6, c = i & 63;
      const uint8_t* p = s_src + row * BLUR_SPITCH + c + 2;  // p[0] is x-2
      s_h[i] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint16_t* h = s_h + r * ORBX_BLUR_TW + c4 + k;  // h[0] is y-2
      const uint32_t S = h[0] + 4u * h[ORBX_BLUR_TW] + 6u * h[2 * ORBX_BLUR_TW] + 4u * h[3 * ORBX_BLUR_TW] +
                         h[4 * ORBX_BLUR_TW];
      const uint32_t v = (S + 127u + ((S >> 8) & 1u)) >> 8;  // round-half-even of S/256
      if (gx + k < L.w) packed |= v << (8 * k);
    }
  } else {
    const int kw[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t S = 0;
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) S += (uint32_t)kw[i * 5 + j] * s_src[(r + i) * BLUR_SPITCH + c4 + k + 2 + j];
      const uint32_t v = (2u * S + 273u) / 546u;  // nearest integer to S/273 (no ties exist)
      if (gx + k < L.w) packed |= v << (8 * k);
    }
  }
  if (gy < L.h && gx < L.pitch) *reinterpret_cast<uint32_t*>(out + (size_t)gy * L.pitch + gx) = packed;
}

// (2b. register-streaming separable blur: orbx_blur.hip)

// (3. FAST-n segment test + score + NMS: orbx_fast.hip)

// ---------------------------------------------------------------------------
// 4. (stage operators Fast()/NMS() only; the whole path uses the fused kernel below)
//    ordered compaction of the survivor mask: one workgroup per (level,
//    frame) walks the mask words in row-major order, a block-wide exclusive
//    scan of popcounts gives every set bit its row-major rank, the first
//    `cap` are written (src/orb_cpu.cpp:108-110 order and cap -- NOT the
//    atomicAdd order of src/cuda/NMS.cu:123).
__global__ __launch_bounds__(256) void k_compact(OrbxPlan plan, const u64* __restrict__ mask,
                                                 orbx_keypoint* __restrict__ cand,
                                                 int32_t* __restrict__ cand_count,
                                                 int32_t* __restrict__ cand_total, int need_total) {
  __shared__ int s_wsum[4];
  const int l = blockIdx.x, f = blockIdx.y;
  const OrbxLevel& L = plan.L[l];
  const u64* m = mask + (size_t)f * plan.mask_words + L.mask_off;
  orbx_keypoint* out = cand + (size_t)f * plan.cand_total + L.cand_off;
  const int nwords = L.h * L.mask_wpr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int w0 = 0; w0 < nwords; w0 += 256) {
    const int i = w0 + tid;
    u64 v = i < nwords ? m[i] : 0ull;
    const int c = __popcll(v);
    const int incl = wave_scan_incl(c);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = s_wsum[k];
      if (k < wave) woff += s;
      tot += s;
    }
    int pos = base + woff + incl - c;
    if (c && pos < L.cap) {
      const int y = i / L.mask_wpr, xw = i - y * L.mask_wpr;
      while (v && pos < L.cap) {
        const int b = __ffsll((long long)v) - 1;
        v &= v - 1;
        orbx_keypoint kp;
        kp.x = orbx_mask_x(L, xw, b);
        kp.y = y;
        out[pos++] = kp;
      }
    }
    base += tot;
    __syncthreads();
    // the cap is reached (block-uniform): later rows cannot contribute; `total`
    // is then only a lower bound, which the whole-path callers never read
    if (!need_total && base >= L.cap) break;
  }
  if (tid == 0) {
    cand_total[f * plan.nlevels + l] = base;
    cand_count[f * plan.nlevels + l] = base < L.cap ? base : L.cap;
  }
}

// ---------------------------------------------------------------------------
// 5. Harris response at the candidate keypoints (intent of
//    src/cuda/HarrisScore.cu:23-89, DESIGN.md "Harris"): float Sobel with
//    REFLECT_101, window x window Gaussian of Ix^2, IxIy, Iy^2 accumulated in
//    (i,j) row-major order, R = (AC - B^2) - (k*(A+C))*(A+C).
//    Evaluated only in the (window+2)^2 neighbourhood of each candidate
//    instead of 5 full-frame convolutions.
__device__ __forceinline__ float harris_at(const uint8_t* img, int w, int h, int pitch, int x, int y,
                                           const float* __restrict__ g, int K, float kk) {
  const int r = K / 2;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = 0; i < K; i++) {
    const int yy = reflect101(y - r + i, h);
    const uint8_t* r0 = img + (size_t)reflect101(yy - 1, h) * pitch;
    const uint8_t* r1 = img + (size_t)yy * pitch;
    const uint8_t* r2 = img + (size_t)reflect101(yy + 1, h) * pitch;
    for (int j = 0; j < K; j++) {
      const int xx = reflect101(x - r + j, w);
      const int xl = reflect101(xx - 1, w), xr = reflect101(xx + 1, w);
      const int p00 = r0[xl], p01 = r0[xx], p02 = r0[xr];
      const int p10 = r1[xl], p12 = r1[xr];
      const int p20 = r2[xl], p21 = r2[xx], p22 = r2[xr];
      const float gx = (float)((p02 + 2 * p12 + p22) - (p00 + 2 * p10 + p20));
      const float gy = (float)((p20 + 2 * p21 + p22) - (p00 + 2 * p01 + p02));
      const float wgt = g[i * K + j];
      a = __fadd_rn(a, __fmul_rn(__fmul_rn(gx, gx), wgt));
      c = __fadd_rn(c, __fmul_rn(__fmul_rn(gy, gy), wgt));
      b = __fadd_rn(b, __fmul_rn(__fmul_rn(gx, gy), wgt));
    }
  }
  const float det = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, b));
  const float trace = __fadd_rn(a, c);
  return __fsub_rn(det, __fmul_rn(__fmul_rn(kk, trace), trace));
}

// Fast path for window K <= 7 and keypoints whose (K+2)^2 pixel neighbourhood leaves
// the image by at most ONE row / column (FAST keypoints keep 3 px from the border,
// so with K = 7 that is every keypoint of the whole path; the reflected row is a
// different row pointer, the reflected column one v_perm per row): the
// neighbourhood is fetched once with 3 aligned dword loads per row and
// byte-aligned in registers (v_alignbyte), pixels are converted with
// v_cvt_f32_ubyteN, Sobel sums are shared between neighbouring taps.  Every
// intermediate is a small exact integer in float, and the weighted
// accumulation keeps the (i,j) order and the separate mul/mul/add roundings of
// harris_at, so the result is bit-identical to it.
template <int K>
__device__ __forceinline__ float harris_fast(const uint8_t* img, int w, int h, int pitch, int x, int y,
                                             const float* __restrict__ g, float kk) {
  constexpr int r = K / 2, P = K + 2;
  static_assert(P <= 9, "the third dword of a row supplies pixel 8 only");
  const int xs = x - r - 1;               // >= -1
  const int a0 = xs & ~3, off = xs - a0;  // xs == -1: a0 = -4, off = 3
  // BORDER_REFLECT_101 of the one column that may lie outside: pixel -1 is pixel 1
  // (byte 2 of the first aligned group), pixel w is pixel w-2 (byte 2 of the second)
  const uint32_t sel_l = xs < 0 ? 0x03020102u : 0x03020100u;
  const uint32_t sel_r = xs + P - 1 >= w ? 0x0c0c0c06u : 0x0c0c0c00u;
  // all (K+2) x 3 dword loads first (one memory latency), then the window is
  // consumed row by row with three rolling pixel rows / horizontal sums, which
  // keeps the live set around 100 VGPRs instead of 160+
  uint32_t raw[P][3];
#pragma unroll
  for (int i = 0; i < P; i++) {
    int yy = y - r - 1 + i;  // only the first / last row can be outside (by one)
    if (i == 0) yy = yy < 0 ? 1 : yy;
    if (i == P - 1) yy = yy >= h ? h - 2 : yy;
    const int o = yy * pitch + a0;
    // o == -4 only in row 0 with the left column reflected; that dword supplies nothing
    // but the replaced pixel, so any readable address will do
    raw[i][0] = *reinterpret_cast<const uint32_t*>(img + max(o, 0));
    raw[i][1] = *reinterpret_cast<const uint32_t*>(img + (o + 4));
    raw[i][2] = *reinterpret_cast<const uint32_t*>(img + (o + 8));  // may run into the next row / the pool's tail slack
  }
  auto cvt_row = [&](int i, float (&out)[P]) {
    uint32_t q[3] = {__builtin_amdgcn_alignbyte(raw[i][1], raw[i][0], off),
                     __builtin_amdgcn_alignbyte(raw[i][2], raw[i][1], off), raw[i][2] >> (8 * off)};
    q[2] = __builtin_amdgcn_perm(q[1], q[2], sel_r);
    q[0] = __builtin_amdgcn_perm(q[0], q[0], sel_l);
#pragma unroll
    for (int j = 0; j < P; j++) out[j] = (float)((q[j >> 2] >> (8 * (j & 3))) & 0xffu);
  };
  auto hsum = [&](const float (&pr)[P], float (&out)[K]) {
#pragma unroll
    for (int j = 0; j < K; j++) out[j] = pr[j] + 2.0f * pr[j + 1] + pr[j + 2];
  };
  float pr[3][P], hs[3][K];
  cvt_row(0, pr[0]);
  cvt_row(1, pr[1]);
  hsum(pr[0], hs[0]);
  hsum(pr[1], hs[1]);
  float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
  for (int i = 0; i < K; i++) {
    // rows i, i+1, i+2 of the patch live in slots i%3, (i+1)%3, (i+2)%3
    float(&p0)[P] = pr[i % 3];
    float(&p1)[P] = pr[(i + 1) % 3];
    float(&p2)[P] = pr[(i + 2) % 3];
    cvt_row(i + 2, p2);
    hsum(p2, hs[(i + 2) % 3]);
    float vs[P];
#pragma unroll
    for (int j = 0; j < P; j++) vs[j] = p0[j] + 2.0f * p1[j] + p2[j];
#pragma unroll
    for (int j = 0; j < K; j++) {
      const float gx = vs[j + 2] - vs[j];
      const float gy = hs[(i + 2) % 3][j] - hs[i % 3][j];
      const float wgt = g[i * K + j];
      a = __fadd_rn(a, __fmul_rn(__fmul_rn(gx, gx), wgt));
      c = __fadd_rn(c, __fmul_rn(__fmul_rn(gy, gy), wgt));
      b = __fadd_rn(b, __fmul_rn(__fmul_rn(gx, gy), wgt));
    }
  }
  const float det = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, b));
  const float trace = __fadd_rn(a, c);
  return __fsub_rn(det, __fmul_rn(__fmul_rn(kk, trace), trace));
}

__device__ __forceinline__ float harris_any(const uint8_t* img, int w, int h, int pitch, int x, int y,
                                            const float* __restrict__ g, int K, float kk) {
  const int m = K / 2 + 1;  // the fast path takes (K+2)^2 windows that leave the image by at most one pixel
  if (x >= m - 1 && y >= m - 1 && x <= w - m && y <= h - m && w >= 4 && h >= 4) {
    if (K == 7) return harris_fast<7>(img, w, h, pitch, x, y, g, kk);
    if (K == 5) return harris_fast<5>(img, w, h, pitch, x, y, g, kk);
    if (K == 3) return harris_fast<3>(img, w, h, pitch, x, y, g, kk);
  }
  return harris_at(img, w, h, pitch, x, y, g, K, kk);
}

// Eight lanes per keypoint (second generation): lane r of a group computes window
// row r -- three patch rows, the Sobel sums and the 3 x K weighted products of that
// row -- in parallel with its neighbours; the float accumulators then travel
// lane 0 -> 1 -> ... -> K-1 (DPP row_shr:1), each lane adding its K products in
// order, which reproduces the oracle's (i,j) row-major summation bit for bit.
// 8x more waves than thread-per-keypoint (the old kernel had 2 waves per SIMD)
// and a ~6x shorter dependent chain per keypoint.
// RPL window rows per lane, G = lanes per keypoint (power of two >= ceil(K/RPL)).
template <int K, int RPL, int G>
__device__ __forceinline__ float harris_row_group(const uint8_t* img, int pitch, int x, int y,
                                                  const float* __restrict__ g, float kk, int sub) {
  constexpr int r = K / 2, P = K + 2, NR = RPL + 2;  // NR patch rows feed RPL window rows
  constexpr int NL = (K + RPL - 1) / RPL;            // lanes that own rows
  const int row0 = (sub < NL ? sub : NL - 1) * RPL;  // idle lanes shadow the last group (sums never used)
  const int xs = x - r - 1;
  const int a0 = xs & ~3, off = xs - a0;
  float p[NR][P];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    // rows past the window (last lane when K is not a multiple of RPL) repeat the last patch row
    const int pr = min(row0 + i, P - 1);
    const uint32_t* row = reinterpret_cast<const uint32_t*>(img + (size_t)(y - r - 1 + pr) * pitch + a0);
    const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];
    const uint32_t q[3] = {__builtin_amdgcn_alignbyte(d1, d0, off), __builtin_amdgcn_alignbyte(d2, d1, off),
                           d2 >> (8 * off)};
#pragma unroll
    for (int j = 0; j < P; j++) p[i][j] = (float)((q[j >> 2] >> (8 * (j & 3))) & 0xffu);
  }
  float hs[NR][K];
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < K; j++) hs[i][j] = p[i][j] + 2.0f * p[i][j + 1] + p[i][j + 2];
  float pa[RPL][K], pb[RPL][K], pc[RPL][K];
#pragma unroll
  for (int w = 0; w < RPL; w++) {
    float vs[P];
#pragma unroll
    for (int j = 0; j < P; j++) vs[j] = p[w][j] + 2.0f * p[w + 1][j] + p[w + 2][j];
    const int wr = min(row0 + w, K - 1);
#pragma unroll
    for (int j = 0; j < K; j++) {
      const float gx = vs[j + 2] - vs[j], gy = hs[w + 2][j] - hs[w][j];
      const float wgt = g[wr * K + j];
      pa[w][j] = __fmul_rn(__fmul_rn(gx, gx), wgt);
      pc[w][j] = __fmul_rn(__fmul_rn(gy, gy), wgt);
      pb[w][j] = __fmul_rn(__fmul_rn(gx, gy), wgt);
    }
  }
  float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
  for (int rr = 0; rr < NL; rr++) {
    if (sub == rr) {
#pragma unroll
      for (int w = 0; w < RPL; w++)
        if (rr * RPL + w < K) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            a = __fadd_rn(a, pa[w][j]);
            c = __fadd_rn(c, pc[w][j]);
            b = __fadd_rn(b, pb[w][j]);
          }
        }
    }
    if (rr + 1 < NL) {  // hand the running sums to the next lane of the group
      const float an = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xf, 0xf, true));
      const float bn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x111, 0xf, 0xf, true));
      const float cn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x111, 0xf, 0xf, true));
      if (sub == rr + 1) {
        a = an;
        b = bn;
        c = cn;
      }
    }
  }
  const float det = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, b));
  const float trace = __fadd_rn(a, c);
  return __fsub_rn(det, __fmul_rn(__fmul_rn(kk, trace), trace));  // valid in lane sub == NL-1
}

// one keypoint per G-lane group; the result is valid in the lane `writer` says
template <int RPL, int G>
__device__ __forceinline__ float harris_group(const uint8_t* img, int w, int h, int pitch, int x, int y,
                                              const float* __restrict__ g, int K, float kk, int sub, bool active,
                                              bool& writer) {
  const int m = K / 2 + 1;
  const bool fast = (K == 7 || K == 5 || K == 3) && x >= m && y >= m && x < w - m && y < h - m;
  float res = 0.f;
  writer = false;
  // the fast path is taken group-wise but executed wave-wide (DPP needs all lanes of the row active)
  const bool any_fast = __any(active && fast);
  if (any_fast) {
    const int xx = (active && fast) ? x : m, yy = (active && fast) ? y : m;  // harmless in-image stand-in
    float v = 0.f;
    int last = 0;
    if (K == 7) {
      v = harris_row_group<7, RPL, G>(img, pitch, xx, yy, g, kk, sub);
      last = (7 + RPL - 1) / RPL - 1;
    } else if (K == 5) {
      v = harris_row_group<5, RPL, G>(img, pitch, xx, yy, g, kk, sub);
      last = (5 + RPL - 1) / RPL - 1;
    } else {
      v = harris_row_group<3, RPL, G>(img, pitch, xx, yy, g, kk, sub);
      last = (3 + RPL - 1) / RPL - 1;
    }
    if (active && fast && sub == last) {
      res = v;
      writer = true;
    }
  }
  if (active && !fast && sub == 0) {
    res = harris_at(img, w, h, pitch, x, y, g, K, kk);
    writer = true;
  }
  return res;
}

__global__ __launch_bounds__(256) void k_harris2_flat(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                      const orbx_keypoint* __restrict__ kps, int nkp,
                                                      const float* __restrict__ gauss, int K, float kk,
                                                      float* __restrict__ resp) {
  constexpr int G = 8, RPL = 1;
  const int j = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  const bool active = j < nkp;
  orbx_keypoint kp = {4, 4};
  if (active) kp = kps[j];
  bool writer;
  const float v = harris_group<RPL, G>(img, w, h, pitch, kp.x, kp.y, gauss, K, kk, sub, active, writer);
  if (writer) resp[j] = v;
}


// ---------------------------------------------------------------------------
// 4-6 fused: ordered compaction -> Harris -> top-N selection, one 512-thread
// workgroup per (level, frame).  The three separate kernels were each bound by
// launch + latency (512..2048 short workgroups, 10 + 35 + 22 us per batch); one
// workgroup now carries its level's candidates from the survivor mask to the
// selected list through LDS, with no intermediate global traffic and two
// launches fewer.
//   phase 1  row-major walk of the mask words (block scan of popcounts), first
//            `cap` survivors -> LDS (src/orb_cpu.cpp:108-110 order and cap)
//   phase 2  Harris response per candidate (thread per candidate, harris_any)
//   phase 3  rank by the 64-bit (response desc, index asc) key, keep `quota`
// The selected keypoints of level l go to the STATIC slots [out_off_l, out_off_l
// + quota_l) of the frame (out_off_l = sum of the lower levels' quotas) plus a
// per-level count; k_describe2 compacts them into the final order.
#define LVL_THREADS 512
__global__ __launch_bounds__(LVL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_level_select(OrbxPlan plan, int mode, const u64* __restrict__ mask,
                                                              const uint8_t* __restrict__ pyr,
                                                              const float* __restrict__ gauss, int K, float kk,
                                                              orbx_keypoint* __restrict__ sel_lkp,
                                                              float* __restrict__ sel_resp,
                                                              int32_t* __restrict__ sel_count,
                                                              uint32_t* __restrict__ need) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ int s_wsum[4][LVL_THREADS / 64];
  // grid = (frames, levels), the frame index dispatched fastest.  Workgroups are dealt round-robin over
  // the 8 XCDs in linear order: with (levels, frames) and 8 levels every XCD got ONE level of every
  // frame, level 0 (7x the work of level 7) all on one XCD, and the launch took as long as that XCD
  // needed (84 us per 256 frames; XCD placement is a matter of speed only).
  const int f = blockIdx.x, l = blockIdx.y;
  const OrbxLevel& L = plan.L[l];
  const int cap = L.cap, cap2 = (cap + 1) & ~1;
  u64* s_key = reinterpret_cast<u64*>(s_dyn);                         // [cap2]
  uint32_t* s_kp = reinterpret_cast<uint32_t*>(s_dyn + 8 * (size_t)cap2);  // [cap]  y << 16 | x
  float* s_r = reinterpret_cast<float*>(s_dyn + 12 * (size_t)cap2);   // [cap]
  const u64* m = mask + (size_t)f * plan.mask_words + L.mask_off;
  const int nwords = L.h * L.mask_wpr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // phase 1: ordered compaction.  Four chunks of LVL_THREADS words per round: their loads are in
  // flight together (one memory round trip per 2048 words; level 0 usually needs one round)
  int base = 0;
  for (int w0 = 0; w0 < nwords && base < cap; w0 += 4 * LVL_THREADS) {
    u64 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = w0 + LVL_THREADS * q + tid;
      v[q] = i < nwords ? m[i] : 0ull;
    }
    int incl[4], c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      c[q] = __popcll(v[q]);
      incl[q] = wave_scan_incl(c[q]);
      if (lane == 63) s_wsum[q][wave] = incl[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int woff = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < LVL_THREADS / 64; k++) {
        const int sv = s_wsum[q][k];
        if (k < wave) woff += sv;
        tot += sv;
      }
      int pos = base + woff + incl[q] - c[q];
      if (c[q] && pos < cap) {
        const int i = w0 + LVL_THREADS * q + tid;
        const int y = i / L.mask_wpr, xw = i - y * L.mask_wpr;
        u64 w = v[q];
        while (w && pos < cap) {
          const int b = __ffsll((long long)w) - 1;
          w &= w - 1;
          s_kp[pos++] = ((uint32_t)y << 16) | (uint32_t)orbx_mask_x(L, xw, b);
        }
      }
      base += tot;
    }
    __syncthreads();
  }
  const int n = base < cap ? base : cap;
  const int keep = n < L.quota ? n : L.quota;
  const size_t so = (size_t)f * plan.out_cap + L.out_off;
  if (tid == 0) sel_count[f * plan.nlevels + l] = keep;
  // How many rows of the level did it take to fill the cap (all of them if it never filled)?  The maximum over the
  // batch's frames goes to the host, which sizes the first pass of the top-rows-first pipeline by it (orbx_api.cpp,
  // adapt_tile_rows): a heuristic's input, results never depend on it.
  if (need && tid == 0 && cap > 0) atomicMax(&need[l], base >= cap ? (s_kp[cap - 1] >> 16) + 1u : (uint32_t)L.h);

  if (mode == ORBX_SELECT_ROWMAJOR) {
    for (int i = tid; i < keep; i += LVL_THREADS) {
      const uint32_t p = s_kp[i];
      orbx_keypoint kp;
      kp.x = (int)(p & 0xffffu);
      kp.y = (int)(p >> 16);
      sel_lkp[so + i] = kp;
      sel_resp[so + i] = 0.0f;
    }
    return;
  }

  // phase 2: Harris responses -> (response desc, index asc) keys
  const uint8_t* img = pyr + (size_t)f * plan.frame_bytes + L.img_off;
  const int n2 = (n + 1) & ~1;
  for (int i = tid; i < n2; i += LVL_THREADS) {
    u64 key = 0ull;  // padding: the smallest key, never outranks anything
    if (i < n) {
      const uint32_t p = s_kp[i];
      const float r = harris_any(img, L.w, L.h, L.pitch, (int)(p & 0xffffu), (int)(p >> 16), gauss, K, kk);
      s_r[i] = r;
      uint32_t u = orbx_f2u(r);
      if (u == 0x80000000u) u = 0u;
      u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
      key = ((u64)u << 32) | (uint32_t)~(uint32_t)i;
    }
    s_key[i] = key;
  }
  __syncthreads();

  // phase 3: rank and scatter
  for (int i = tid; i < n; i += LVL_THREADS) {
    const u64 ki = s_key[i];
    int rank = 0;
#pragma unroll 4
    for (int j = 0; j < n2; j += 2) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&s_key[j]);
      rank += v.x > ki;
      rank += v.y > ki;
    }
    if (rank < keep) {
      const uint32_t p = s_kp[i];
      orbx_keypoint kp;
      kp.x = (int)(p & 0xffffu);
      kp.y = (int)(p >> 16);
      sel_lkp[so + rank] = kp;
      sel_resp[so + rank] = s_r[i];
    }
  }
}

// ---------------------------------------------------------------------------
// 6c. The same selection as k_level_select, spread over the whole chip.  k_level_select does
//     everything for one (level, frame) in ONE workgroup, which is the faster arrangement while
//     a level has at most one candidate per thread (KITTI, 1000 features: 27 us vs 31 us for the
//     three kernels below at batch 64, 119 vs 134 us for a single frame end to end).  With larger
//     per-level caps (1920x1080, 4000 features: level 0 has ~1400 candidates) it serialises
//     Harris and the n^2 ranking inside single workgroups (110 us at batch 16); then three short
//     kernels take over (79 us; orbx_launch_level_select_auto):
//       k_lvl_compact  (level, frame)                   first `cap` survivors, row-major
//       k_lvl_harris   (256 candidates, level, frame)   thread per candidate
//       k_lvl_rank     (64 candidates, level, frame)    4 threads per candidate
//     Same arithmetic, same order, same tie rule: results are identical.
__global__ __launch_bounds__(256) void k_lvl_compact(OrbxPlan plan, const u64* __restrict__ mask,
                                                     uint32_t* __restrict__ cand, int32_t* __restrict__ ncand,
                                                     uint32_t* __restrict__ need) {
  __shared__ int s_wsum[4][4];
  const int l = blockIdx.x, f = blockIdx.y;
  const OrbxLevel& L = plan.L[l];
  const int cap = L.cap;
  const u64* m = mask + (size_t)f * plan.mask_words + L.mask_off;
  uint32_t* out = cand + (size_t)f * plan.cand_total + L.cand_off;
  const int nwords = L.h * L.mask_wpr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  // four 256-word chunks per round: their loads are in flight together (one memory
  // round trip per 1024 words); words after the first `cap` survivors are never needed
  for (int w0 = 0; w0 < nwords && base < cap; w0 += 1024) {
    u64 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = w0 + 256 * q + tid;
      v[q] = i < nwords ? m[i] : 0ull;
    }
    int incl[4], c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      c[q] = __popcll(v[q]);
      incl[q] = wave_scan_incl(c[q]);
      if (lane == 63) s_wsum[q][wave] = incl[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int woff = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int sv = s_wsum[q][k];
        if (k < wave) woff += sv;
        tot += sv;
      }
      int pos = base + woff + incl[q] - c[q];
      if (c[q] && pos < cap) {
        const int i = w0 + 256 * q + tid;
        const int y = i / L.mask_wpr, xw = i - y * L.mask_wpr;
        u64 w = v[q];
        while (w && pos < cap) {
          const int b = __ffsll((long long)w) - 1;
          w &= w - 1;
          out[pos++] = ((uint32_t)y << 16) | (uint32_t)orbx_mask_x(L, xw, b);
        }
      }
      base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) ncand[f * plan.nlevels + l] = base < cap ? base : cap;
  // rows the level needed to fill its cap, for the adaptive first pass (see k_level_select; out[] was written by
  // this workgroup before the loop's last barrier)
  if (need && tid == 0 && cap > 0) atomicMax(&need[l], base >= cap ? (out[cap - 1] >> 16) + 1u : (uint32_t)L.h);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_lvl_harris(
    OrbxPlan plan, const uint8_t* __restrict__ pyr, const float* __restrict__ gauss, int K, float kk,
    const uint32_t* __restrict__ cand, const int32_t* __restrict__ ncand, float* __restrict__ resp) {
  const int l = blockIdx.y, f = blockIdx.z;
  const OrbxLevel& L = plan.L[l];
  const int n = ncand[f * plan.nlevels + l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.x * 256 >= n) return;  // whole workgroup
  if (i >= n) return;
  const size_t o = (size_t)f * plan.cand_total + L.cand_off + i;
  const uint32_t p = cand[o];
  const uint8_t* img = pyr + (size_t)f * plan.frame_bytes + L.img_off;
  resp[o] = harris_any(img, L.w, L.h, L.pitch, (int)(p & 0xffffu), (int)(p >> 16), gauss, K, kk);
}

#define RANK_SLICE 64
__global__ __launch_bounds__(256) void k_lvl_rank(OrbxPlan plan, const uint32_t* __restrict__ cand,
                                                  const int32_t* __restrict__ ncand, const float* __restrict__ resp,
                                                  orbx_keypoint* __restrict__ sel_lkp, float* __restrict__ sel_resp,
                                                  int32_t* __restrict__ sel_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ int s_part[4][RANK_SLICE];
  u64* s_key = reinterpret_cast<u64*>(s_dyn);  // [n rounded up to 8]
  const int l = blockIdx.y, f = blockIdx.z;
  const OrbxLevel& L = plan.L[l];
  const int n = ncand[f * plan.nlevels + l];
  const int keep = n < L.quota ? n : L.quota;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) sel_count[f * plan.nlevels + l] = keep;
  const int first = blockIdx.x * RANK_SLICE;
  if (first >= n) return;  // whole workgroup
  const size_t co = (size_t)f * plan.cand_total + L.cand_off;
  // all keys of the (level, frame): (response desc, index asc) as one u64, like k_level_select
  const int n8 = (n + 7) & ~7;
  for (int i = tid; i < n8; i += 256) {
    u64 key = 0ull;  // padding: the smallest key, never outranks anything
    if (i < n) {
      uint32_t u = orbx_f2u(resp[co + i]);
      if (u == 0x80000000u) u = 0u;
      u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
      key = ((u64)u << 32) | (uint32_t)~(uint32_t)i;
    }
    s_key[i] = key;
  }
  __syncthreads();
  // thread (ci, part): candidate first + ci against quarter `part` of the keys
  const int ci = tid & (RANK_SLICE - 1), part = tid >> 6;
  const int i = first + ci;
  const u64 ki = s_key[i < n8 ? i : 0];
  const int per = n8 / 4;  // multiple of 2
  int rank = 0;
  for (int j = part * per; j < (part + 1) * per; j += 2) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&s_key[j]);
    rank += v.x > ki;
    rank += v.y > ki;
  }
  s_part[part][ci] = rank;
  __syncthreads();
  if (part == 0 && i < n) {
    rank = s_part[0][ci] + s_part[1][ci] + s_part[2][ci] + s_part[3][ci];
    if (rank < keep) {
      const uint32_t p = cand[co + i];
      const size_t so = (size_t)f * plan.out_cap + L.out_off;
      orbx_keypoint kp;
      kp.x = (int)(p & 0xffffu);
      kp.y = (int)(p >> 16);
      sel_lkp[so + rank] = kp;
      sel_resp[so + rank] = resp[co + i];
    }
  }
}

__global__ __launch_bounds__(256) void k_select_flat(const float* __restrict__ resp, int n, int keep,
                                                     int32_t* __restrict__ idx) {
  const int m = n < keep ? n : keep;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float ri = resp[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const float rj = resp[j];
      rank += (rj > ri) || (rj == ri && j < i);
    }
    if (rank < m) idx[rank] = i;
  }
}

// ---------------------------------------------------------------------------
// 7. (stage operators Orientations()/Brief() only; the whole path uses 7b)
//    orientation + rotated BRIEF-256, one wavefront per keypoint
//    (src/orb_cpu.cpp:139-258; src/cuda/Orientations.cu:22-63,
//    src/cuda/Brief.cu:40-95).
//    The 41x41 neighbourhood (pattern radius 18 + 5x5 box radius 2, and the
//    orientation patch) is loaded once into LDS with aligned dword loads,
//    zero-filled outside the image.  Moments are exact int32 reduced across
//    the wave; each lane then evaluates 4 of the 256 tests with 5x5 box sums
//    taken straight from the u8 patch (exact integers, no integral image),
//    and 4 ballots assemble the descriptor.
#define DESC_R 20
#define DESC_ROWS (2 * DESC_R + 1)  // 41
#define DESC_PITCH 48               // bytes per LDS patch row (12 dwords)
#define DESC_HP 40                  // u16 entries per row of the box-sum tables
#define DESC_BROWS (DESC_ROWS - 4)  // 37 rows of 5x5 box sums

// per-wavefront LDS working set
struct DescLds {
  // the patch is dead once the horizontal sums exist, so the 5x5 table reuses its
  // space (6.2 KB per wave -> 6 workgroups per CU instead of 4)
  union {
    uint32_t patch[DESC_ROWS * DESC_PITCH / 4];  // 41 x 48 u8
    uint16_t box[DESC_BROWS * DESC_HP];          // 5x5 sums: box[r][j] = sum hs[r..r+4][j]
  };
  uint16_t hs[DESC_ROWS * DESC_HP];  // horizontal 5-sums: hs[r][j] = sum patch[r][j..j+4]
};

// the four 5-byte sums b[k..k+4], k = 0..3, of the 8 bytes (d0 low), as packed u16.
// v_qsad_pk_u16_u8 against 0 gives the four sliding 4-byte sums b[k..k+3] plus a packed
// accumulator, which carries b[k+4]: one instruction instead of ~20 byte extracts and adds.
__device__ __forceinline__ u64 hsum5x4(uint32_t d0, uint32_t d1) {
  const u64 acc = (u64)__builtin_amdgcn_perm(d1, d1, 0x0c010c00u) |
                  ((u64)__builtin_amdgcn_perm(d1, d1, 0x0c030c02u) << 32);
  return __builtin_amdgcn_qsad_pk_u16_u8((u64)d0 | ((u64)d1 << 32), 0u, acc);
}

struct DescJob {
  const uint8_t* img;
  int w, h, pitch;
  int x, y;
  bool valid;
};

// The 512 box sums a descriptor needs are looked up in a per-keypoint table of
// ALL 5x5 box sums of the 41x41 neighbourhood, built with two separable passes
// of wide, conflict-free LDS accesses (2 dword reads -> 4 sums -> one 8-byte
// write; 5 8-byte reads -> one 8-byte write) instead of 25 scattered byte reads
// per sample.  Sums are exact integers (<= 6375), identical to what the
// reference derives from its integral image (src/orb_cpu.cpp:190-201).
__device__ __forceinline__ void describe_wave(const DescJob& jb, DescLds& lds, int patch_size,
                                              bool use_given_angle, float given_angle, bool do_brief,
                                              float& angle_out, u64 desc_out[4]) {
  const int lane = lane_id();
  const uint8_t* s_patch = reinterpret_cast<const uint8_t*>(lds.patch);
  const int px0 = jb.x - DESC_R, py0 = jb.y - DESC_R;
  const int ax0 = px0 & ~3;  // floor to a multiple of 4 (two's complement)
  const int off = px0 - ax0;
  if (jb.valid) {
    // item i = lane + 64k -> (row, dword column); 64 = 5 * 12 + 4 gives the increments
    int row = lane / (DESC_PITCH / 4), c = lane - row * (DESC_PITCH / 4);
#pragma unroll
    for (int k = 0; k < (DESC_ROWS * (DESC_PITCH / 4) + 63) / 64; k++) {
      const int i = lane + 64 * k;
      if (i < DESC_ROWS * (DESC_PITCH / 4)) {
        const int gy = py0 + row, gx = ax0 + 4 * c;
        uint32_t v = 0;
        // gx, pitch are multiples of 4: (unsigned)gx < pitch  <=>  0 <= gx && gx + 4 <= pitch
        if ((unsigned)gy < (unsigned)jb.h && (unsigned)gx < (unsigned)jb.pitch)
          v = *reinterpret_cast<const uint32_t*>(jb.img + (uint32_t)(gy * jb.pitch + gx));
        lds.patch[i] = v;
      }
      row += 5;
      c += 4;
      if (c >= DESC_PITCH / 4) {
        c -= DESC_PITCH / 4;
        row += 1;
      }
    }
  }
  __syncthreads();
  float angle = 0.0f;
  if (jb.valid) {
    if (use_given_angle) {
      angle = given_angle;
    } else {
      const int pr = patch_size / 2, P = 2 * pr + 1;
      // full patch must lie inside the image, else 0 (src/orb_cpu.cpp:152-156)
      if (!(jb.x - pr < 0 || jb.x + pr >= jb.w || jb.y - pr < 0 || jb.y + pr >= jb.h)) {
        // lanes = patch columns (two row groups when the patch is <= 32 wide),
        // rows in the loop: no per-pixel division, one LDS byte read per step
        const int grp_shift = P <= 32 ? 5 : 6;
        const int col = lane & ((1 << grp_shift) - 1), grp = lane >> grp_shift, ngrp = 64 >> grp_shift;
        int colsum = 0, m01 = 0;
        if (col < P) {
          const uint8_t* pc = s_patch + (DESC_R - pr) * DESC_PITCH + (col - pr + DESC_R + off);
          for (int rr = grp; rr < P; rr += ngrp) {
            const int I = pc[rr * DESC_PITCH];
            colsum += I;
            m01 += (rr - pr) * I;
          }
        }
        int m10 = (col - pr) * colsum;
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
        // the reference accumulates in float; every partial sum is an integer
        // below 2^24, hence exact, hence equal to this int32 sum
        angle = orbx_atan2f((float)m01, (float)m10);
      }
    }
  }
  angle_out = angle;
  if (!do_brief) return;

  // pass 1: horizontal 5-sums, 4 per item (row r, dword group g); item i = lane + 64k,
  // 64 = 6 * 10 + 4 gives the increments
  if (jb.valid) {
    int r = lane / 10, g = lane - r * 10;
#pragma unroll
    for (int k = 0; k < (DESC_ROWS * 10 + 63) / 64; k++) {
      if (lane + 64 * k < DESC_ROWS * 10) {
        const uint32_t d0 = lds.patch[r * (DESC_PITCH / 4) + g], d1 = lds.patch[r * (DESC_PITCH / 4) + g + 1];
        *reinterpret_cast<u64*>(&lds.hs[r * DESC_HP + 4 * g]) = hsum5x4(d0, d1);
      }
      r += 6;
      g += 4;
      if (g >= 10) {
        g -= 10;
        r += 1;
      }
    }
  }
  __syncthreads();
  // pass 2: vertical 5-sums of the horizontal sums
  if (jb.valid) {
    int r = lane / 10, g = lane - r * 10;
#pragma unroll
    for (int k = 0; k < (DESC_BROWS * 10 + 63) / 64; k++) {
      if (lane + 64 * k < DESC_BROWS * 10) {
        const uint2* hp = reinterpret_cast<const uint2*>(&lds.hs[r * DESC_HP + 4 * g]);
        uint2 acc = hp[0];
#pragma unroll
        for (int q = 1; q < 5; q++) {
          const uint2 v = hp[q * (DESC_HP / 4)];
          acc.x = pk_add(acc.x, v.x);
          acc.y = pk_add(acc.y, v.y);
        }
        *reinterpret_cast<uint2*>(&lds.box[r * DESC_HP + 4 * g]) = acc;
      }
      r += 6;
      g += 4;
      if (g >= 10) {
        g -= 10;
        r += 1;
      }
    }
  }
  __syncthreads();

  const float c = orbx_cosf(angle), s = orbx_sinf(angle);
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    bool bit = false;
    if (jb.valid) {
      const int i = k * 64 + lane;
      const int32_t pk = reinterpret_cast<const int32_t*>(c_pattern)[i];
      const float x1 = (float)(int8_t)(pk & 0xff), y1 = (float)(int8_t)((pk >> 8) & 0xff);
      const float x2 = (float)(int8_t)((pk >> 16) & 0xff), y2 = (float)(int8_t)((pk >> 24) & 0xff);
      // lround(c*x - s*y), lround(s*x + c*y): separate IEEE mul / add-sub,
      // round half away from zero (src/orb_cpu.cpp:228-232)
      int dx1 = orbx_lroundf(__fsub_rn(__fmul_rn(c, x1), __fmul_rn(s, y1)));
      int dy1 = orbx_lroundf(__fadd_rn(__fmul_rn(s, x1), __fmul_rn(c, y1)));
      int dx2 = orbx_lroundf(__fsub_rn(__fmul_rn(c, x2), __fmul_rn(s, y2)));
      int dy2 = orbx_lroundf(__fadd_rn(__fmul_rn(s, x2), __fmul_rn(c, y2)));
      const int cx1 = jb.x + dx1, cy1 = jb.y + dy1, cx2 = jb.x + dx2, cy2 = jb.y + dy2;
      // bounds check of the reference, in image terms: integral width-2 ==
      // cols-1 (src/orb_cpu.cpp:240-245)
      const bool skip = cx1 < 2 || cy1 < 2 || cx1 > jb.w - 1 || cy1 > jb.h - 1 || cx2 < 2 || cy2 < 2 ||
                        cx2 > jb.w - 1 || cy2 > jb.h - 1;
      if (!skip) {
        // memory safety for angles outside the documented domain
        dx1 = min(max(dx1, -18), 18);
        dy1 = min(max(dy1, -18), 18);
        dx2 = min(max(dx2, -18), 18);
        dy2 = min(max(dy2, -18), 18);
        const int s1 = lds.box[(dy1 + 18) * DESC_HP + dx1 + 18 + off];
        const int s2 = lds.box[(dy2 + 18) * DESC_HP + dx2 + 18 + off];
        bit = s1 < s2;
      }
    }
    desc_out[k] = __ballot(bit);
  }
}

// ---------------------------------------------------------------------------
// 7b. orientation + BRIEF, second generation: a workgroup owns 16 keypoints.
//   The per-keypoint transcendental work (restated atan2f, sinf, cosf: ~350
//   instructions, identical in every lane of a wave-per-keypoint design) is
//   done by 16 THREADS, one per keypoint, between two passes in which each
//   wave serves its four keypoints:
//     pass A  patch -> LDS, exact int32 moments (wave reduce)
//     trig    thread j: angle_j, cos_j, sin_j
//     pass C  patch -> LDS again (L1/L2 hit), separable 5x5 box table,
//             256 rotated tests, 4 ballots
//   Keypoints whose whole 41x41 neighbourhood is inside the image (wave-uniform
//   test) skip the per-test bounds checks.
// KPW keypoints per wave (template parameter of k_describe2): 4 for throughput (the trig runs in 16 of a
// workgroup's threads), 1 for the few-keypoint / single-frame case where the chip is nearly empty and the
// latency of four descriptor passes in a row is what counts


#define DESC_NLD ((DESC_ROWS * (DESC_PITCH / 4) + 63) / 64)  // dwords per lane per patch (8)

// global -> registers (all loads of a patch issued back to back; the caller
// fetches the patches of ALL its keypoints before using the first one)
__device__ __forceinline__ void desc_fetch_patch(const DescJob& jb, int lane, uint32_t (&regs)[DESC_NLD],
                                                 int& off_out) {
  const int px0 = jb.x - DESC_R, py0 = jb.y - DESC_R;
  const int ax0 = px0 & ~3;
  off_out = px0 - ax0;
  int row = lane / (DESC_PITCH / 4), c = lane - row * (DESC_PITCH / 4);
#pragma unroll
  for (int k = 0; k < DESC_NLD; k++) {
    const int gy = py0 + row, gx = ax0 + 4 * c;
    uint32_t v = 0;
    if (lane + 64 * k < DESC_ROWS * (DESC_PITCH / 4) && (unsigned)gy < (unsigned)jb.h &&
        (unsigned)gx < (unsigned)jb.pitch)
      v = *reinterpret_cast<const uint32_t*>(jb.img + (uint32_t)(gy * jb.pitch + gx));
    regs[k] = v;
    row += 5;
    c += 4;
    if (c >= DESC_PITCH / 4) {
      c -= DESC_PITCH / 4;
      row += 1;
    }
  }
}

// per-wavefront LDS working set of k_describe2: its box table is written over `hs`, the patch stays intact, so
// there is no separate table (5.1 KB per wave instead of 6.1)
struct DescLds2 {
  uint32_t patch[DESC_ROWS * DESC_PITCH / 4];  // 41 x 48 u8
  uint16_t hs[DESC_ROWS * DESC_HP];            // horizontal 5-sums, then (37 rows) the 5x5 box sums
};

template <class L>
__device__ __forceinline__ void desc_store_patch(L& lds, int lane, const uint32_t (&regs)[DESC_NLD]) {
#pragma unroll
  for (int k = 0; k < DESC_NLD; k++)
    if (lane + 64 * k < DESC_ROWS * (DESC_PITCH / 4)) lds.patch[lane + 64 * k] = regs[k];
}

__device__ __forceinline__ DescJob desc_job(const OrbxPlan& plan, const uint8_t* pyr, int f, orbx_keypoint kp,
                                            int level) {
  const OrbxLevel& L = plan.L[level];
  DescJob jb;
  jb.img = pyr + (size_t)f * plan.frame_bytes + L.img_off;
  jb.w = L.w;
  jb.h = L.h;
  jb.pitch = L.pitch;
  jb.x = kp.x;
  jb.y = kp.y;
  jb.valid = true;
  return jb;
}

// The 5x5 box-sum table of k_describe2, horizontal and vertical pass fused: lane (chunk, g)
// walks 6 (chunk 0: 7) consecutive table rows of column group g (4 columns).  Per patch row it
// reads 8 bytes and gets the four horizontal 5-sums with one v_qsad_pk_u16_u8; the vertical sum
// is a running sum, box[r+1] = box[r] + hs[r+5] - hs[r], with the five live hs rows in
// registers.  The table (37 x DESC_HP u16) is written over lds.hs; the patch stays intact.
// LDS traffic per keypoint: 11 reads + 7 writes per lane instead of 32 accesses for two
// separate passes through an intermediate hs array.
template <class L>
__device__ __forceinline__ void desc_box_table_fused(L& lds, int lane) {
  static_assert(DESC_BROWS == 37, "6 chunks: 7 + 5 x 6 rows");
  const int chunk = lane / 10, g = lane - chunk * 10;
  if (chunk < 6) {
    const int r0 = chunk == 0 ? 0 : 6 * chunk + 1, nrows = chunk == 0 ? 7 : 6;
    const uint32_t* pp = &lds.patch[r0 * (DESC_PITCH / 4) + g];
    uint2* bp = reinterpret_cast<uint2*>(&lds.hs[r0 * DESC_HP + 4 * g]);
    auto hrow = [&](int i) {
      const u64 v = hsum5x4(pp[i * (DESC_PITCH / 4)], pp[i * (DESC_PITCH / 4) + 1]);
      return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    uint2 q[5];
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = hrow(i);
    uint2 acc;
    acc.x = pk_add(pk_add(pk_add(q[0].x, q[1].x), pk_add(q[2].x, q[3].x)), q[4].x);
    acc.y = pk_add(pk_add(pk_add(q[0].y, q[1].y), pk_add(q[2].y, q[3].y)), q[4].y);
    bp[0] = acc;
#pragma unroll
    for (int i = 1; i < 7; i++) {
      if (i < nrows) {
        const uint2 nw = hrow(i + 4);
        const uint2 od = q[(i - 1) % 5];
        acc.x = pk_sub(pk_add(acc.x, nw.x), od.x);
        acc.y = pk_sub(pk_add(acc.y, nw.y), od.y);
        q[(i - 1) % 5] = nw;
        bp[i * (DESC_HP / 4)] = acc;
      }
    }
  }
  wave_lds_sync();
}

// lroundf (half away from zero) with a float result, for |v| < 2^22: trunc(v) + trunc(2 * (v - trunc(v)));
// every step is exact, so this equals orbx_lroundf
__device__ __forceinline__ float lround_f(float v) {
  const float t = __builtin_truncf(v);
  const float fr = __fsub_rn(v, t);
  return __fadd_rn(t, __builtin_truncf(__fadd_rn(fr, fr)));
}

typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t lround_f2(f2_t v) {
  const f2_t t = __builtin_elementwise_trunc(v);
  const f2_t fr = v - t;
  return t + __builtin_elementwise_trunc(fr + fr);
}

template <int DESC_KPW, int DESC_NW, int DESC_OCC>
__global__ __launch_bounds__(64 * DESC_NW, DESC_OCC) void k_describe2(OrbxPlan plan, const uint8_t* __restrict__ pyr, int patch_size,
                                                   const int32_t* __restrict__ sel_count,
                                                   const orbx_keypoint* __restrict__ sel_lkp,
                                                   const float* __restrict__ sel_resp,
                                                   int32_t* __restrict__ out_count,
                                                   orbx_keypoint* __restrict__ out_lkp,
                                                   float* __restrict__ out_resp,
                                                   int32_t* __restrict__ out_level,
                                                   orbx_keypoint* __restrict__ out_kp,
                                                   uint32_t* __restrict__ out_kp16,
                                                   float* __restrict__ out_angle,
                                                   orbx_descriptor* __restrict__ out_desc,
                                                   const uint32_t* __restrict__ feedback, uint32_t* __restrict__ feedback_host,
                                                   OrbxHostRecord host) {
  constexpr int DESC_KPB = DESC_NW * DESC_KPW;
  static_assert(2 * DESC_KPB <= 64 * DESC_NW, "trig: two threads per keypoint");
  __shared__ __attribute__((aligned(16))) DescLds2 s_lds[DESC_NW];
  __shared__ int s_m[DESC_KPB][2];
  __shared__ float s_cs[DESC_KPB][2];
  __shared__ uint2 s_mw[4 * (DESC_PITCH / 4)];
  // the workgroup's compact records {x | y << 16, angle, descriptor} on their way to the host (orbx_set_host_results)
  __shared__ __attribute__((aligned(16))) uint32_t s_rec_kp[DESC_KPB], s_rec_angle[DESC_KPB], s_rec_desc[DESC_KPB][8];
  // grid = (frames, keypoint groups), the frame index dispatched fastest: with 8 XCDs dealt round-robin all
  // groups of a frame gather their patches through the same L2 (2.5 % faster than (groups, frames))
  const int f = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Final keypoint order = levels back to back (src/orb.cpp:100-102).  The selection
  // kernel left level l's keypoints in static slots; the prefix sums of the per-level
  // counts (lane l holds level l: one load, one DPP scan) map an output slot to them.
  static_assert(ORBX_MAX_LEVELS <= 64, "one lane per level");
  const int lvl_cnt = lane < plan.nlevels ? sel_count[f * plan.nlevels + lane] : 0;
  const int lvl_end = wave_scan_incl(lvl_cnt);  // slots of levels 0..lane
  const int count = __builtin_amdgcn_readlane(lvl_end, 63);
  if (grp == 0 && tid == 0) {
    out_count[f] = count;
    if (host.counts) host.counts[f] = count;
  }
  // (last kernel of a batch: the running totals of the top-rows-first pipeline's second pass go to the host's pinned
  // word pair, which the host reads without waiting when it enqueues later batches -- a copy node less per batch)
  // (+ the rows-needed word of every level: k_level_select above)
  if (feedback_host && grp == 0 && f == 0 && tid < ORBX_FEEDBACK_WORDS) feedback_host[tid] = feedback[tid];
  const int slot0 = grp * DESC_KPB;
  if (slot0 >= count) return;  // whole workgroup
  DescLds2& lds = s_lds[wave];
  const int pr = patch_size / 2;
  // moment weights of the four bytes of patch dword column c when the keypoint's patch starts
  // `off` bytes into its first dword: {x - x_kp + pr for bytes inside the orientation window
  // (else 0), 1 for bytes inside the window (else 0)}, for v_dot4_u32_u8
  if (tid < 4 * (DESC_PITCH / 4)) {
    const int off = tid / (DESC_PITCH / 4), c = tid - off * (DESC_PITCH / 4);
    uint32_t xw = 0, on = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int t = 4 * c + b - off - (DESC_R - pr);
      if (t >= 0 && t <= 2 * pr) {
        xw |= (uint32_t)t << (8 * b);
        on |= 1u << (8 * b);
      }
    }
    s_mw[tid] = make_uint2(xw, on);
  }
  // keypoint-independent geometry of the DESC_NLD patch dwords a lane holds: dword
  // lane + 64k is (row rowk[k], dword column (c0 + 4k) mod 12)
  int rowk[DESC_NLD], c3[3];
  uint32_t rin = 0;  // bit k: the dword exists and its row is inside the orientation window
  {
    int row = lane / (DESC_PITCH / 4), c = lane - row * (DESC_PITCH / 4);
#pragma unroll
    for (int k = 0; k < DESC_NLD; k++) {
      rowk[k] = row;
      if (k < 3) c3[k] = c;
      if (lane + 64 * k < DESC_ROWS * (DESC_PITCH / 4) && row >= DESC_R - pr && row <= DESC_R + pr) rin |= 1u << k;
      row += 5;
      c += 4;
      if (c >= DESC_PITCH / 4) {
        c -= DESC_PITCH / 4;
        row += 1;
      }
    }
  }
  __syncthreads();
  const size_t fo = (size_t)f * plan.out_cap;

  // this wave's keypoints; all their patches are requested before the first is used
  const int nk = min(DESC_KPW, max(0, count - (slot0 + wave * DESC_KPW)));  // wave-uniform
  DescJob jobs[DESC_KPW];
  float scale[DESC_KPW];
  uint32_t regs[DESC_KPW][DESC_NLD];
  int offs[DESC_KPW];
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    const int slot = min(slot0 + wave * DESC_KPW + j, count - 1);
    // level = number of levels that end at or before the slot (count > slot, so it is < nlevels)
    const int level = __popcll(__ballot(lvl_end <= slot));
    const int lvl_begin = __builtin_amdgcn_readlane(lvl_end - lvl_cnt, level);
    const size_t src = fo + plan.L[level].out_off + (slot - lvl_begin);
    const orbx_keypoint kp = sel_lkp[src];
    jobs[j] = desc_job(plan, pyr, f, kp, level);
    scale[j] = plan.L[level].scale;
    if (j < nk && lane == 0) {  // the compacted per-keypoint records of the result block
      out_lkp[fo + slot] = kp;
      out_resp[fo + slot] = sel_resp[src];
      out_level[fo + slot] = level;
    }
  }
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    const DescJob& jb = jobs[j];
    const int py0 = jb.y - DESC_R, ax0 = (jb.x - DESC_R) & ~3;
    // wave-uniform: every dword of the 41 x 48 neighbourhood lies inside the (zero-padded) level
    if (py0 >= 0 && py0 + DESC_ROWS <= jb.h && ax0 >= 0 && ax0 + DESC_PITCH <= jb.pitch) {
      offs[j] = jb.x - DESC_R - ax0;
      const uint32_t base = (uint32_t)(py0 * jb.pitch + ax0);
#pragma unroll
      for (int k = 0; k < DESC_NLD; k++) {
        regs[j][k] = 0;
        if (lane + 64 * k < DESC_ROWS * (DESC_PITCH / 4))
          // (full-rate 24-bit multiply: rows < 41, pitch <= 16384 -- v_mul_lo_u32 is a quarter-rate instruction)
          regs[j][k] = *reinterpret_cast<const uint32_t*>(jb.img + (base + (__umul24((uint32_t)rowk[k], (uint32_t)jb.pitch) + 4u * (uint32_t)c3[k % 3])));
      }
    } else {
      desc_fetch_patch(jb, lane, regs[j], offs[j]);
    }
  }

  // pass A: moments
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    if (j < nk) {
      const DescJob& jb = jobs[j];
      const int q = wave * DESC_KPW + j, off = offs[j];
      int m10 = 0, m01 = 0;
      // full patch must lie inside the image, else angle 0 (src/orb_cpu.cpp:152-156)
      if (!(jb.x - pr < 0 || jb.x + pr >= jb.w || jb.y - pr < 0 || jb.y + pr >= jb.h)) {
        // straight from the registers: per dword one dot product with the x weights and one
        // with the window mask; exact integers (< 2^24, like the reference's float sums)
        uint2 mw[3];
#pragma unroll
        for (int m = 0; m < 3; m++) mw[m] = s_mw[off * (DESC_PITCH / 4) + c3[m]];
        uint32_t X = 0, S = 0, M = 0;
#pragma unroll
        for (int k = 0; k < DESC_NLD; k++) {
          const uint32_t z = regs[j][k] & (uint32_t)(-(int32_t)((rin >> k) & 1u));
          X = __builtin_amdgcn_udot4(z, mw[k % 3].x, X, false);
          const uint32_t rs = __builtin_amdgcn_udot4(z, mw[k % 3].y, 0u, false);
          S += rs;
          M += __umul24((uint32_t)rowk[k], rs);  // (rows < 41, rs <= 1020: the full-rate 24-bit multiply-add)
        }
        m10 = wave_sum((int)X - __mul24(pr, (int)S));  // (S <= 8 x 1020, pr <= 20: 24-bit operands)
        m01 = wave_sum((int)M - __mul24(DESC_R, (int)S));
      }
      if (lane == 0) {
        s_m[q][0] = m10;
        s_m[q][1] = m01;
      }
    }
  }
  __syncthreads();
  // trig: TWO threads per keypoint -- both compute the angle, then the even one its cosine and the odd
  // one its sine through the shared sincosf body (one pass through its two polynomial branches
  // instead of two).  Moments are exact integers < 2^24, so the reference's float accumulation
  // equals them; atan2f(0,0) = 0 covers the border case.
  if (tid < 2 * DESC_KPB && slot0 + (tid >> 1) < count) {
    const int q = tid >> 1, want_sin = tid & 1;
    const float angle = orbx_atan2f((float)s_m[q][1], (float)s_m[q][0]);
    s_cs[q][want_sin] = orbx_sincosf_impl(angle, want_sin ^ 1);
    if (!want_sin) {
      out_angle[fo + slot0 + q] = angle;
      s_rec_angle[q] = __float_as_uint(angle);
    }
  }
  __syncthreads();

  // pass C: descriptors (the patches are still in registers).  A lane evaluates the same
  // four tests for every keypoint: its pattern points are converted to float once.
  f2_t patx[4], paty[4];  // {x1, x2}, {y1, y2} of test 64k + lane
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int32_t pk = reinterpret_cast<const int32_t*>(c_pattern)[k * 64 + lane];
    patx[k] = f2_t{(float)(int8_t)(pk & 0xff), (float)(int8_t)((pk >> 16) & 0xff)};
    paty[k] = f2_t{(float)(int8_t)((pk >> 8) & 0xff), (float)(int8_t)((pk >> 24) & 0xff)};
  }
#pragma unroll
  for (int j = 0; j < DESC_KPW; j++) {
    if (j < nk) {
      const DescJob& jb = jobs[j];
      const int q = wave * DESC_KPW + j, slot = slot0 + q, off = offs[j];
      desc_store_patch(lds, lane, regs[j]);
      wave_lds_sync();
      desc_box_table_fused(lds, lane);
      const float c = s_cs[q][0], s = s_cs[q][1];
      // every rotated centre is within 18 px of the keypoint; with a 20 px margin no test can be skipped
      const bool interior = jb.x >= DESC_R && jb.y >= DESC_R && jb.x < jb.w - DESC_R && jb.y < jb.h - DESC_R;
      u64 d[4];
      if (interior) {
        // no test can be skipped: lround as float arithmetic (trunc(v) + trunc(2 * frac), exact), the
        // table index as one exact fma, the constant part of the index in the instruction offset
        const uint16_t* tbl = &lds.hs[18 * DESC_HP + 18 + off];
        // both points of a test side by side in the packed-f32 lanes (v_pk_mul/add/fma_f32);
        // no contraction (the TU is built with -ffp-contract=off), so each product and sum
        // rounds like the reference's scalar code
        const f2_t C2 = {c, c}, S2 = {s, s}, HP2 = {(float)DESC_HP, (float)DESC_HP};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const f2_t X = patx[k], Y = paty[k];
          const f2_t rx = lround_f2(C2 * X - S2 * Y), ry = lround_f2(S2 * X + C2 * Y);
          const f2_t idx = __builtin_elementwise_fma(ry, HP2, rx);
          d[k] = __ballot(tbl[(int)idx.x] < tbl[(int)idx.y]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float x1 = patx[k].x, y1 = paty[k].x, x2 = patx[k].y, y2 = paty[k].y;
          const int dx1 = orbx_lroundf(__fsub_rn(__fmul_rn(c, x1), __fmul_rn(s, y1)));
          const int dy1 = orbx_lroundf(__fadd_rn(__fmul_rn(s, x1), __fmul_rn(c, y1)));
          const int dx2 = orbx_lroundf(__fsub_rn(__fmul_rn(c, x2), __fmul_rn(s, y2)));
          const int dy2 = orbx_lroundf(__fadd_rn(__fmul_rn(s, x2), __fmul_rn(c, y2)));
          const int cx1 = jb.x + dx1, cy1 = jb.y + dy1, cx2 = jb.x + dx2, cy2 = jb.y + dy2;
          const bool ok = !(cx1 < 2 || cy1 < 2 || cx1 > jb.w - 1 || cy1 > jb.h - 1 || cx2 < 2 || cy2 < 2 ||
                            cx2 > jb.w - 1 || cy2 > jb.h - 1);
          const int s1 = lds.hs[(dy1 + 18) * DESC_HP + dx1 + 18 + off];
          const int s2 = lds.hs[(dy2 + 18) * DESC_HP + dx2 + 18 + off];
          d[k] = __ballot(ok && s1 < s2);
        }
      }
      wave_lds_sync();
      if (lane == 0) {
        orbx_keypoint g;  // kp.x *= scale on int (src/orb.cpp:94-98)
        g.x = (int)__fmul_rn((float)jb.x, scale[j]);
        g.y = (int)__fmul_rn((float)jb.y, scale[j]);
        out_kp[fo + slot] = g;
        // the same once more as x | y << 16 (coordinates <= 16384): the compact host record, 4 bytes less per slot
        out_kp16[fo + slot] = (uint32_t)g.x | ((uint32_t)g.y << 16);
        u64* dd = reinterpret_cast<u64*>(out_desc + fo + slot);
        dd[0] = d[0];
        dd[1] = d[1];
        dd[2] = d[2];
        dd[3] = d[3];
        if (host.desc) {
          s_rec_kp[q] = (uint32_t)g.x | ((uint32_t)g.y << 16);
          u64* sd = reinterpret_cast<u64*>(s_rec_desc[q]);
          sd[0] = d[0];
          sd[1] = d[1];
          sd[2] = d[2];
          sd[3] = d[3];
        }
      }
    }
  }
  // The compact record of the workgroup's keypoints, straight into the PINNED HOST mirror of the result block: three
  // coalesced stores (16 x 32 B of descriptors, 16 x 4 B of packed keypoints, 16 x 4 B of angles) that travel the
  // host link while the kernel runs -- no copy kernel afterwards, no wait between two streams (orbx_set_host_results)
  if (host.desc) {  // kernel argument: uniform
    __syncthreads();
    const int nq = min(DESC_KPB, count - slot0);
    if (tid < 2 * nq) {
      const uint4 v = reinterpret_cast<const uint4*>(s_rec_desc[tid >> 1])[tid & 1];
      reinterpret_cast<uint4*>(host.desc + fo + slot0)[tid] = v;
    }
    if (tid < nq) {
      host.kp16[fo + slot0 + tid] = s_rec_kp[tid];
      host.angle[fo + slot0 + tid] = __uint_as_float(s_rec_angle[tid]);
    }
  }
}

__global__ __launch_bounds__(256) void k_describe_flat(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                       const orbx_keypoint* __restrict__ kps, int nkp,
                                                       int patch_size, int use_given_angles, int do_brief,
                                                       float* __restrict__ angles,
                                                       orbx_descriptor* __restrict__ desc) {
  __shared__ __attribute__((aligned(16))) DescLds s_lds[4];
  const int wave = threadIdx.x >> 6;
  const int slot = blockIdx.x * 4 + wave;
  DescJob jb;
  jb.valid = slot < nkp;
  jb.img = img;
  jb.w = w;
  jb.h = h;
  jb.pitch = pitch;
  jb.x = jb.valid ? kps[slot].x : 0;
  jb.y = jb.valid ? kps[slot].y : 0;
  const float given = (jb.valid && use_given_angles) ? angles[slot] : 0.0f;
  float angle;
  u64 d[4];
  describe_wave(jb, s_lds[wave], patch_size, use_given_angles != 0, given, do_brief != 0, angle, d);
  if (jb.valid && lane_id() == 0) {
    if (!use_given_angles) angles[slot] = angle;
    if (do_brief) {
      u64* dd = reinterpret_cast<u64*>(desc + slot);
      dd[0] = d[0];
      dd[1] = d[1];
      dd[2] = d[2];
      dd[3] = d[3];
    }
  }
}

// ---------------------------------------------------------------------------
// 8. Hamming 2-nearest-neighbour search over 256-bit descriptors + ratio test
//    (next row, SURVEY.md §8f rank 1: flann->knnMatch(des1, des2, matches, 2) and
//    the `m.distance < 0.8 * n.distance` filter, src/feature_matching.cpp:166-181,
//    src/feature_tracking.cpp:203-219 -- exact brute force instead of FLANN's
//    approximate LSH index).
//    One thread owns one query descriptor (4 x u64 in registers); the train set
//    streams through LDS in tiles of 256 descriptors (every lane reads the same
//    16-byte words: broadcast, conflict-free); distance = 4 x popcount(xor).
//    Ties keep the lower train index.  blockIdx.y = pair p: query set p is
//    qbase + p*qstride with qcount[p] entries, train set likewise.
__global__ __launch_bounds__(256) void k_knn2(const orbx_descriptor* __restrict__ qbase,
                                              const int32_t* __restrict__ qcount, size_t qstride,
                                              const orbx_descriptor* __restrict__ tbase,
                                              const int32_t* __restrict__ tcount, size_t tstride, double ratio,
                                              int32_t* __restrict__ knn_idx, int32_t* __restrict__ knn_dist,
                                              int32_t* __restrict__ match, size_t ostride) {
  __shared__ __attribute__((aligned(16))) uint4 s_t[256 * 2];
  const int p = blockIdx.y, tid = threadIdx.x;
  const int nq = qcount[p], nt = tcount[p];
  if ((int)(blockIdx.x * 256) >= nq) return;  // whole workgroup
  const int q = blockIdx.x * 256 + tid;
  const bool valid = q < nq;
  const uint4* qp = reinterpret_cast<const uint4*>(qbase + p * qstride + (valid ? q : 0));
  const uint4 qa = qp[0], qb = qp[1];
  const u64 q0 = ((u64)qa.y << 32) | qa.x, q1 = ((u64)qa.w << 32) | qa.z;
  const u64 q2 = ((u64)qb.y << 32) | qb.x, q3 = ((u64)qb.w << 32) | qb.z;
  int d1 = 1 << 30, d2 = 1 << 30, j1 = -1, j2 = -1;
  const orbx_descriptor* tp = tbase + p * tstride;
  for (int t0 = 0; t0 < nt; t0 += 256) {
    const int m = min(256, nt - t0);
    if (tid < m) {
      const uint4* src = reinterpret_cast<const uint4*>(tp + t0 + tid);
      s_t[2 * tid] = src[0];
      s_t[2 * tid + 1] = src[1];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < m; j++) {
      const uint4 a = s_t[2 * j], b = s_t[2 * j + 1];
      const int d = __popcll(q0 ^ (((u64)a.y << 32) | a.x)) + __popcll(q1 ^ (((u64)a.w << 32) | a.z)) +
                    __popcll(q2 ^ (((u64)b.y << 32) | b.x)) + __popcll(q3 ^ (((u64)b.w << 32) | b.z));
      if (d < d1) {
        d2 = d1;
        j2 = j1;
        d1 = d;
        j1 = t0 + j;
      } else if (d < d2) {
        d2 = d;
        j2 = t0 + j;
      }
    }
    __syncthreads();
  }
  if (valid) {
    const size_t o = p * ostride + q;
    knn_idx[2 * o] = j1;
    knn_idx[2 * o + 1] = j2;
    knn_dist[2 * o] = j1 >= 0 ? d1 : -1;
    knn_dist[2 * o + 1] = j2 >= 0 ? d2 : -1;
    // DMatch::distance is a float; the reference compares in double (0.8 is a double literal)
    const bool pass = j2 >= 0 && (double)(float)d1 < ratio * (double)(float)d2;
    match[o] = pass ? j1 : -1;
  }
}

// ---------------------------------------------------------------------------
// stage-level helpers (not on the batched hot path)

// thresholded NMS of an arbitrary float score map (src/cuda/NMS.cu:21-128):
// border of width r rejected, val > threshold, no strictly greater neighbour.
__global__ __launch_bounds__(256) void k_nms_f32(const float* __restrict__ scores, int w, int h, int r,
                                                 float threshold, u64* __restrict__ mask, int mask_wpr) {
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  bool keep = false;
  if (y < h && x < w && !(x < r || y < r || x >= w - r || y >= h - r)) {
    const float v = scores[(size_t)y * w + x];
    keep = v > threshold;
    if (keep)
      for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) keep = keep && !(scores[(size_t)(y + dy) * w + (x + dx)] > v);
  }
  const u64 m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && y < h) mask[(size_t)y * mask_wpr + blockIdx.x] = m;
}

// generic KxK correlation (src/cuda/Convolution.cu:20-55 arithmetic: float
// accumulate in (i,j) row-major order) with the wrapper's CV_8U conversion
// (round-half-even + saturate, :100).  reflect_pad=0: valid correlation of a
// pre-padded image; reflect_pad=1: output same size, REFLECT_101 borders
// (== copyMakeBorder + conv2d, src/GaussianBlur.cpp:44-46, src/Sobel.cpp:29-31).
__global__ __launch_bounds__(256) void k_conv2d(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                const float* __restrict__ kern, int K, int reflect_pad,
                                                uint8_t* __restrict__ dst, int dst_pitch) {
  const int wo = reflect_pad ? w : w - K + 1, ho = reflect_pad ? h : h - K + 1;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= wo || y >= ho) return;
  const int r = K / 2;
  float sum = 0.f;
  for (int i = 0; i < K; i++)
    for (int j = 0; j < K; j++) {
      int yy, xx;
      if (reflect_pad) {
        yy = reflect101(y - r + i, h);
        xx = reflect101(x - r + j, w);
      } else {
        yy = y + i;
        xx = x + j;
      }
      sum = __fadd_rn(sum, __fmul_rn((float)img[(size_t)yy * pitch + xx], kern[i * K + j]));
    }
  int v = __float2int_rn(sum);
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  dst[(size_t)y * dst_pitch + x] = (uint8_t)v;
}

// ---------------------------------------------------------------------------
// launchers

#define ORBX_LAUNCH_CHECK() hipGetLastError()

hipError_t orbx_launch_pyramid2(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int w0,
                                int h0, int n_frames, const uint8_t* d_in, int in_stride, size_t in_frame_stride,
                                const OrbxResizeTap* d_taps, uint8_t* d_pyr) {
  if (n_tiles <= 0) return hipSuccess;
  dim3 grid(n_tiles | 1, n_frames);  // odd: the XCD assignment rotates from frame to frame (orbx_blur.hip)
  hipLaunchKernelGGL(k_pyramid2, grid, dim3(256), 0, s, d_tiles, n_tiles, frame_bytes, w0, h0, d_in, in_stride,
                     in_frame_stride, d_taps, d_pyr);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_blur(hipStream_t s, const OrbxPlan& plan, const OrbxTileMap& tm, int n_frames,
                            const uint8_t* d_src, uint8_t* d_dst, int first_level, int kind) {
  dim3 grid(tm.begin[plan.nlevels], n_frames);
  hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, s, plan, tm, d_src, d_dst, first_level, kind);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_compact(hipStream_t s, const OrbxPlan& plan, int n_frames,
                               const unsigned long long* d_mask, orbx_keypoint* d_cand, int32_t* d_cand_count,
                               int32_t* d_cand_total, int need_total) {
  dim3 grid(plan.nlevels, n_frames);
  hipLaunchKernelGGL(k_compact, grid, dim3(256), 0, s, plan, d_mask, d_cand, d_cand_count, d_cand_total,
                     need_total);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_level_select(hipStream_t s, const OrbxPlan& plan, int n_frames, int mode,
                                    const unsigned long long* d_mask, const uint8_t* d_pyr, const float* d_gauss,
                                    int window, float k, orbx_keypoint* d_sel_lkp, float* d_sel_resp,
                                    int32_t* d_sel_count, uint32_t* d_need) {
  int maxcap = 2;
  for (int l = 0; l < plan.nlevels; l++) maxcap = plan.L[l].cap > maxcap ? plan.L[l].cap : maxcap;
  const size_t lds = (size_t)((maxcap + 1) & ~1) * 16;
  dim3 grid(n_frames, plan.nlevels);
  hipLaunchKernelGGL(k_level_select, grid, dim3(LVL_THREADS), lds, s, plan, mode, d_mask, d_pyr, d_gauss, window, k,
                     d_sel_lkp, d_sel_resp, d_sel_count, d_need);
  return ORBX_LAUNCH_CHECK();
}

// one workgroup per (level, frame) or the three spread kernels, whichever suits the shape
// (see 6c); force = 0 fused, 1 spread, -1 automatic
hipError_t orbx_launch_level_select_auto(hipStream_t s, const OrbxPlan& plan, int n_frames, int mode, int force,
                                         const unsigned long long* d_mask, const uint8_t* d_pyr,
                                         const float* d_gauss, int window, float k, uint32_t* d_cand,
                                         int32_t* d_ncand, float* d_cresp, orbx_keypoint* d_sel_lkp,
                                         float* d_sel_resp, int32_t* d_sel_count, uint32_t* d_need) {
  int maxcap = 2;
  for (int l = 0; l < plan.nlevels; l++) maxcap = plan.L[l].cap > maxcap ? plan.L[l].cap : maxcap;
  const bool spread = mode == ORBX_SELECT_HARRIS &&
                      (force >= 0 ? force != 0 : maxcap > LVL_THREADS);  // more than one candidate per thread
  if (!spread)
    return orbx_launch_level_select(s, plan, n_frames, mode, d_mask, d_pyr, d_gauss, window, k, d_sel_lkp,
                                    d_sel_resp, d_sel_count, d_need);
  hipLaunchKernelGGL(k_lvl_compact, dim3(plan.nlevels, n_frames), dim3(256), 0, s, plan, d_mask, d_cand, d_ncand, d_need);
  hipLaunchKernelGGL(k_lvl_harris, dim3((maxcap + 255) / 256, plan.nlevels, n_frames), dim3(256), 0, s, plan, d_pyr,
                     d_gauss, window, k, d_cand, d_ncand, d_cresp);
  const size_t lds = (size_t)((maxcap + 7) & ~7) * 8;
  hipLaunchKernelGGL(k_lvl_rank, dim3((maxcap + RANK_SLICE - 1) / RANK_SLICE, plan.nlevels, n_frames), dim3(256), lds,
                     s, plan, d_cand, d_ncand, d_cresp, d_sel_lkp, d_sel_resp, d_sel_count);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_describe(hipStream_t s, const OrbxPlan& plan, int n_frames, const uint8_t* d_pyr,
                                int patch_size, const int32_t* d_sel_count, const orbx_keypoint* d_sel_lkp,
                                const float* d_sel_resp, int32_t* d_out_count, orbx_keypoint* d_out_lkp,
                                float* d_out_resp, int32_t* d_out_level, orbx_keypoint* d_out_kp, uint32_t* d_out_kp16,
                                float* d_out_angle, orbx_descriptor* d_out_desc, const uint32_t* d_feedback,
                                uint32_t* h_feedback, const OrbxHostRecord* host_record) {
  const OrbxHostRecord host = host_record ? *host_record : OrbxHostRecord{};
  if (plan.out_cap <= 0) return hipSuccess;
  // few keypoints in flight (single frames, small batches): one keypoint per wave, four times the
  // workgroups, a quarter of the serial work per wave; else four per wave
  if ((long long)plan.out_cap * n_frames <= 8192) {
    dim3 grid(n_frames, (plan.out_cap + 3) / 4);
    hipLaunchKernelGGL((k_describe2<1, 4, 1>), grid, dim3(256), 0, s, plan, d_pyr, patch_size, d_sel_count, d_sel_lkp,
                       d_sel_resp, d_out_count, d_out_lkp, d_out_resp, d_out_level, d_out_kp, d_out_kp16, d_out_angle, d_out_desc, d_feedback,
                       h_feedback, host);
  } else {
    // Four waves of four keypoints, registers capped for 7 waves per SIMD (5.1 KB of LDS per wave: 7 workgroups
    // per CU).  Measured per 256 frames: 8-wave workgroups at 6 waves per SIMD (the trig -- two threads per
    // keypoint -- fills a wave: -1.8 % instructions) 203 us, 7-wave workgroups 221 us, this 197 us.
    dim3 grid(n_frames, (plan.out_cap + 15) / 16);
    hipLaunchKernelGGL((k_describe2<4, 4, 7>), grid, dim3(256), 0, s, plan, d_pyr, patch_size, d_sel_count, d_sel_lkp,
                       d_sel_resp, d_out_count, d_out_lkp, d_out_resp, d_out_level, d_out_kp, d_out_kp16, d_out_angle, d_out_desc, d_feedback,
                       h_feedback, host);
  }
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_describe_flat(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch,
                                     const orbx_keypoint* d_kps, int nkp, int patch_size, int use_given_angles,
                                     int do_brief, float* d_angles, orbx_descriptor* d_desc) {
  if (nkp <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_describe_flat, dim3((nkp + 3) / 4), dim3(256), 0, s, d_img, w, h, pitch, d_kps, nkp,
                     patch_size, use_given_angles, do_brief, d_angles, d_desc);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_harris_flat(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch,
                                   const orbx_keypoint* d_kps, int nkp, const float* d_gauss, int K, float kk,
                                   float* d_resp) {
  if (nkp <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_harris2_flat, dim3((nkp + 31) / 32), dim3(256), 0, s, d_img, w, h, pitch, d_kps, nkp,
                     d_gauss, K, kk, d_resp);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_nms_f32(hipStream_t s, const float* d_scores, int w, int h, int radius, float threshold,
                               unsigned long long* d_mask, int mask_wpr) {
  dim3 grid((w + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_nms_f32, grid, dim3(256), 0, s, d_scores, w, h, radius, threshold, d_mask, mask_wpr);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_conv2d(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch, const float* d_kernel,
                              int K, int reflect_pad, uint8_t* d_dst, int dst_pitch) {
  const int wo = reflect_pad ? w : w - K + 1, ho = reflect_pad ? h : h - K + 1;
  if (wo <= 0 || ho <= 0) return hipSuccess;
  dim3 grid((wo + 63) / 64, (ho + 3) / 4);
  hipLaunchKernelGGL(k_conv2d, grid, dim3(256), 0, s, d_img, w, h, pitch, d_kernel, K, reflect_pad, d_dst,
                     dst_pitch);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_select_flat(hipStream_t s, const float* d_resp, int n, int keep, int32_t* d_idx) {
  if (n <= 0) return hipSuccess;
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_select_flat, dim3(blocks), dim3(256), 0, s, d_resp, n, keep, d_idx);
  return ORBX_LAUNCH_CHECK();
}

hipError_t orbx_launch_knn2(hipStream_t s, int npairs, int max_nq, const orbx_descriptor* d_q, const int32_t* d_qcount,
                            size_t qstride, const orbx_descriptor* d_t, const int32_t* d_tcount, size_t tstride,
                            double ratio, int32_t* d_idx, int32_t* d_dist, int32_t* d_match, size_t ostride) {
  if (npairs <= 0 || max_nq <= 0) return hipSuccess;
  dim3 grid((max_nq + 255) / 256, npairs);
  hipLaunchKernelGGL(k_knn2, grid, dim3(256), 0, s, d_q, d_qcount, qstride, d_t, d_tcount, tstride, ratio, d_idx,
                     d_dist, d_match, ostride);
  return ORBX_LAUNCH_CHECK();
}
