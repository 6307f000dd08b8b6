// orbx_lk.hip -- pyramidal Lucas-Kanade tracking for gfx950.
//
// Replaces cv::calcOpticalFlowPyrLK(img1, img2, pts1, pts2, status, err, Size(21,21), 3,
// TermCriteria(COUNT+EPS, 30, 0.01)) as called by src/feature_tracking.cpp:175-181.  The
// arithmetic follows OpenCV 4.x's published algorithm (lkpyramid.cpp: 14-bit fixed-point
// bilinear weights, 5 fractional bits on intensities, Scharr derivatives, minimum
// eigenvalue test, epsilon / oscillation stopping rules) with ONE documented difference:
// the window sums are accumulated exactly in integers (wave reduction of 16-bit halves)
// and rounded to float once, where OpenCV accumulates in float in a SIMD-width dependent
// order.  oracle/lk_oracle.c restates the same thing on the CPU; results are bit-identical.
//
// Kernels
//   k_lk_pyrdown  cv::pyrDown (5x5 [1 4 6 4 1]^2 / 256, REFLECT_101), thread per output pixel
//   k_lk_scharr   Scharr (3,10,3) derivative pairs as int16, REFLECT_101, thread per pixel
//   k_lk_track    ONE WAVEFRONT PER POINT walks the levels top -> 0.  The 21x21 template
//                 (intensity + two gradients, int16) and a 26x26 neighbourhood of the next
//                 image live in the wave's LDS slice; every Newton step samples the 441
//                 window pixels from LDS (7 per lane) and reduces the two mismatch sums with
//                 DPP; the neighbourhood is re-fetched only when the window leaves it.  All
//                 control flow is wave-uniform, there is no workgroup barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_internal.h"

namespace {

__device__ __forceinline__ int lk_reflect(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

__global__ __launch_bounds__(256) void k_lk_pyrdown(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                    uint8_t* __restrict__ dst, int dw, int dh, int dpitch) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  int xs[5];
#pragma unroll
  for (int j = 0; j < 5; j++) xs[j] = lk_reflect(2 * x + j - 2, sw);
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint8_t* row = src + (size_t)lk_reflect(2 * y + i - 2, sh) * spitch;
    const int hs = row[xs[0]] + 4 * row[xs[1]] + 6 * row[xs[2]] + 4 * row[xs[3]] + row[xs[4]];
    sum += (i == 0 || i == 4) ? hs : (i == 2 ? 6 * hs : 4 * hs);
  }
  dst[(size_t)y * dpitch + x] = (uint8_t)((sum + 128) >> 8);
}

__global__ __launch_bounds__(256) void k_lk_scharr(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                   short2* __restrict__ deriv) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const uint8_t* r0 = img + (size_t)lk_reflect(y - 1, h) * pitch;
  const uint8_t* r1 = img + (size_t)y * pitch;
  const uint8_t* r2 = img + (size_t)lk_reflect(y + 1, h) * pitch;
  const int xl = lk_reflect(x - 1, w), xr = lk_reflect(x + 1, w);
  const int t0l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, t0r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
  const int t1l = r2[xl] - r0[xl], t1c = r2[x] - r0[x], t1r = r2[xr] - r0[xr];
  deriv[(size_t)y * w + x] = make_short2((short)(t0r - t0l), (short)((t1r + t1l) * 3 + t1c * 10));
}

__device__ __forceinline__ int lk_wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141 /*row_half_mirror*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140 /*row_mirror*/, 0xf, 0xf, true);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// exact sum over the wave of per-lane int32 partials, as the float the oracle gets from
// (float)(double)int64_sum: the 16-bit halves are reduced separately (no overflow), joined
// exactly in double, rounded once
__device__ __forceinline__ float lk_exact_sum_f32(int partial) {
  const int lo = lk_wave_sum(partial & 0xffff), hi = lk_wave_sum(partial >> 16);
  return (float)((double)hi * 65536.0 + (double)lo);
}

__device__ __forceinline__ int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

struct LkWeights {
  int w00, w01, w10, w11;
};
__device__ __forceinline__ LkWeights lk_weights(float a, float b) {
  LkWeights r;
  const float s = (float)(1 << 14);
  r.w00 = (int)rintf(__fmul_rn(__fmul_rn(1.f - a, 1.f - b), s));
  r.w01 = (int)rintf(__fmul_rn(__fmul_rn(a, 1.f - b), s));
  r.w10 = (int)rintf(__fmul_rn(__fmul_rn(1.f - a, b), s));
  r.w11 = (1 << 14) - r.w00 - r.w01 - r.w10;
  return r;
}

// bilinear sample of an 8-bit level at (X, Y)..(X+1, Y+1), 5 fractional bits kept.
// interior: the four pixels are inside the image (wave-uniform fact about the whole window)
__device__ __forceinline__ int lk_sample(const OrbxLkLevel& L, int X, int Y, const LkWeights& w, bool interior) {
  int p00, p01, p10, p11;
  if (interior) {
    const uint8_t* p = L.img + (size_t)Y * L.pitch + X;
    p00 = p[0];
    p01 = p[1];
    p10 = p[L.pitch];
    p11 = p[L.pitch + 1];
  } else {
    const int x0 = lk_reflect(X, L.w), x1 = lk_reflect(X + 1, L.w);
    const uint8_t* r0 = L.img + (size_t)lk_reflect(Y, L.h) * L.pitch;
    const uint8_t* r1 = L.img + (size_t)lk_reflect(Y + 1, L.h) * L.pitch;
    p00 = r0[x0];
    p01 = r0[x1];
    p10 = r1[x0];
    p11 = r1[x1];
  }
  return lk_descale(p00 * w.w00 + p01 * w.w01 + p10 * w.w10 + p11 * w.w11, 14 - 5);
}

__device__ __forceinline__ short2 lk_deriv_at(const OrbxLkLevel& L, int X, int Y, bool interior) {
  if (!interior && (X < 0 || Y < 0 || X >= L.w || Y >= L.h)) return make_short2(0, 0);
  return reinterpret_cast<const short2*>(L.deriv)[(size_t)Y * L.w + X];
}

#define LK_MAX_WIN 31
// one pass over the window items of a lane: unrolled when the trip count is a template constant
#define LK_ITEM_LOOP                                                                           \
  _Pragma("unroll 8") for (int it_ = 0, idx = lane; it_ < (NIT ? NIT : (nitem + 63) / 64); it_++, idx += 64) \
    if (idx < nitem)
// The Newton steps of a level move the window by fractions of a pixel, so its integer
// origin hardly ever changes: the (win + 1 + 2 * LK_JC_MARGIN)^2 neighbourhood of the next
// image is cached in LDS (border reflection resolved while loading) and re-fetched only when
// the window leaves it.  One global round trip per level instead of one per iteration.
#define LK_JC_MARGIN 2
#define LK_JC_MAX (LK_MAX_WIN + 1 + 2 * LK_JC_MARGIN)  // 36

__device__ __forceinline__ void lk_wave_lds_sync() {
  // LDS operations of one wave execute in order; this only pins the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LkCache {
  uint8_t* px;  // jce x jce bytes
  int jce, x0, y0;
  uint32_t rcp;
  bool valid;
};

// makes sure the cache covers [inx, inx + win] x [iny, iny + win]
__device__ __forceinline__ void lk_cache_cover(LkCache& c, const OrbxLkLevel& J, int inx, int iny, int win, int lane) {
  if (c.valid && inx >= c.x0 && iny >= c.y0 && inx + win + 1 <= c.x0 + c.jce && iny + win + 1 <= c.y0 + c.jce) return;
  lk_wave_lds_sync();  // earlier reads of the old contents are done
  c.x0 = inx - LK_JC_MARGIN;
  c.y0 = iny - LK_JC_MARGIN;
  const bool interior = c.x0 >= 0 && c.y0 >= 0 && c.x0 + c.jce <= J.w && c.y0 + c.jce <= J.h;
  for (int idx = lane; idx < c.jce * c.jce; idx += 64) {
    const int y = (int)(((uint32_t)idx * c.rcp) >> 16), x = idx - y * c.jce;
    int v;
    if (interior)
      v = J.img[(size_t)(c.y0 + y) * J.pitch + (c.x0 + x)];
    else
      v = J.img[(size_t)lk_reflect(c.y0 + y, J.h) * J.pitch + lk_reflect(c.x0 + x, J.w)];
    c.px[idx] = (uint8_t)v;
  }
  c.valid = true;
  lk_wave_lds_sync();
}

// bilinear sample of the cached next image at window position (x, y) of origin (inx, iny)
__device__ __forceinline__ int lk_sample_cached(const LkCache& c, int inx, int iny, int x, int y, const LkWeights& w) {
  const uint8_t* p = c.px + (iny - c.y0 + y) * c.jce + (inx - c.x0 + x);
  return lk_descale(p[0] * w.w00 + p[1] * w.w01 + p[c.jce] * w.w10 + p[c.jce + 1] * w.w11, 14 - 5);
}

// NIT: passes of the 64 lanes over the win x win window, known at compile time for the
// reference's 21 x 21 window (7: the loops unroll and the LDS / global reads of a pass set are
// all in flight together), 0 = run-time trip count for any other window size
template <int NIT>
__global__ __launch_bounds__(256) void k_lk_track(OrbxLkPyr P, OrbxLkPyr N, int n, const float2* __restrict__ prev_pts,
                                                  float2* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                  float* __restrict__ err, int win, int max_iters, double eps2) {
  // per wave: the template, (intensity, dx, dy) as int16 (4th lane of the short4 unused)
  __shared__ short4 s_tpl[4][LK_MAX_WIN * LK_MAX_WIN];
  __shared__ uint8_t s_jc[4][LK_JC_MAX * LK_JC_MAX];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;  // whole wave
  short4* tpl = s_tpl[wave];
  LkCache jc;
  jc.px = s_jc[wave];
  jc.jce = win + 1 + 2 * LK_JC_MARGIN;
  jc.rcp = (65536u + (uint32_t)jc.jce - 1u) / (uint32_t)jc.jce;  // idx / jce for idx < 36^2
  jc.x0 = jc.y0 = 0;
  jc.valid = false;
  const int nitem = win * win;
  const uint32_t rcp = (65536u + (uint32_t)win - 1u) / (uint32_t)win;  // idx / win == (idx * rcp) >> 16 for idx < 961
  const float half = (float)(win - 1) * 0.5f;
  const float flt_scale = 1.f / (float)(1 << 20);
  // the point is the same in every lane: keep it (and everything derived from it) wave-uniform
  float2 pp = prev_pts[i];
  pp.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pp.x)));
  pp.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pp.y)));
  int st = 1;
  float ev = 0.f;
  float outx = 0.f, outy = 0.f;
  for (int level = P.top; level >= 0; level--) {
    const OrbxLkLevel& I = P.L[level];
    const OrbxLkLevel& J = N.L[level];
    jc.valid = false;
    const float sc = (float)(1.0 / (double)(1 << level));
    float px = __fmul_rn(pp.x, sc), py = __fmul_rn(pp.y, sc);
    float nx, ny;
    if (level == P.top) {
      nx = px;
      ny = py;
    } else {
      nx = __fmul_rn(outx, 2.f);
      ny = __fmul_rn(outy, 2.f);
    }
    outx = nx;
    outy = ny;
    px -= half;
    py -= half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
      if (level == 0) {
        st = 0;
        ev = 0.f;
      }
      continue;
    }
    // template + gradient matrix
    const LkWeights wi = lk_weights(px - (float)ipx, py - (float)ipy);
    const bool in_i = ipx >= 0 && ipy >= 0 && ipx + win + 1 <= I.w && ipy + win + 1 <= I.h;
    int a11 = 0, a12 = 0, a22 = 0;
    LK_ITEM_LOOP {
      const int y = (int)(((uint32_t)idx * rcp) >> 16), x = idx - y * win;
      const int X = ipx + x, Y = ipy + y;
      const int iv = lk_sample(I, X, Y, wi, in_i);
      const short2 d00 = lk_deriv_at(I, X, Y, in_i), d01 = lk_deriv_at(I, X + 1, Y, in_i);
      const short2 d10 = lk_deriv_at(I, X, Y + 1, in_i), d11 = lk_deriv_at(I, X + 1, Y + 1, in_i);
      const int ix = lk_descale(d00.x * wi.w00 + d01.x * wi.w01 + d10.x * wi.w10 + d11.x * wi.w11, 14);
      const int iy = lk_descale(d00.y * wi.w00 + d01.y * wi.w01 + d10.y * wi.w10 + d11.y * wi.w11, 14);
      tpl[idx] = make_short4((short)iv, (short)ix, (short)iy, 0);
      a11 += ix * ix;
      a12 += ix * iy;
      a22 += iy * iy;
    }
    const float A11 = __fmul_rn(lk_exact_sum_f32(a11), flt_scale), A12 = __fmul_rn(lk_exact_sum_f32(a12), flt_scale),
                A22 = __fmul_rn(lk_exact_sum_f32(a22), flt_scale);
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dif = __fsub_rn(A11, A22);
    const float disc = __fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12));
    const float min_eig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), sqrtf(disc)), (float)(2 * win * win));
    if (min_eig < 1e-4f || D < 1.1920928955078125e-07f) {
      if (level == 0) st = 0;
      continue;
    }
    D = __fdiv_rn(1.f, D);
    nx -= half;
    ny -= half;
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < max_iters; j++) {
      const int inx = (int)floorf(nx), iny = (int)floorf(ny);
      if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
        if (level == 0) st = 0;
        break;
      }
      const LkWeights wj = lk_weights(nx - (float)inx, ny - (float)iny);
      lk_cache_cover(jc, J, inx, iny, win, lane);
      int b1 = 0, b2 = 0;
      LK_ITEM_LOOP {
        const int y = (int)(((uint32_t)idx * rcp) >> 16), x = idx - y * win;
        const short4 t = tpl[idx];
        const int diff = lk_sample_cached(jc, inx, iny, x, y, wj) - t.x;
        b1 += diff * t.y;
        b2 += diff * t.z;
      }
      const float B1 = __fmul_rn(lk_exact_sum_f32(b1), flt_scale), B2 = __fmul_rn(lk_exact_sum_f32(b2), flt_scale);
      const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, B2), __fmul_rn(A22, B1)), D);
      const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, B1), __fmul_rn(A11, B2)), D);
      nx = __fadd_rn(nx, dx);
      ny = __fadd_rn(ny, dy);
      outx = __fadd_rn(nx, half);
      outy = __fadd_rn(ny, half);
      if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= eps2) break;
      if (j > 0 && (double)fabsf(__fadd_rn(dx, pdx)) < 0.01 && (double)fabsf(__fadd_rn(dy, pdy)) < 0.01) {
        outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f));
        outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
        break;
      }
      pdx = dx;
      pdy = dy;
    }
    if (st && level == 0) {
      const float ex = __fsub_rn(outx, half), ey = __fsub_rn(outy, half);
      const int inx = (int)floorf(ex), iny = (int)floorf(ey);
      if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
        st = 0;
        continue;
      }
      const LkWeights wj = lk_weights(ex - (float)inx, ey - (float)iny);
      lk_cache_cover(jc, J, inx, iny, win, lane);
      int e = 0;
      LK_ITEM_LOOP {
        const int y = (int)(((uint32_t)idx * rcp) >> 16), x = idx - y * win;
        const int diff = lk_sample_cached(jc, inx, iny, x, y, wj) - tpl[idx].x;
        e += diff < 0 ? -diff : diff;
      }
      ev = __fmul_rn(lk_exact_sum_f32(e), __fdiv_rn(1.f, (float)(32 * win * win)));
    }
  }
  if (lane == 0) {
    next_pts[i] = make_float2(outx, outy);
    status[i] = (uint8_t)st;
    if (err) err[i] = ev;
  }
}

}  // namespace

hipError_t orbx_launch_lk_pyrdown(hipStream_t s, const uint8_t* d_src, int sw, int sh, int spitch, uint8_t* d_dst,
                                  int dw, int dh, int dpitch) {
  dim3 grid((dw + 63) / 64, (dh + 3) / 4);
  hipLaunchKernelGGL(k_lk_pyrdown, grid, dim3(256), 0, s, d_src, sw, sh, spitch, d_dst, dw, dh, dpitch);
  return hipGetLastError();
}

hipError_t orbx_launch_lk_scharr(hipStream_t s, const uint8_t* d_img, int w, int h, int pitch, int16_t* d_deriv) {
  dim3 grid((w + 63) / 64, (h + 3) / 4);
  hipLaunchKernelGGL(k_lk_scharr, grid, dim3(256), 0, s, d_img, w, h, pitch, reinterpret_cast<short2*>(d_deriv));
  return hipGetLastError();
}

hipError_t orbx_launch_lk_track(hipStream_t s, const OrbxLkPyr& prev, const OrbxLkPyr& next, int n,
                                const float* d_prev_pts, float* d_next_pts, uint8_t* d_status, float* d_err, int win,
                                int max_iters, double eps2) {
  if (n <= 0) return hipSuccess;
  if (win < 3 || win > LK_MAX_WIN) return hipErrorInvalidValue;
  const float2* pp = reinterpret_cast<const float2*>(d_prev_pts);
  float2* np_ = reinterpret_cast<float2*>(d_next_pts);
  if (win == 21)
    hipLaunchKernelGGL(k_lk_track<7>, dim3((n + 3) / 4), dim3(256), 0, s, prev, next, n, pp, np_, d_status, d_err, win,
                       max_iters, eps2);
  else
    hipLaunchKernelGGL(k_lk_track<0>, dim3((n + 3) / 4), dim3(256), 0, s, prev, next, n, pp, np_, d_status, d_err, win,
                       max_iters, eps2);
  return hipGetLastError();
}
