// orbx_blur.hip -- separable 5x5 Gaussian blur [1 4 6 4 1]/16 x [1 4 6 4 1]/16, REFLECT_101,
// round-half-even: exactly rne(sum_ij w_i w_j p / 256)  (src/cuda/GaussianBlur1D.cu:34-163; every
// float intermediate there is an exact dyadic, so integer arithmetic reproduces it bit for bit).
// gfx950 only.  Register streaming, NO LDS, NO barriers:
//   * one wavefront owns a 256-pixel-wide column strip (64 lanes x one aligned dword, ALL lanes
//     productive: a wave row is one aligned 256-byte segment for the load and for the store --
//     tools/bw_probe.hip: the 248-byte rows of the previous version cost 29 % of the copy rate
//     from the Infinity Cache and 36 % from HBM) and walks DOWN it over a tall row band (up to
//     ORBX_BLUR3_RH = 64 rows; the host balances the bands of a level), so the 4 warm-up rows of
//     the vertical pass are paid once per band, not once per 16 rows as in the first streaming
//     version (which recomputed 25 % of its horizontal passes);
//   * rows are requested five ahead (one group of buffer loads in flight while the previous group
//     is being worked on); row offsets are scalar, lanes outside the image use the hardware range
//     check instead of exec masks;
//   * left / right neighbour dwords come from the adjacent lanes with DPP wave_shr / wave_shl; the
//     two dwords beyond the strip are fetched by lanes 0 and 63 with one extra (two-lane) load per
//     row and slip in as the DPP "old" value of the lanes that have no neighbour;
//   * horizontal pass on packed 16-bit lanes (v_perm_b32 gathers the shifted byte pairs,
//     v_pk_mad_u16 does two pixels per op); the five most recent H rows live in registers (the
//     loop is unrolled by five so the ring indices are static), so the vertical pass touches no
//     memory; rne(S/256) = high byte of S + 127 + ((S >> 8) & 1) on the packed pair; one v_perm
//     packs the four result bytes, one dword store per lane per row.
//   REFLECT_101: rows by (scalar) index; columns by per-lane v_perm selectors, applied only in the
//   waves that hold x = 0 or the image's last pixel.
#include <hip/hip_runtime.h>

#include "orbx_internal.h"
#include "orbx_wave.h"

namespace {

// a * k + c on both 16-bit lanes with the packed multiplier in a register the optimiser cannot see
// through (pk_opaque): a literal 4 would be strength-reduced to shift + add, two instructions
// where v_pk_mad_u16 is one
__device__ __forceinline__ uint32_t pk_mad_r(uint32_t a, uint32_t k, uint32_t c) {
  return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) * __builtin_bit_cast(us2_t, k) +
                                              __builtin_bit_cast(us2_t, c)));
}
__device__ __forceinline__ uint32_t pk_opaque(uint32_t k) {
  asm volatile("" : "+s"(k));
  return k;
}
// round-half-even of S/256 on both 16-bit lanes: the result is the HIGH byte of each lane of
// S + 127 + ((S >> 8) & 1)  (S <= 255 * 256 per lane, so the sums never carry into the
// neighbouring lane and plain 32-bit operations do: shift, and, one three-input add)
__device__ __forceinline__ uint32_t pk_rne8_hi(uint32_t S) { return S + 0x007f007fu + ((S >> 8) & 0x00010001u); }

// wave-uniform REFLECT_101 on scalars (row index); valid for -len < p < 2 * len - 1
__device__ __forceinline__ int reflect101_s(int p, int len) {
  p = p < 0 ? -p : p;
  return p >= len ? 2 * len - p - 2 : p;
}

struct Blur3Strip {
  __amdgpu_buffer_rsrc_t rin, rout;
  uint32_t voff, voff_halo;       // lane x; lane 0: x - 4, lane 63: x + 4 (or out of range)
  uint32_t selL, selC, selR;      // column REFLECT_101 selectors (identity except in the border lanes)
  uint32_t vmask;                 // bytes of this lane's dword that are inside the image
  int y0, yend, h, pitch;
};

// horizontal pass of one incoming row -> the packed sums of the even (0,2) and odd (1,3) pixels
template <bool PATCH>
__device__ __forceinline__ void blur3_h(const Blur3Strip& S, uint32_t C0, uint32_t H, uint32_t k4, uint32_t k6,
                                        uint32_t& he, uint32_t& ho) {
  // bound_ctrl off: a lane without a source lane (0 under wave_shr, 63 under wave_shl) keeps `old` = its halo dword
  const uint32_t Ld = __builtin_amdgcn_update_dpp(H, C0, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
  const uint32_t Lw = PATCH ? __builtin_amdgcn_perm(C0, Ld, S.selL) : Ld;
  const uint32_t C = PATCH ? __builtin_amdgcn_perm(C0, Lw, S.selC) : C0;
  const uint32_t Rd = __builtin_amdgcn_update_dpp(H, C, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
  const uint32_t Rw = PATCH ? __builtin_amdgcn_perm(C0, Rd, S.selR) : Rd;
  // byte pairs (16-bit lanes): perm bytes 0-3 = 2nd argument, 4-7 = 1st
  const uint32_t A = __builtin_amdgcn_perm(C, Lw, 0x0c040c02u);   // (L.b2, C.b0)
  const uint32_t B = __builtin_amdgcn_perm(C, Lw, 0x0c050c03u);   // (L.b3, C.b1)
  const uint32_t Cc = __builtin_amdgcn_perm(C, C, 0x0c020c00u);   // (C.b0, C.b2)
  const uint32_t D = __builtin_amdgcn_perm(C, C, 0x0c030c01u);    // (C.b1, C.b3)
  const uint32_t E = __builtin_amdgcn_perm(Rw, C, 0x0c040c02u);   // (C.b2, R.b0)
  const uint32_t F = __builtin_amdgcn_perm(Rw, C, 0x0c050c03u);   // (C.b3, R.b1)
  he = pk_mad_r(pk_add(B, D), k4, pk_mad_r(Cc, k6, pk_add(A, E)));   // pixels 0,2
  ho = pk_mad_r(pk_add(Cc, E), k4, pk_mad_r(D, k6, pk_add(B, F)));   // pixels 1,3
}

// vertical pass over the ring slots of rows y-2 .. y+2 (a = oldest) and the store of output row y
template <bool PATCH>
__device__ __forceinline__ void blur3_v(const Blur3Strip& S, const uint32_t (&he)[5], const uint32_t (&ho)[5], int a,
                                        int b, int c, int d, int e, uint32_t k4, uint32_t k6, int y) {
  const uint32_t te = pk_rne8_hi(pk_mad_r(pk_add(he[b], he[d]), k4, pk_mad_r(he[c], k6, pk_add(he[a], he[e]))));
  const uint32_t to = pk_rne8_hi(pk_mad_r(pk_add(ho[b], ho[d]), k4, pk_mad_r(ho[c], k6, pk_add(ho[a], ho[e]))));
  // bytes: px0 = te lane0 high byte, px1 = to lane0 high, px2 = te lane1 high, px3 = to lane1 high
  const uint32_t out = __builtin_amdgcn_perm(to, te, 0x07030501u);
  // (interior waves lie wholly left of the image's last dword: nothing to mask)
  if (y < S.yend) __builtin_amdgcn_raw_buffer_store_b32(PATCH ? (out & S.vmask) : out, S.rout, S.voff, y * S.pitch, 0);
}

template <bool PATCH>
__device__ __forceinline__ void blur3_strip(const Blur3Strip& S) {
  const uint32_t k4 = pk_opaque(0x00040004u), k6 = pk_opaque(0x00060006u);
  uint32_t he[5], ho[5], raw[5], nxt[5], rawh[5], nxth[5];
  // input row r of the strip is image row y0 - 2 + r; output row y0 + r - 4 is complete after row r
  auto load = [&](int r, uint32_t& c, uint32_t& hh) {
    const int so = reflect101_s(S.y0 - 2 + r, S.h) * S.pitch;
    c = __builtin_amdgcn_raw_buffer_load_b32(S.rin, S.voff, so, 0);
    hh = __builtin_amdgcn_raw_buffer_load_b32(S.rin, S.voff_halo, so, 0);
  };
  const int nr = S.yend - S.y0 + 4;  // input rows
#pragma unroll
  for (int k = 0; k < 5; k++) load(k, raw[k], rawh[k]);
#pragma unroll
  for (int k = 0; k < 5; k++) load(min(5 + k, nr + 1), nxt[k], nxth[k]);
  // first group: four warm-up rows, the fifth completes output row y0
#pragma unroll
  for (int k = 0; k < 5; k++) blur3_h<PATCH>(S, raw[k], rawh[k], k4, k6, he[k], ho[k]);
  blur3_v<PATCH>(S, he, ho, 0, 1, 2, 3, 4, k4, k6, S.y0);
  for (int rb = 5; rb < nr; rb += 5) {  // wave-uniform
#pragma unroll
    for (int k = 0; k < 5; k++) {
      raw[k] = nxt[k];
      rawh[k] = nxth[k];
    }
    // rows past the strip (last group) are loaded from reflected / clamped indices and never stored
#pragma unroll
    for (int k = 0; k < 5; k++) load(min(rb + 5 + k, nr + 1), nxt[k], nxth[k]);
    const int y = S.y0 + rb - 4;
    blur3_h<PATCH>(S, raw[0], rawh[0], k4, k6, he[0], ho[0]);
    blur3_v<PATCH>(S, he, ho, 1, 2, 3, 4, 0, k4, k6, y);
    blur3_h<PATCH>(S, raw[1], rawh[1], k4, k6, he[1], ho[1]);
    blur3_v<PATCH>(S, he, ho, 2, 3, 4, 0, 1, k4, k6, y + 1);
    blur3_h<PATCH>(S, raw[2], rawh[2], k4, k6, he[2], ho[2]);
    blur3_v<PATCH>(S, he, ho, 3, 4, 0, 1, 2, k4, k6, y + 2);
    blur3_h<PATCH>(S, raw[3], rawh[3], k4, k6, he[3], ho[3]);
    blur3_v<PATCH>(S, he, ho, 4, 0, 1, 2, 3, k4, k6, y + 3);
    blur3_h<PATCH>(S, raw[4], rawh[4], k4, k6, he[4], ho[4]);
    blur3_v<PATCH>(S, he, ho, 0, 1, 2, 3, 4, k4, k6, y + 4);
  }
}

// grid = (strip table entries of ONE frame / 4, frames); wave w of a workgroup owns entry 4 * blockIdx.x + w:
// level `l`, strip `tx` (256 px), rows [ty, ty + f).
__global__ __launch_bounds__(256) void k_blur3(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int frame_bytes,
                                               const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                               int first_level) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: row math runs on the SALU
  const int ti = blockIdx.x * 4 + wave;
  if (ti >= n_tiles) return;  // whole wave
  const OrbxTileDesc d = tiles[ti];  // one scalar load instead of decoding through the plan
  const int w = d.w, h = d.h, pitch = d.pitch;
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int x = d.tx * ORBX_BLUR3_TW + lane * 4;
  Blur3Strip S;
  S.y0 = d.ty;
  S.yend = min(d.ty + d.f, h);
  S.h = h;
  S.pitch = pitch;
  if (S.y0 >= S.yend) return;
  // Buffer descriptors over the level image: the (scalar) row base goes in the scalar offset, the
  // lane's x in the vector offset; lanes whose x is outside [0, pitch) use an out-of-range vector
  // offset, so the hardware range check zero-fills their loads and drops their stores -- no
  // exec-mask juggling, no 64-bit per-lane address arithmetic.
  const size_t level_off = (size_t)f * frame_bytes + d.img_off;
  S.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) + level_off, 0, pitch * h, 0x00020000);
  S.rout = __builtin_amdgcn_make_buffer_rsrc(dst + level_off, 0, pitch * h, 0x00020000);
  S.voff = x < pitch ? (uint32_t)x : 0xffffffffu;
  const int xh = lane == 0 ? x - 4 : x + 4;  // the dwords beyond the strip, fetched by lanes 0 and 63
  S.voff_halo = ((lane == 0 || lane == 63) && xh >= 0 && xh < pitch) ? (uint32_t)xh : 0xffffffffu;

  if (d.l < first_level) {  // pass-through copy
    for (int y = S.y0; y < S.yend; y++) {
      const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(S.rin, S.voff, y * pitch, 0);
      __builtin_amdgcn_raw_buffer_store_b32(v, S.rout, S.voff, y * pitch, 0);
    }
    return;
  }

  // column REFLECT_101 as per-lane v_perm selectors (identity except in the lane holding x = 0
  // and the lane holding the image's last pixel)
  const int e4 = (w - 1) & ~3;    // x of the dword holding the last pixel
  const int rbyte = (w - 1) & 3;  // its position inside that dword
  const bool edge = (x == e4);
  // Lw = perm(C, Ldpp, selL): bytes 0-3 = Ldpp, 4-7 = C.   x=-1 <- x=1, x=-2 <- x=2
  S.selL = x == 0 ? 0x05060c0cu : 0x03020100u;
  // C' = perm(C, Lw, selC): bytes 0-3 = Lw, 4-7 = C
  const uint32_t selC_e = rbyte == 3 ? 0x07060504u : rbyte == 2 ? 0x05060504u : rbyte == 1 ? 0x03040504u : 0x07020304u;
  S.selC = edge ? selC_e : 0x07060504u;
  // Rw = perm(C0, Rdpp, selR): bytes 0-3 = Rdpp, 4-7 = C0 (the unpatched edge dword)
  const uint32_t selR_e = rbyte == 3 ? 0x0c0c0506u : rbyte == 2 ? 0x0c0c0c04u : 0x0c0c0c0cu;
  // lane 63 fetches its right neighbour from memory, unpatched: if that dword holds only the image's
  // last pixel (rbyte == 0), its byte 1 (x = w) is the reflection of x = w - 2, this lane's own byte 3
  const bool edge_right = lane == 63 && x + 4 == e4 && rbyte == 0;
  S.selR = edge ? selR_e : edge_right ? 0x0c0c0700u : 0x03020100u;
  // output bytes at x >= w are written as zero (padding stays zero)
  const int nvalid = w - x;
  S.vmask = nvalid >= 4 ? 0xffffffffu : nvalid <= 0 ? 0u : ((1u << (8 * nvalid)) - 1u);

  // only the waves that hold x = 0 or the image's last pixel need the column patches
  const int x_lo = d.tx * ORBX_BLUR3_TW, x_hi = x_lo + 4 * 63;
  const bool patch = (x_lo <= 0) || (e4 >= x_lo && e4 <= x_hi) || (e4 == x_hi + 4 && rbyte == 0);  // wave-uniform
  if (patch)
    blur3_strip<true>(S);
  else
    blur3_strip<false>(S);
}

}  // namespace

// separable blur of every level >= first_level (lower levels are copied); d_tiles: the strip table
// of ONE frame (orbx_api.cpp: build_blur_tiles)
hipError_t orbx_launch_blur3(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int n_frames,
                             const uint8_t* d_src, uint8_t* d_dst, int first_level) {
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  // four waves per workgroup (measured: 1 -> 53 us, 2 -> 48 us, 4 -> 43 us, 8 / 16 -> 46 us per 64-frame batch)
  dim3 grid((n_tiles + 3) / 4, n_frames);
  hipLaunchKernelGGL(k_blur3, grid, dim3(256), 0, s, d_tiles, n_tiles, frame_bytes, d_src, d_dst, first_level);
  return hipGetLastError();
}
