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

// ---------------------------------------------------------------------------------------------
// The same blur with SIXTEEN pixels per lane (k_blur4; the stand-alone blur of the stage operator and of the
// blur_levels = none / upper modes).  tools/bw_probe.hip: a wave row of 64 x 4 bytes reads at 3.8 TB/s and copies at
// 4.7, 64 x 16 bytes at 6.2 / 5.1 -- and k_blur3 moved its traffic at exactly the 4-byte copy rate.  A 1024-pixel
// wave row would waste most lanes on this path's level widths (1241, 1034, 862, ... : 61 % of the lanes at level 0),
// so the wave is FOUR row bands of 16 lanes: lane = 16 * band + column, a lane owns 16 pixels (one dwordx4 per row
// for the load and for the store), a wave 256 pixels x 4 bands of the same strip.  The neighbours of the horizontal
// pass are the lane's own dwords except at its two ends: DPP row_shr / row_shl (rows of 16 lanes = the bands) -- 2
// DPP + 18 v_perm per 16 pixels where four 4-pixel lanes need 8 + 24 --, the two dwords beyond a band's strip
// arrive with one extra 8-lane load per row.  Bands are tall (a level's height / 4, at most ORBX_BLUR4_RH rows), so
// the 4 warm-up rows of the vertical pass are 4 % of a KITTI level instead of 6.
struct Blur4Lane {
  __amdgpu_buffer_rsrc_t rin, rout;
  uint32_t xoff, xhalo;  // byte offset of the lane's 16 pixels / of its halo dword (or out of range)
  int y0, yend, h, pitch;
  uint32_t selL[4], selC[4], selR[4], vmask[4];  // column REFLECT_101 per dword (PATCH waves only)
};

template <bool PATCH>
__device__ __forceinline__ void blur4_rows(const Blur4Lane& S, int nr) {
  const uint32_t k4 = pk_opaque(0x00040004u), k6 = pk_opaque(0x00060006u);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t he[5][4], ho[5][4];
  u32x4 raw[5];
  uint32_t rawh[5];
  // input row r of a band is level row reflect(y0 - 2 + r) (per lane: the bands of a wave start at different rows)
  auto load = [&](int r, u32x4& c, uint32_t& hh) {
    int y = S.y0 - 2 + r;
    y = max(y, -y);
    y = min(y, 2 * S.h - 2 - y);
    const uint32_t ro = (uint32_t)(y * S.pitch);
    c = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(S.rin, ro + S.xoff, 0, 0));
    hh = __builtin_amdgcn_raw_buffer_load_b32(S.rin, ro + S.xhalo, 0, 0);
  };
  auto hpass = [&](int k) {
    const u32x4 c = raw[k];
    const uint32_t C0[4] = {c.x, c.y, c.z, c.w};
    // rows of 16 lanes = bands: the lane without a source lane (column 0 / 15) keeps `old` = its halo dword
    const uint32_t Ld = __builtin_amdgcn_update_dpp(rawh[k], C0[3], 0x111 /*row_shr:1*/, 0xf, 0xf, false);
    const uint32_t Rd = __builtin_amdgcn_update_dpp(rawh[k], C0[0], 0x101 /*row_shl:1*/, 0xf, 0xf, false);
    if (PATCH) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t Lraw = j == 0 ? Ld : C0[j - 1], Rraw = j == 3 ? Rd : C0[j + 1];
        const uint32_t Lw = __builtin_amdgcn_perm(C0[j], Lraw, S.selL[j]);
        const uint32_t C = __builtin_amdgcn_perm(C0[j], Lw, S.selC[j]);
        const uint32_t Rw = __builtin_amdgcn_perm(C0[j], Rraw, S.selR[j]);
        const uint32_t A = __builtin_amdgcn_perm(C, Lw, 0x0c040c02u), B = __builtin_amdgcn_perm(C, Lw, 0x0c050c03u);
        const uint32_t Cc = __builtin_amdgcn_perm(C, C, 0x0c020c00u), D = __builtin_amdgcn_perm(C, C, 0x0c030c01u);
        const uint32_t E = __builtin_amdgcn_perm(Rw, C, 0x0c040c02u), F = __builtin_amdgcn_perm(Rw, C, 0x0c050c03u);
        he[k][j] = pk_mad_r(pk_add(B, D), k4, pk_mad_r(Cc, k6, pk_add(A, E)));
        ho[k][j] = pk_mad_r(pk_add(Cc, E), k4, pk_mad_r(D, k6, pk_add(B, F)));
      }
    } else {
      uint32_t A = __builtin_amdgcn_perm(C0[0], Ld, 0x0c040c02u), B = __builtin_amdgcn_perm(C0[0], Ld, 0x0c050c03u);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t C = C0[j], Rw = j == 3 ? Rd : C0[j + 1];
        const uint32_t Cc = __builtin_amdgcn_perm(C, C, 0x0c020c00u), D = __builtin_amdgcn_perm(C, C, 0x0c030c01u);
        const uint32_t E = __builtin_amdgcn_perm(Rw, C, 0x0c040c02u), F = __builtin_amdgcn_perm(Rw, C, 0x0c050c03u);
        he[k][j] = pk_mad_r(pk_add(B, D), k4, pk_mad_r(Cc, k6, pk_add(A, E)));
        ho[k][j] = pk_mad_r(pk_add(Cc, E), k4, pk_mad_r(D, k6, pk_add(B, F)));
        A = E;  // (x+2, x+4) and (x+3, x+5) of this dword are (x-2, x) and (x-1, x+1) of the next one
        B = F;
      }
    }
  };
  // vertical pass over ring slots a..e (a = oldest) -> output row y of every band (y is per lane)
  auto vpass = [&](int a, int b, int c, int d, int e, int rel) {
    const int y = S.y0 + rel;
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t te = pk_rne8_hi(pk_mad_r(pk_add(he[b][j], he[d][j]), k4, pk_mad_r(he[c][j], k6, pk_add(he[a][j], he[e][j]))));
      const uint32_t to = pk_rne8_hi(pk_mad_r(pk_add(ho[b][j], ho[d][j]), k4, pk_mad_r(ho[c][j], k6, pk_add(ho[a][j], ho[e][j]))));
      o[j] = __builtin_amdgcn_perm(to, te, 0x07030501u);
      if (PATCH) o[j] &= S.vmask[j];
    }
    const uint32_t vo = y < S.yend ? (uint32_t)(y * S.pitch) + S.xoff : 0x7fffff00u;  // (out of range: dropped)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned,
                                                              u32x4{o[0], o[1], o[2], o[3]}),
                                           S.rout, vo, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < 5; k++) load(k, raw[k], rawh[k]);
  // first group: four warm-up rows, the fifth completes output row y0; each row's registers are refilled with the
  // row five further down as soon as its horizontal pass has consumed them
#pragma unroll
  for (int k = 0; k < 5; k++) {
    hpass(k);
    load(min(5 + k, nr + 1), raw[k], rawh[k]);
  }
  vpass(0, 1, 2, 3, 4, 0);
  for (int rb = 5; rb < nr; rb += 5) {  // wave-uniform
    const int rel = rb - 4;
    hpass(0);
    load(min(rb + 5, nr + 1), raw[0], rawh[0]);
    vpass(1, 2, 3, 4, 0, rel);
    hpass(1);
    load(min(rb + 6, nr + 1), raw[1], rawh[1]);
    vpass(2, 3, 4, 0, 1, rel + 1);
    hpass(2);
    load(min(rb + 7, nr + 1), raw[2], rawh[2]);
    vpass(3, 4, 0, 1, 2, rel + 2);
    hpass(3);
    load(min(rb + 8, nr + 1), raw[3], rawh[3]);
    vpass(4, 0, 1, 2, 3, rel + 3);
    hpass(4);
    load(min(rb + 9, nr + 1), raw[4], rawh[4]);
    vpass(0, 1, 2, 3, 4, rel + 4);
  }
}

// grid = (table entries of ONE frame / 4, frames); wave w of a workgroup owns entry 4 * blockIdx.x + w: level `l`,
// strip `tx` (256 px), four bands of `f` rows starting at row ty.
__global__ __launch_bounds__(256) void k_blur4(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int frame_bytes,
                                               const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                               int first_level) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = blockIdx.x * 4 + wave;
  if (ti >= n_tiles) return;  // whole wave
  const OrbxTileDesc d = tiles[ti];
  const int w = d.w, h = d.h, pitch = d.pitch, rows = d.f;
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63, band = lane >> 4, col = lane & 15;
  const int x = d.tx * ORBX_BLUR4_TW + 16 * col;
  Blur4Lane S;
  S.h = h;
  S.pitch = pitch;
  S.y0 = d.ty + band * rows;
  S.yend = min(S.y0 + rows, h);
  const bool live = x < pitch && S.y0 < h;  // (a band past the level's last row, a lane past its pitch: nothing to do)
  if (!live) S.y0 = 0, S.yend = 0;
  const size_t level_off = (size_t)f * frame_bytes + d.img_off;
  S.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) + level_off, 0, pitch * h, 0x00020000);
  S.rout = __builtin_amdgcn_make_buffer_rsrc(dst + level_off, 0, pitch * h, 0x00020000);
  constexpr uint32_t OUT = 0x7fffff00u;  // with any row offset added still far beyond a level: loads give 0, stores are dropped
  S.xoff = live ? (uint32_t)x : OUT;
  const int xh = col == 0 ? x - 4 : x + 16;
  S.xhalo = (live && (col == 0 || col == 15) && xh >= 0 && xh < pitch) ? (uint32_t)xh : OUT;

  if (d.l < first_level) {  // pass-through copy
    for (int r = 0; r < rows; r++) {  // wave-uniform trip count; lanes past their band address out of range
      const int y = S.y0 + r;
      const uint32_t o = y < S.yend ? (uint32_t)(y * pitch) + S.xoff : OUT;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(S.rin, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v, S.rout, o, 0, 0);
    }
    return;
  }
  const int e4 = (w - 1) & ~3, rbyte = (w - 1) & 3;
  const int x_lo = d.tx * ORBX_BLUR4_TW, x_hi = x_lo + ORBX_BLUR4_TW - 4;
  // only the waves that hold x = 0, the image's last pixels or padding need the column patches (wave-uniform)
  const bool patch = x_lo == 0 || x_hi + 4 >= e4;
  const int nr = rows + 4;  // input rows of the tallest band
  if (patch) {
    const uint32_t selC_e = rbyte == 3 ? 0x07060504u : rbyte == 2 ? 0x05060504u : rbyte == 1 ? 0x03040504u : 0x07020304u;
    const uint32_t selR_e = rbyte == 3 ? 0x0c0c0506u : rbyte == 2 ? 0x0c0c0c04u : 0x0c0c0c0cu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int xj = x + 4 * j;
      const bool edge = xj == e4;
      // (neighbour dwords are taken RAW here, so the dword left of an edge dword that holds one pixel builds x = w
      // -- the reflection of its own last byte -- itself, whichever lane or dword the edge is in)
      S.selL[j] = xj == 0 ? 0x05060c0cu : 0x03020100u;
      S.selC[j] = edge ? selC_e : 0x07060504u;
      S.selR[j] = edge ? selR_e : (xj + 4 == e4 && rbyte == 0) ? 0x0c0c0700u : 0x03020100u;
      const int nvalid = w - xj;
      S.vmask[j] = nvalid >= 4 ? 0xffffffffu : nvalid <= 0 ? 0u : ((1u << (8 * nvalid)) - 1u);
    }
    blur4_rows<true>(S, nr);
  } else {
    blur4_rows<false>(S, nr);
  }
}

// ---------------------------------------------------------------------------------------------
// Pyramid + blur in ONE pass (whole path, blur on every level): the wave produces each row of its
// strip of level l straight from the input frame -- level 0: the frame's own pixels; level >= 1:
// OpenCV's 8UC1 fixed-point bilinear resize of level 0 (src/orb.cpp:111-120; the arithmetic of
// k_pyramid2, orbx_kernels.hip) -- and feeds it to the streaming blur above.  The un-blurred
// pyramid is never written nor read back: 2 x 1.5 MB of traffic per frame and one launch less.
// The halo dwords of a strip have to be COMPUTED here, not loaded, so this kernel keeps the
// 248-pixel strips (lanes 1..62 productive, lanes 0 / 63 compute the halo and store nothing; at the
// image borders the halo is a reflection and those lanes are productive as well).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
struct __attribute__((packed, aligned(1))) uint2_unaligned {
  uint32_t x, y;
};

struct PyrSrc {
  __amdgpu_buffer_rsrc_t rsrc;  // the input frame (= level 0), exactly its bytes
  const uint8_t* src;
  int in_stride, w0, h0;
  int x, w, h;                  // lane x, level size
  // y taps of the strip's input rows: lane r holds the tap of input row r (level row reflect(y0 - 2 + r)),
  // fetched once per strip with ONE vector load and handed out with v_readlane -- a scalar load per row
  // (plus its wait) sat in front of every row's gathers
  int yt_ofs;
  uint32_t yt_cc;               // c0 | c1 << 16
  int n8;                       // staged mode: 8-byte chunks of a source row the strip needs (<= 128)
};
struct PyrLane {  // x taps of the lane's four pixels
  uint32_t ofs[4], cc[4], sel[4], base;
  uint32_t lds_a[4];  // staged mode: dword-aligned byte offset of pixel k's pair within a staged source row
};
// staged mode: per wave two source rows of ORBX_PYR_STAGE_BYTES (+ 8: the pair of the row's last byte is read as
// two dwords)
#define PYR_STAGE_ROW (ORBX_PYR_STAGE_BYTES + 8)

// MODE 0: level 0 (one unaligned dword of the frame); 1: resize through one 8-byte window per source
// row (scale <= 2, host-verified per level); 2: resize with four 2-byte pair gathers per source row; 3: the same
// pairs read from LDS, where the strip's span of the two source rows has been STAGED with four coalesced 8-byte
// loads per output row (scale <= ~3.2, host-verified: OrbxLevel::win8 == 3).  The gathers of mode 2 are 8
// vector-memory instructions with 64 scattered addresses each per output row, and they, not the arithmetic, were
// what its strips took 1.5x the window mode's time per pixel for (with a quarter of them: the window mode's time)
template <int MODE>
struct PyrRaw {
  uint32_t q0[MODE == 0 ? 1 : MODE == 1 ? 2 : 4], q1[MODE == 0 ? 1 : MODE == 1 ? 2 : 4];
  int b0, b1;
};

// issue the loads of input row r of the strip (level row yl = reflect(y0 - 2 + r)).  The row offsets are
// scalars (buffer loads: frame descriptor + scalar row offset + the lane's constant byte offset), no
// 64-bit per-lane address arithmetic.
template <int MODE>
__device__ __forceinline__ void pyr_issue(const PyrSrc& P, const PyrLane& T, int r, int yl, PyrRaw<MODE>& R) {
  if (MODE == 0) {
    const int nvalid = P.w - P.x;
    uint32_t v = 0;
    if (P.x >= 0 && (nvalid >= 4 || (nvalid > 0 && yl + 1 < P.h))) {
      // (a partial last dword of a row above the last one runs into the next row of the same
      // frame: readable, the blur's column patch / byte mask ignores those bytes)
      v = __builtin_amdgcn_raw_buffer_load_b32(P.rsrc, (uint32_t)P.x, yl * P.in_stride, 0);
    } else if (P.x >= 0 && nvalid > 0) {  // last dword of the last row: never read past the frame
      const uint8_t* row = P.src + (size_t)yl * P.in_stride;
      for (int k = 0; k < nvalid; k++) v |= (uint32_t)row[P.x + k] << (8 * k);
    }
    R.q0[0] = v;
    R.b0 = R.b1 = 0;
    return;
  }
  const int tofs = __builtin_amdgcn_readlane(P.yt_ofs, r);
  const uint32_t tcc = (uint32_t)__builtin_amdgcn_readlane((int)P.yt_cc, r);
  const int sy0 = min(max(tofs, 0), P.h0 - 1), sy1 = min(max(tofs + 1, 0), P.h0 - 1);
  const int so0 = sy0 * P.in_stride, so1 = sy1 * P.in_stride;
  R.b0 = (int)(tcc & 0xffffu);
  R.b1 = (int)(tcc >> 16);
  if (MODE == 1) {
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    const v2u a = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base, so0, 0));
    const v2u b = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base, so1, 0));
    R.q0[0] = a.x;
    R.q0[1] = a.y;
    R.q1[0] = b.x;
    R.q1[1] = b.y;
  } else if (MODE == 3) {
    // lane i: bytes [8 i, 8 i + 8) and [512 + 8 i, 512 + 8 i + 8) of the strip's span of both source rows, the
    // lanes past the span's end switched off (the texture addresser takes a cycle per four lane-dwords, coalesced
    // or not: what these rows cost is the number of active lanes x dwords); their registers keep the last row's
    // values and land in a part of the staging row nobody reads
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    if (lane < P.n8) {
      const v2u a = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base, so0, 0));
      const v2u b = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base, so1, 0));
      R.q0[0] = a.x;
      R.q0[MODE == 3 ? 1 : 0] = a.y;
      R.q1[0] = b.x;
      R.q1[MODE == 3 ? 1 : 0] = b.y;
    }
    if (P.n8 > 64) {  // wave-uniform
      if (lane + 64 < P.n8) {
        const v2u a2 = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base + 512u, so0, 0));
        const v2u b2 = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(P.rsrc, T.base + 512u, so1, 0));
        R.q0[MODE == 3 ? 2 : 0] = a2.x;
        R.q0[MODE == 3 ? 3 : 0] = a2.y;
        R.q1[MODE == 3 ? 2 : 0] = b2.x;
        R.q1[MODE == 3 ? 3 : 0] = b2.y;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < (MODE == 2 ? 4 : 0); k++) {
      // ofs <= w0-2 always (host table): one unaligned 16-bit load fetches src[ofs], src[ofs+1]
      R.q0[k] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(P.rsrc, T.ofs[k], so0, 0);
      R.q1[k] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(P.rsrc, T.ofs[k], so1, 0);
    }
  }
}

// the row's dword of four level pixels (OpenCV: ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2)
struct __attribute__((packed, aligned(4))) uint2_a4 {
  uint32_t x, y;
};
// staged mode: the loaded spans of the two source rows -> the wave's staging buffer (wave-private: no barrier)
__device__ __forceinline__ void pyr_stage(const PyrSrc& P, const PyrRaw<3>& R, uint8_t* stage) {
  const int lane8 = (threadIdx.x & 63) * 8;
  *reinterpret_cast<uint2*>(stage + lane8) = make_uint2(R.q0[0], R.q0[1]);
  *reinterpret_cast<uint2*>(stage + PYR_STAGE_ROW + lane8) = make_uint2(R.q1[0], R.q1[1]);
  if (P.n8 > 64) {  // wave-uniform
    if ((int)(threadIdx.x & 63) + 64 < P.n8) {  // (the staging row ends with the longest span: ORBX_PYR_STAGE_BYTES)
      *reinterpret_cast<uint2*>(stage + 512 + lane8) = make_uint2(R.q0[2], R.q0[3]);
      *reinterpret_cast<uint2*>(stage + PYR_STAGE_ROW + 512 + lane8) = make_uint2(R.q1[2], R.q1[3]);
    }
  }
  wave_lds_sync();
}
template <int MODE>
__device__ __forceinline__ uint32_t pyr_finish(const PyrLane& T, const PyrRaw<MODE>& R, uint8_t* stage = nullptr) {
  if (MODE == 0) return R.q0[0];
  uint32_t out = 0;
  const uint32_t bs0 = ((uint32_t)R.b0 << 12) & 0xffffffu, bs1 = ((uint32_t)R.b1 << 12) & 0xffffffu;  // b <= 2048
  uint32_t selp = T.sel[0];
  if (MODE == 3) asm volatile("" : "+v"(selp));  // (opaque: or the unpacking is hoisted out of the row loop, into four registers)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t p0, p1;  // the pixel pair as two u16 lanes
    if (MODE == 1) {
      p0 = __builtin_amdgcn_perm(R.q0[1], R.q0[0], T.sel[k]);
      p1 = __builtin_amdgcn_perm(R.q1[1], R.q1[0], T.sel[k]);
    } else if (MODE == 3) {  // the pair's two aligned dwords of each staged row, the pair cut out by v_perm
      // (an unaligned 16-bit LDS read per pair is legal on gfx950 and took 4x the LDS time)
      // (the four selectors of a lane travel as 4-bit fields of ONE register, sel_k = 0x0c000c00 | ((sel[0] >> 4 k) &
      // 0x00070003): the kernel has no registers to spare, and spilling them costs more than the unpacking)
      const uint2_a4 v0 = *reinterpret_cast<const uint2_a4*>(stage + T.lds_a[k]);
      const uint2_a4 v1 = *reinterpret_cast<const uint2_a4*>(stage + PYR_STAGE_ROW + T.lds_a[k]);
      const uint32_t sel = ((selp >> (4 * k)) & 0x00070003u) | 0x0c000c00u;
      p0 = __builtin_amdgcn_perm(v0.y, v0.x, sel);
      p1 = __builtin_amdgcn_perm(v1.y, v1.x, sel);
    } else {
      p0 = __builtin_amdgcn_perm(R.q0[MODE == 2 ? k : 0], R.q0[MODE == 2 ? k : 0], 0x0c010c00u);
      p1 = __builtin_amdgcn_perm(R.q1[MODE == 2 ? k : 0], R.q1[MODE == 2 ? k : 0], 0x0c010c00u);
    }
    // horizontal pass: src[ofs] * c0 + src[ofs+1] * c1 is one v_dot2_u32_u16 of the pixel pair
    // with the tap's packed (c0, c1)
    const us2_t cw = __builtin_bit_cast(us2_t, T.cc[k]);
    const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, p0), cw, 0u, false);
    const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, p1), cw, 0u, false);
    // vertical pass: (b * (r >> 4)) >> 16 == ((r & ~15) * (b << 12)) >> 32 with both factors below
    // 2^24: one v_and + one full-rate v_mul_hi_u32_u24 per term; the sum is <= 1022
    const uint32_t t0 = (uint32_t)(((u64)(r0 & 0xfffff0u) * (u64)bs0) >> 32);
    const uint32_t t1 = (uint32_t)(((u64)(r1 & 0xfffff0u) * (u64)bs1) >> 32);
    out |= ((t0 + t1 + 2u) >> 2) << (8 * k);
  }
  return out;
}

template <int MODE, bool PATCH>
__device__ __forceinline__ void pyrblur_strip(const Blur3Strip& S, const PyrSrc& P, const PyrLane& T,
                                              uint8_t* stage = nullptr) {
  // Loads in flight per wave: the five rows of a group (modes 0 and 1: <= 4 registers per row), or the
  // next row only (pair-gather mode: 8 registers per row).  With <= 64 registers the kernel keeps 8
  // waves per SIMD, i.e. ALL waves of a 64-frame KITTI batch are resident at once (7.4 per SIMD): at
  // 6 per SIMD the second, partly filled round of waves cost 40 % (100 us instead of 70).
  // (staged mode: the next row only as well, and its registers are free again once they are in LDS)
  constexpr bool ROWWISE = MODE == 2 || MODE == 3;
  constexpr int NB = MODE == 3 ? 1 : MODE == 2 ? 2 : 5;
  const uint32_t k4 = pk_opaque(0x00040004u), k6 = pk_opaque(0x00060006u);
  uint32_t he[5], ho[5];
  PyrRaw<MODE> raw[NB];
  if (MODE == 3) {  // (the switched-off lanes of pyr_issue never write theirs)
#pragma unroll
    for (int k = 0; k < (MODE == 3 ? 4 : 0); k++) raw[0].q0[k] = raw[0].q1[k] = 0;
  }
  // input row r of the strip is level row y0 - 2 + r; output row y0 + r - 4 is complete after row r
  const int nr = S.yend - S.y0 + 4;  // input rows
  // (rows past the strip, in the last group, are computed from clamped indices and never stored)
  auto issue = [&](int r, PyrRaw<MODE>& R) {
    const int rr = min(r, nr + 1);
    pyr_issue<MODE>(P, T, rr, reflect101_s(S.y0 - 2 + rr, S.h), R);
  };
  auto hpass = [&](int rb, int k) {
    if (MODE == 3) {
      // row rb + k: registers -> LDS, the next row's loads into the same registers, this row's pairs from LDS.
      // (ONE staging buffer per wave: a wave's LDS operations execute in order, the next row's writes cannot
      // overtake this row's reads)
      PyrRaw<MODE> cur;
      cur.b0 = raw[0].b0;
      cur.b1 = raw[0].b1;
      if constexpr (MODE == 3) pyr_stage(P, raw[0], stage);
      if (k < 4) issue(rb + k + 1, raw[0]);
      blur3_h<PATCH>(S, pyr_finish<MODE>(T, cur, stage), 0u, k4, k6, he[k], ho[k]);
      return;
    }
    if (MODE == 2 && k < 4) issue(rb + k + 1, raw[(k + 1) & 1]);
    blur3_h<PATCH>(S, pyr_finish<MODE>(T, raw[MODE == 2 ? (k & 1) : MODE == 3 ? 0 : k], stage), 0u, k4, k6, he[k], ho[k]);
  };
  auto fetch_group = [&](int rb) {
    if (ROWWISE) {
      issue(rb, raw[0]);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) issue(rb + k, raw[ROWWISE ? 0 : k]);
    }
  };
  fetch_group(0);
#pragma unroll
  for (int k = 0; k < 5; k++) hpass(0, k);
  blur3_v<PATCH>(S, he, ho, 0, 1, 2, 3, 4, k4, k6, S.y0);
  for (int rb = 5; rb < nr; rb += 5) {  // wave-uniform
    fetch_group(rb);
    const int y = S.y0 + rb - 4;
    hpass(rb, 0);
    blur3_v<PATCH>(S, he, ho, 1, 2, 3, 4, 0, k4, k6, y);
    hpass(rb, 1);
    blur3_v<PATCH>(S, he, ho, 2, 3, 4, 0, 1, k4, k6, y + 1);
    hpass(rb, 2);
    blur3_v<PATCH>(S, he, ho, 3, 4, 0, 1, 2, k4, k6, y + 2);
    hpass(rb, 3);
    blur3_v<PATCH>(S, he, ho, 4, 0, 1, 2, 3, k4, k6, y + 3);
    hpass(rb, 4);
    blur3_v<PATCH>(S, he, ho, 0, 1, 2, 3, 4, k4, k6, y + 4);
  }
}

// grid = (strip table entries of ONE frame / 4, frames); entry: level `l`, strip `tx` of `pad` (248 px), rows
// [ty, ty + f), u0 / u1 / u2 = x-tap offset / y-tap offset / 8-byte-window flag of the level.
__global__ __launch_bounds__(256, 8) void k_pyrblur(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int frame_bytes,
                                                 int w0, int h0, const uint8_t* __restrict__ in, int in_stride,
                                                 size_t in_frame_stride, const OrbxResizeTap* __restrict__ taps,
                                                 uint8_t* __restrict__ dst, int group, int n_frames,
                                                 const u64* __restrict__ row_stat, uint32_t* __restrict__ feedback,
                                                 OrbxTopLevels top, u64* __restrict__ zero_stat) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int wg = blockIdx.x, f = blockIdx.y;
  if (group) {
    // 1-D grid: groups of `group` frames one after the other; inside a group the workgroup index of the
    // (cost-sorted, heaviest first) strip table is the slow index and the frame the fast one
    const int nwg = (n_tiles + 3) >> 2, per_group = group * nwg;
    const int g = blockIdx.x / per_group, r = blockIdx.x - g * per_group;
    const int gsize = min(group, n_frames - g * group);
    wg = r / gsize;
    f = g * group + (r - wg * gsize);
    if (wg >= nwg) return;  // (the last, smaller group)
  }
  // First pass of the top-rows-first pipeline: the first workgroup of a frame clears the frame's FAST early-exit
  // statistics (the FAST launch that fills them comes after this kernel): a memset node less per batch
  if (zero_stat && wg == 0)
    for (int i = threadIdx.x; i < ORBX_FAST_STAT_WORDS; i += 256) zero_stat[(size_t)f * ORBX_FAST_STAT_WORDS + i] = 0ull;
  // ... and the first workgroup of the batch the rows-needed words the selection kernel will fill (a maximum per batch)
  if (zero_stat && feedback && wg == 0 && f == 0 && threadIdx.x < ORBX_MAX_LEVELS) feedback[2 + threadIdx.x] = 0u;
  const int ti = wg * 4 + wave;
  if (ti >= n_tiles) return;  // whole wave
  const OrbxTileDesc d = tiles[ti];
  const int w = d.w, h = d.h, pitch = d.pitch;
  const int lane = threadIdx.x & 63;
  // Second pass of the top-rows-first pipeline (orbx_api.cpp, enqueue_batch): the FAST tiles of the
  // level's first `na` tile rows have all run (a launch earlier), and their survivor counts tell whether
  // anything below can matter: keypoints are kept in ROW-MAJOR order up to `cap` (src/orb_cpu.cpp:108-110,
  // src/orb.cpp:63), so once those rows hold >= cap survivors no later stage reads a pyramid row at or below
  // this strip's first one (the host cuts the second-pass strips below the last row a descriptor of a
  // keypoint in the top rows can reach), and the strip is not produced at all.
  if (row_stat) {
    const int na = (int)((d.mask_off >> 32) & 0xffu), cap = (int)(uint32_t)d.mask_off;
    if (na > 0) {
      const u64* st = row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + d.stat_index;
      u64 v = 0;
      if (lane < na) v = __hip_atomic_load(&st[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool dead = wave_sum((int)(uint32_t)v) >= cap;
      // ONE strip per frame reports the verdicts of all the frame's levels: running totals {levels skipped,
      // levels produced} (a u32 pair, added to as one u64) that the host reads now and then to decide whether
      // building the pyramid in two passes pays on this stream.  (One report per (frame, level) meant ~1800
      // atomics on the same address per launch: they serialise in the L2 and took 20 of this launch's 25 us.)
      if ((d.mask_off >> 62) && feedback) {
        int ndead = 0;
        for (int i = 0; i < top.n; i++) {
          u64 q = 0;
          if (lane < top.rows[i])
            q = __hip_atomic_load(row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + top.stat_index[i] + lane, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
          ndead += wave_sum((int)(uint32_t)q) >= top.cap[i] ? 1 : 0;
        }
        if (lane == 0)
          atomicAdd(reinterpret_cast<unsigned long long*>(feedback),
                    (unsigned long long)(uint32_t)ndead | ((unsigned long long)(uint32_t)(top.n - ndead) << 32));
      }
      if (dead) return;  // whole wave
    }
  }
  // Strip s starts at dword 62 s: lanes 1..62 are productive, lane 0 / 63 hold the neighbours' dwords --
  // except at the image borders, where the neighbour is a REFLECTION the lane builds from its own dword:
  // lane 0 of the first strip and lane 63 of the last one are productive too (a 1241-px level needs
  // 311 dwords = 63 + 3 x 62 + 63: five strips, not six)
  const int x = d.tx * ORBX_PYRBLUR_TW + lane * 4;
  const bool first_strip = d.tx == 0, last_strip = d.tx + 1 == (int)d.pad;
  Blur3Strip S;
  S.y0 = d.ty;
  S.yend = min(d.ty + d.f, h);
  S.h = h;
  S.pitch = pitch;
  if (S.y0 >= S.yend) return;
  S.rin = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0, 0x00020000);  // (unused: rows are computed, not loaded)
  S.rout = __builtin_amdgcn_make_buffer_rsrc(dst + ((size_t)f * frame_bytes + d.img_off), 0, pitch * h, 0x00020000);
  S.voff = ((lane >= 1 || first_strip) && (lane <= 62 || last_strip) && x < pitch) ? (uint32_t)x : 0xffffffffu;
  S.voff_halo = 0xffffffffu;
  // column REFLECT_101 selectors, as in k_blur3
  const int e4 = (w - 1) & ~3, rbyte = (w - 1) & 3;
  const bool edge = (x == e4);
  S.selL = x == 0 ? 0x05060c0cu : 0x03020100u;
  const uint32_t selC_e = rbyte == 3 ? 0x07060504u : rbyte == 2 ? 0x05060504u : rbyte == 1 ? 0x03040504u : 0x07020304u;
  S.selC = edge ? selC_e : 0x07060504u;
  const uint32_t selR_e = rbyte == 3 ? 0x0c0c0506u : rbyte == 2 ? 0x0c0c0c04u : 0x0c0c0c0cu;
  S.selR = edge ? selR_e : 0x03020100u;
  const int nvalid = w - x;
  S.vmask = nvalid >= 4 ? 0xffffffffu : nvalid <= 0 ? 0u : ((1u << (8 * nvalid)) - 1u);
  const int x_lo = d.tx * ORBX_PYRBLUR_TW, x_hi = x_lo + 4 * 63;
  const bool patch = first_strip || (e4 >= x_lo && e4 <= x_hi);  // wave-uniform

  PyrSrc P;
  P.src = in + (size_t)f * in_frame_stride;
  P.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(P.src), 0, in_stride * (h0 - 1) + w0, 0x00020000);
  P.in_stride = in_stride;
  P.w0 = w0;
  P.h0 = h0;
  P.x = x;
  P.w = w;
  P.h = h;
  P.yt_ofs = 0;
  P.yt_cc = 0;
  P.n8 = 0;
  if (d.l > 0) {  // lane r: the y tap of the strip's input row r (a band has at most 58 rows: r <= 63)
    const OrbxResizeTap ty = taps[d.u1 + reflect101_s(S.y0 - 2 + min(lane, S.yend - S.y0 + 5), h)];
    P.yt_ofs = ty.ofs;
    P.yt_cc = (uint32_t)(uint16_t)ty.c0 | ((uint32_t)(uint16_t)ty.c1 << 16);
  }
  PyrLane T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.ofs[k] = T.cc[k] = T.sel[k] = T.lds_a[k] = 0;
  T.base = 0;
  if (d.l == 0) {
    if (patch)
      pyrblur_strip<0, true>(S, P, T);
    else
      pyrblur_strip<0, false>(S, P, T);
    return;
  }
  // x taps of this lane's four pixels (zero taps for lanes outside the image)
  if (x >= 0 && x < w) {
    const uint4* tp = reinterpret_cast<const uint4*>(taps + d.u0 + x);
    const uint4 t01 = tp[0], t23 = tp[1];
    T.ofs[0] = t01.x; T.ofs[1] = t01.z; T.ofs[2] = t23.x; T.ofs[3] = t23.z;
    T.cc[0] = t01.y; T.cc[1] = t01.w; T.cc[2] = t23.y; T.cc[3] = t23.w;
  }
  if (d.u2 == 3) {
    // staged rows: the strip's span of a source row starts at the (8-byte aligned) pair of the strip's first pixel
    __shared__ __attribute__((aligned(8))) uint8_t s_stage[4][2 * PYR_STAGE_ROW];
    // (scalar loads), aligned so that the dwords of the span end with the source row: the frame's buffer descriptor
    // returns ZERO for a dword that straddles the frame's last byte (orbx_api.cpp, build_plan, verifies the level)
    const int ofs_first = taps[d.u0 + x_lo].ofs, ofs_last = taps[d.u0 + min(x_lo + 255, w - 1)].ofs;
    const uint32_t span0 = (uint32_t)max(ofs_first - ((ofs_first - w0) & 3), 0);
    P.n8 = __builtin_amdgcn_readfirstlane((ofs_last + 2 - (int)span0 + 7) >> 3);
    T.base = span0 + (uint32_t)lane * 8u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t rel = (x >= 0 && x < w) ? T.ofs[k] - span0 : 0u;  // <= ORBX_PYR_STAGE_BYTES - 2 (host-verified)
      T.lds_a[k] = rel & ~3u;
      const uint32_t sb = rel & 3u;  // of {dword a, dword a + 1}: bytes sb, sb + 1
      T.sel[0] |= (sb | ((sb + 1) << 16)) << (4 * k);
    }
    if (patch)
      pyrblur_strip<3, true>(S, P, T, s_stage[wave]);
    else
      pyrblur_strip<3, false>(S, P, T, s_stage[wave]);
  } else if (d.u2) {
    // the window start is clamped so that it never reads past the source row
    T.base = min(T.ofs[0], (uint32_t)(w0 - 8));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t sb = T.ofs[k] - T.base;             // 0..6
      T.sel[k] = 0x0c000c00u | ((sb + 1) << 16) | sb;    // (src[ofs], src[ofs+1]) as two u16 lanes
    }
    if (patch)
      pyrblur_strip<1, true>(S, P, T);
    else
      pyrblur_strip<1, false>(S, P, T);
  } else {
    if (patch)
      pyrblur_strip<2, true>(S, P, T);
    else
      pyrblur_strip<2, false>(S, P, T);
  }
}

}  // namespace

// separable blur of every level >= first_level (lower levels are copied); d_tiles: the strip table
// of ONE frame (orbx_api.cpp: build_blur_tiles)
hipError_t orbx_launch_blur3(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int n_frames,
                             const uint8_t* d_src, uint8_t* d_dst, int first_level) {
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  // four waves per workgroup (measured: 1 -> 53 us, 2 -> 48 us, 4 -> 43 us, 8 / 16 -> 46 us per 64-frame batch)
  dim3 grid(((n_tiles + 3) / 4) | 1, n_frames);  // odd: see orbx_launch_pyrblur
  hipLaunchKernelGGL(k_blur3, grid, dim3(256), 0, s, d_tiles, n_tiles, frame_bytes, d_src, d_dst, first_level);
  return hipGetLastError();
}

// the same with 16 pixels per lane; d_tiles: orbx_api.cpp: build_blur4_tiles (level, strip, first row, rows per band)
hipError_t orbx_launch_blur4(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int n_frames,
                             const uint8_t* d_src, uint8_t* d_dst, int first_level) {
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  dim3 grid(((n_tiles + 3) / 4) | 1, n_frames);  // odd: see orbx_launch_pyrblur
  hipLaunchKernelGGL(k_blur4, grid, dim3(256), 0, s, d_tiles, n_tiles, frame_bytes, d_src, d_dst, first_level);
  return hipGetLastError();
}

// pyramid + blur of every level in one pass (blur on every level, separable kind); d_tiles: the strip
// table of ONE frame (orbx_api.cpp: build_pyrblur_tiles)
hipError_t orbx_launch_pyrblur(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int frame_bytes, int w0, int h0,
                               int n_frames, const uint8_t* d_in, int in_stride, size_t in_frame_stride,
                               const OrbxResizeTap* d_taps, uint8_t* d_dst, int group,
                               const unsigned long long* d_row_stat, uint32_t* d_feedback, const OrbxTopLevels* top,
                               unsigned long long* d_zero_stat) {
  OrbxTopLevels tl{};
  if (top) tl = *top;
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  if (group > 0) {
    const int nwg = (n_tiles + 3) / 4, ngroups = (n_frames + group - 1) / group;
    hipLaunchKernelGGL(k_pyrblur, dim3((unsigned)(ngroups * group * nwg)), dim3(256), 0, s, d_tiles, n_tiles, frame_bytes,
                       w0, h0, d_in, in_stride, in_frame_stride, d_taps, d_dst, group, n_frames, d_row_stat, d_feedback, tl, d_zero_stat);
    return hipGetLastError();
  }
  // Workgroups are dealt round-robin over the 8 XCDs in linear order (x fastest).  With a grid.x that
  // shares a factor with 8 every XCD would get the SAME strips of every frame -- one XCD only cheap
  // level-0 strips, another only resize strips -- and the launch would take as long as the slowest
  // XCD's share (measured: 32 workgroups per frame 467 us, 35 per frame 345 us).  An odd grid.x rotates
  // the assignment from frame to frame; the padding workgroup exits at once.
  dim3 grid(((n_tiles + 3) / 4) | 1, n_frames);
  hipLaunchKernelGGL(k_pyrblur, grid, dim3(256), 0, s, d_tiles, n_tiles, frame_bytes, w0, h0, d_in, in_stride,
                     in_frame_stride, d_taps, d_dst, 0, n_frames, d_row_stat, d_feedback, tl, d_zero_stat);
  return hipGetLastError();
}
