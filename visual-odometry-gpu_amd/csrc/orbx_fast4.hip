// orbx_fast4.hip -- FAST-n segment test + score + (2R+1)^2 NMS of the WHOLE PATH, register streaming,
// gfx950 only.
//   reference semantics: src/orb_cpu.cpp:23-134 (pre-test :39-58, arc test :61-89, score :91-99,
//   NMS + row-major cap :105-134); replaces d_Fast (src/cuda/Fast.cu:30-209) + d_NMS
//   (src/cuda/NMS.cu:21-128).
//
// What the LDS tile kernel (orbx_fast.hip, kept for the stage operators) paid for besides its arithmetic:
// every tile first travels global -> registers -> LDS, a score tile is cleared, four workgroup barriers
// order the phases, and the tile's halo rows and columns are fetched again by the neighbours (1.44x the
// algorithmic bytes).  With the walk and the evaluation compiled out that machinery still took half the
// kernel's time.  This kernel has no workgroup at all:
//   * ONE WAVE owns a strip of 64 dwords (256 pixels) of one level and walks DOWN a tile row of it, the way
//     the streaming blur does (orbx_blur.hip): a row is one coalesced 256-byte load, requested seven rows
//     ahead; the seven most recent rows live in registers as packed 16-bit pairs (even / odd pixels),
//     unpacked ONCE (2 v_perm) and used as south, centre and north row of three pre-test steps.
//   * The E / W neighbours of the 4-point pre-test come from the adjacent lanes' registers (4 DPP wave
//     shifts + 2 v_alignbit per row of 4 pixels), not from LDS.
//   * Each row is also dropped into a wave-private LDS ring (one ds_write_b32 per lane and row) so that
//     the FEW candidates (~3 % of the pixels) can gather their 16 ring pixels later; every 7 rows the
//     candidate flags (one register per lane: 4 pixels x 7 rows) are compacted into a wave-private queue
//     (DPP prefix sum) and evaluated 64 at a time: arc test as shift-and-AND on a 16-bit mask per lane,
//     score = 16 v_sad_u16.
//   * NMS against a wave-private ring of score rows (7 + 2R rows of 256 u16), lagging R rows behind the
//     evaluation; survivors set bits in a 7-row LDS mask block that leaves with one 8-byte store per lane.
//   No barrier, no atomics on the data path, no score tile per workgroup, the image is read once
//   (strips overlap by one dword per side, tile rows by 2 (R + 3) rows of cheap context).
// A strip's mask rows are 4 words of its own (bit 4 * lane + byte), so neighbouring strips never share a
// word: plain stores, no read-modify-write (OrbxLevel::mask_strip_px; the selection kernels decode it).
// Early exit of tile rows that cannot reach the first `cap` row-major survivors: f3_probe_issue
// (orbx_fast_common.h), unchanged.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "orbx_fast_common.h"
#include "orbx_internal.h"
#include "orbx_wave.h"

namespace {

template <int R>
struct F4 {
  static constexpr int HD = orbx_fast4_halo(R);  // dwords of halo per side: ceil((R + 3) / 4)
  static constexpr int S = 64 - 2 * HD;          // productive lanes of an interior strip
  static constexpr int G = 7;                    // centre rows per group (one flag byte per pixel column)
  static constexpr int IMG_SLOTS = 2 * G + 6;    // ring of 14 rows + rows 0..5 mirrored behind it
  static constexpr int IMG_PITCH = 256;
  static constexpr int NS = R > 0 ? G + 2 * R : 1;  // score rows alive at once
  static constexpr int SC_PITCH = 256;              // u16 elements
  static constexpr int QCAP = 256;                  // >= 256: one row plane always fits
};

// byte b of the result = 0x7f if lo <= x0 + b < hi
__device__ __forceinline__ uint32_t f4_colmask(int x0, int lo, int hi) {
  const int a = min(max(lo - x0, 0), 4), b = min(max(hi - x0, 0), 4);
  if (b <= a) return 0u;
  const uint32_t upto_b = b >= 4 ? 0xffffffffu : ((1u << (8 * b)) - 1u);
  const uint32_t below_a = a >= 4 ? 0xffffffffu : ((1u << (8 * a)) - 1u);
  return (upto_b & ~below_a) & 0x7f7f7f7fu;
}
// bit k (k < 7) of every byte set if lo <= y0 + k < hi  (wave-uniform)
__device__ __forceinline__ uint32_t f4_rowmask(int y0, int lo, int hi) {
  const int a = min(max(lo - y0, 0), 7), b = min(max(hi - y0, 0), 7);
  if (b <= a) return 0u;
  return (((1u << b) - 1u) & ~((1u << a) - 1u)) * 0x01010101u;
}

// grid: one wave per (tile of the frame's band-major strip table, frame), the frame index fastest.
// Table entry: l, tx = strip, ty = tile row, f = rows per tile row of the level, u0 = cap, u1 = mask words per
// row, u2 = strips of the level, stat_index, img_off, mask_off.
template <int R, bool ZT>
__global__ __launch_bounds__(64) void k_fast4(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int n_frames,
                                              const uint8_t* __restrict__ pyr, int frame_bytes, int mask_words,
                                              OrbxFastParams fp, u64* __restrict__ mask, u64* __restrict__ row_stat) {
  typedef F4<R> C;
  typedef F3Ring<C::IMG_PITCH> RG;
  __shared__ __attribute__((aligned(16))) uint32_t s_img32[C::IMG_SLOTS * 64];
  __shared__ __attribute__((aligned(16))) uint16_t s_sc[C::NS * C::SC_PITCH];
  __shared__ __attribute__((aligned(16))) uint16_t s_queue[C::QCAP];
  __shared__ __attribute__((aligned(16))) uint32_t s_cflag[64];
  __shared__ __attribute__((aligned(16))) uint32_t s_mask32[C::G * 8];
  // ring pixel - centre pixel + 255 -> bit 0: brighter by >= t, bit 1: darker by >= t (src/orb_cpu.cpp:75-83)
  __shared__ __attribute__((aligned(16))) uint32_t s_lut32[128];
  __shared__ int s_qn;
  const uint8_t* s_img = reinterpret_cast<const uint8_t*>(s_img32);
  const uint8_t* s_lut = reinterpret_cast<const uint8_t*>(s_lut32);

  const int lane = threadIdx.x;
  const int lin = blockIdx.x;
  const int tile = lin / n_frames, f = lin - tile * n_frames;
  const OrbxTileDesc d = f3_tile(tiles, tile);
  if (row_stat && d.ty > 0) {  // early exit (whole wave)
    u64* st = row_stat + (size_t)f * ORBX_FAST_STAT_WORDS;
    u64* dead = st + ORBX_MAX_LEVELS * ORBX_MAX_BANDS + d.l;
    const u64 pv = f3_probe_issue(st + d.stat_index, dead, d.ty, lane);
    if (f3_probe_decide(pv, dead, d.ty, d.u2, d.u0, lane)) return;
  }
  const int w = d.w, h = d.h, pitch = d.pitch, th = d.f;
  const int y0 = d.ty * th, y1 = min(y0 + th, h);
  const int thr = fp.threshold, narc_any = fp.n;
  const bool first = d.tx == 0, last = d.tx + 1 == d.u2;
  const int dw0 = d.tx * C::S;
  const int x_lane = 4 * (dw0 + lane);
  // pixels this strip emits mask bits for, and (R more to each side) computes scores for
  const int P0 = first ? 0 : 4 * (dw0 + C::HD), P1 = last ? 4 * (dw0 + 64) : 4 * (dw0 + 64 - C::HD);
  const uint32_t scol = f4_colmask(x_lane, max(P0 - R, 3), min(P1 + R, w - 3));
  const uint32_t pcol = f4_colmask(x_lane, max(P0, 3), min(P1, w - 3));
  const int sy_lo = max(y0 - R, 3), sy_hi = min(y1 + R, h - 3);  // centre rows that get scores

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(pyr) + ((size_t)f * (size_t)frame_bytes + d.img_off), 0, pitch * h, 0x00020000);
  const uint32_t voff = x_lane < pitch ? (uint32_t)x_lane : 0xffffffffu;
  const int rbase = y0 - R - 3;  // image row of step 0
  auto issue = [&](int t, uint32_t& r) {
    const int ry = min(max(rbase + t, 0), h - 1);  // (rows outside the image are never looked at)
    r = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, ry * pitch, 0);
  };

  // the comparison table of the candidates' ring pixels; the score ring, the corner flags and the mask block start clear
  {
    uint32_t wlo = 0, whi = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int d0 = 8 * lane + b - 255, d1 = d0 + 4;
      wlo |= (uint32_t)((d0 >= thr ? 1 : 0) | (d0 <= -thr ? 2 : 0)) << (8 * b);
      whi |= (uint32_t)((d1 >= thr ? 1 : 0) | (d1 <= -thr ? 2 : 0)) << (8 * b);
    }
    reinterpret_cast<uint2*>(s_lut32)[lane] = make_uint2(wlo, whi);
    if (lane == 0) s_qn = 0;
  }
  if (R > 0) {
    for (int i = lane; i < C::NS * C::SC_PITCH / 8; i += 64) reinterpret_cast<uint4*>(s_sc)[i] = make_uint4(0u, 0u, 0u, 0u);
    s_cflag[lane] = 0u;
  }
  if (lane < C::G * 8) s_mask32[lane] = 0u;

  uint32_t E[7], O[7], raw[7];
#pragma unroll
  for (int k = 0; k < 7; k++) issue(k, raw[k]);
  int slot = 0;  // t % 14 of the step being stored (wave-uniform)
  const uint32_t lds_lane = (uint32_t)lane * 4u;
  auto store_row = [&](uint32_t c) {
    s_img32[slot * 64 + lane] = c;
    if (slot < 6) s_img32[(14 + slot) * 64 + lane] = c;
    slot = slot == 13 ? 0 : slot + 1;
  };
  (void)lds_lane;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint32_t c = raw[k];
    issue(7 + k, raw[k]);
    store_row(c);
    E[k] = __builtin_amdgcn_perm(c, c, 0x0c020c00u);
    O[k] = __builtin_amdgcn_perm(c, c, 0x0c030c01u);
  }

  const uint32_t T = (uint32_t)thr * 0x00010001u;
  const int ngroups = (y1 - y0 + 2 * R + C::G - 1) / C::G;
  uint32_t prev_c = 0u;  // corner flags of the previous group
  int sm0 = 0;           // (7 g) % NS
  int surv = 0;
  u64* const mask_base = mask + ((size_t)f * (size_t)mask_words + d.mask_off) + (size_t)(4 * d.tx);
  const int wpr = d.u1;

  for (int g = 0; g < ngroups; g++) {
    const int gpar = (g & 1) * 7;    // image ring slot of centre row 7 g
    const int cy0 = y0 - R + 7 * g;  // centre row of step 0 of this group
    uint32_t acc = 0u;  // bit SET = not a candidate
#pragma unroll
    for (int k = 0; k < 7; k++) {
      const int jn = (6 + k) % 7, jc = (3 + k) % 7, jo = k;  // ring slots: new (south), centre, north
      const uint32_t c = raw[jn];
      issue(7 * g + 13 + k, raw[jn]);
      store_row(c);
      E[jn] = __builtin_amdgcn_perm(c, c, 0x0c020c00u);  // pixels 0, 2 as 16-bit lanes
      O[jn] = __builtin_amdgcn_perm(c, c, 0x0c030c01u);  // pixels 1, 3
      const uint32_t Ec = E[jc], Oc = O[jc];
      // neighbours' pairs: lane + 1 (wave_shl) and lane - 1 (wave_shr); the outermost lanes get zeros (their
      // pixels next to the wave's edge are never valid)
      const uint32_t En = __builtin_amdgcn_update_dpp(0u, Ec, 0x130, 0xf, 0xf, true);
      const uint32_t On = __builtin_amdgcn_update_dpp(0u, Oc, 0x130, 0xf, 0xf, true);
      const uint32_t Ep = __builtin_amdgcn_update_dpp(0u, Ec, 0x138, 0xf, 0xf, true);
      const uint32_t Op = __builtin_amdgcn_update_dpp(0u, Oc, 0x138, 0xf, 0xf, true);
      // east = x + 3: (p0, p2) -> (p3, p5) = (Oc.hi, On.lo); (p1, p3) -> (p4, p6) = En
      // west = x - 3: (p0, p2) -> (p-3, p-1) = Op;           (p1, p3) -> (p-2, p0) = (Ep.hi, Ec.lo)
      const uint32_t ee = __builtin_amdgcn_alignbit(On, Oc, 16), eo = En;
      const uint32_t we = Op, wo = __builtin_amdgcn_alignbit(Ec, Ep, 16);
      const uint32_t ze = f3_pretest<ZT>(Ec, E[jo], ee, E[jn], we, T);
      const uint32_t zo = f3_pretest<ZT>(Oc, O[jo], eo, O[jn], wo, T);
      const uint32_t F = __builtin_amdgcn_perm(zo, ze, 0x07030501u);  // the four sign bytes side by side
      acc = (F & 0x80808080u) | (acc >> 1);
    }
    // row k of the group sits in bit k + 1 of every byte
    uint32_t cand = (~acc >> 1) & scol & f4_rowmask(cy0, sy_lo, sy_hi);
    if (R == 0) cand &= pcol & f4_rowmask(cy0, y0, y1);

    // ---- a pass over the set bits of `bits`, 64 at a time: fn(active, entry), entry = lane << 5 | byte << 3 | row.
    // Queue slots are handed out by ONE LDS atomic per lane that has entries (the order of the queue does not
    // matter: every entry is handled on its own, the results are bit masks and scores at fixed places), not by a
    // prefix sum: eight dependent DPP steps twice per group were ~10 % of this kernel's vector instructions.
    auto for_each = [&](uint32_t bits, auto&& fn) {
      if (__ballot(bits != 0u) == 0ull) return;
      const uint32_t ebase = (uint32_t)lane << 5;
      auto pass = [&](uint32_t b) -> int {  // entries of b -> queue; returns their number (at most QCAP are stored)
        const int cnt = __popc(b);
        int pos = 0;
        if (cnt) pos = atomicAdd(&s_qn, cnt);
        wave_lds_sync();
        const int total = __builtin_amdgcn_readfirstlane(s_qn);
        wave_lds_sync();
        if (lane == 0) s_qn = 0;
        if (total <= C::QCAP) {
          while (b) {
            const int bp = __ffs(b) - 1;
            b &= b - 1;
            s_queue[pos++] = (uint16_t)(ebase | (uint32_t)bp);
          }
        }
        return total;
      };
      auto drain = [&](int n) {
        wave_lds_sync();
        for (int q0 = 0; q0 < n; q0 += 64) {
          const int q = q0 + lane;
          const bool active = q < n;
          fn(active, active ? (uint32_t)s_queue[q] : 0u);
        }
        wave_lds_sync();
      };
      const int total = pass(bits);
      if (total <= C::QCAP) {
        drain(total);
      } else {  // noise, tiny thresholds: one row plane at a time (at most 256 entries each)
        for (int r = 0; r < C::G; r++) {
          const uint32_t rb = bits & (0x01010101u << r);
          if (__ballot(rb != 0u) == 0ull) continue;
          drain(pass(rb));
        }
      }
    };

    // ---- full segment test + score of the candidates (src/orb_cpu.cpp:61-101)
    auto eval = [&](bool active, uint32_t e) {
      const int xs = (int)(e >> 3), k = (int)(e & 7u);  // pixel within the strip, row within the group
      const uint8_t* p0 = s_img + (gpar + k) * C::IMG_PITCH + xs - 3;  // pixel (x - 3, y - 3); xs >= 3 for every valid entry
      if (!active) p0 = s_img + 3 * C::IMG_PITCH;
      const int Ip = p0[3 * C::IMG_PITCH + 3];
      int v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = p0[RG::off(i)];
      // the 16 comparisons per polarity come out of the table: ONE add per ring pixel (its address), the two bits
      // of every pixel shifted into a 32-bit word -- 16 pixels x 2 bits = the whole circle, so "n contiguous" is
      // rotate-and-AND on that word for both polarities at once (rotations by even amounts keep them apart)
      const uint8_t* lut = s_lut + (255 - Ip);
      uint32_t M = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) M = (M << 2) | (uint32_t)lut[v[i]];
      auto rot = [](uint32_t x, int sh) { return __builtin_amdgcn_alignbit(x, x, sh); };
      uint32_t a = M;
      if (narc_any == 9) {
        a &= rot(a, 2);
        a &= rot(a, 4);
        a &= rot(a, 8);
        a &= rot(a, 2);
      } else if (narc_any == 12) {
        a &= rot(a, 2);
        a &= rot(a, 4);
        a &= rot(a, 8);
        a &= rot(a, 8);
      } else {
        int kk = 1;
        while (2 * kk <= narc_any) {
          a &= rot(a, 2 * kk);
          kk *= 2;
        }
        if (kk < narc_any) a &= rot(a, 2 * (narc_any - kk));
      }
      const bool corner = a != 0u;
      uint32_t score = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) score = __builtin_amdgcn_sad_u16((uint32_t)Ip, (uint32_t)v[i], score);
      // (threshold 0: a flat neighbourhood passes the arc test with score 0, and a score of 0 is no keypoint:
      // src/orb_cpu.cpp:110 skips scores <= 0 before it looks at the NMS radius)
      if (active && corner && score > 0u) {
        if (R > 0) {
          int s = sm0 + k;
          s = s >= C::NS ? s - C::NS : s;
          // (the score rows are never cleared: an entry carries the low bits of its centre row's index, and a row's
          // slot is reused NS rows later -- the same tag comes back after lcm(NS, 16) >= 144 rows, more than a tile
          // row has)
          s_sc[s * C::SC_PITCH + xs] = (uint16_t)(score | ((uint32_t)((7 * g + k) & 15) << 12));
          atomicOr(&s_cflag[e >> 5], 1u << (e & 31u));
        } else {  // no NMS: every corner is kept
          atomicOr(&s_mask32[k * 8 + (xs >> 5)], 1u << (xs & 31));
        }
      }
    };
    for_each(cand, eval);

    if (R > 0) {
      // corners back to their lanes; NMS row n of this group = centre row 7 g - R + n: the last R rows of the
      // previous group and the first 7 - R rows of this one (their neighbours' scores are all known now)
      const uint32_t cur_c = s_cflag[lane];
      s_cflag[lane] = 0u;
      constexpr uint32_t LOWR = ((1u << R) - 1u) * 0x01010101u;
      uint32_t nbits = ((prev_c >> (7 - R)) & LOWR) | ((cur_c << R) & (0x7f7f7f7fu & ~LOWR));
      prev_c = cur_c;
      nbits &= pcol & f4_rowmask(cy0 - R, y0, y1);
      // ties survive (src/orb_cpu.cpp:110-133)
      auto nms = [&](bool active, uint32_t e) {
        const int xs = (int)(e >> 3), n = (int)(e & 7u);
        int sr = sm0 - R + n;  // score ring slot of the centre row
        sr = sr < 0 ? sr + C::NS : sr;
        sr = sr >= C::NS ? sr - C::NS : sr;
        const int xq = active ? xs : R;
        const int mrow = 7 * g - R + n;  // index of the centre row (its own entry exists: it is a corner)
        const uint32_t s = (uint32_t)s_sc[sr * C::SC_PITCH + xq] & 0xfffu;
        bool keep = true;
#pragma unroll
        for (int dy = -R; dy <= R; dy++) {
          int rr = sr + dy;
          rr = rr < 0 ? rr + C::NS : rr;
          rr = rr >= C::NS ? rr - C::NS : rr;
          const uint16_t* row = s_sc + rr * C::SC_PITCH + xq - R;
          // an entry beats this one if it belongs to row mrow + dy (tag) and its score is larger:
          // entry - (tag << 12) - (s + 1), unsigned, below 4095 - s
          const uint32_t base = ((uint32_t)((mrow + dy) & 15) << 12) + s + 1u, lim = 4095u - s;
#pragma unroll
          for (int dx = 0; dx <= 2 * R; dx++)
            if (dy != 0 || dx != R) keep = keep & !((uint32_t)row[dx] - base < lim);
        }
        if (active && keep) atomicOr(&s_mask32[n * 8 + (xs >> 5)], 1u << (xs & 31));
      };
      for_each(nbits, nms);
    }

    // ---- the mask rows this group finished: lane = (row, word); every row of the tile row leaves exactly once
    wave_lds_sync();
    {
      const int r = lane >> 2, wq = lane & 3;
      const int y = cy0 - R + r;  // (R == 0: cy0 + r)
      if (lane < C::G * 4) {
        const uint2 m = reinterpret_cast<const uint2*>(s_mask32)[lane];
        reinterpret_cast<uint2*>(s_mask32)[lane] = make_uint2(0u, 0u);
        if (y >= y0 && y < y1) {
          mask_base[(size_t)y * wpr + wq] = (u64)m.x | ((u64)m.y << 32);
          surv += __popc(m.x) + __popc(m.y);
        }
      }
    }
    wave_lds_sync();
    sm0 += C::G;
    sm0 = sm0 >= C::NS ? sm0 - C::NS : sm0;
  }
  if (row_stat) {
    const int ws = wave_sum(surv);
    if (lane == 0)
      __hip_atomic_fetch_add(row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + d.stat_index + d.ty,
                             (1ull << 32) | (u64)(uint32_t)ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int R>
void launch_fast4(unsigned wgs, hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames, const uint8_t* d_pyr,
                  int frame_bytes, int mask_words, OrbxFastParams fp, unsigned long long* d_mask,
                  unsigned long long* d_row_stat) {
  if (fp.threshold == 0)
    hipLaunchKernelGGL((k_fast4<R, true>), dim3(wgs), dim3(64), 0, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes,
                       mask_words, fp, d_mask, d_row_stat);
  else
    hipLaunchKernelGGL((k_fast4<R, false>), dim3(wgs), dim3(64), 0, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes,
                       mask_words, fp, d_mask, d_row_stat);
}

}  // namespace

// d_tiles: the n_tiles (strip, tile row) units of ONE frame in band-major order (orbx_api.cpp: build_fast_tiles
// with strips); one wave per (unit, frame).  The level masks are in STRIP layout (OrbxLevel::mask_strip_px).
// d_row_stat: n_frames * ORBX_FAST_STAT_WORDS zeroed u64 (or NULL: no early exit)
hipError_t orbx_launch_fast4(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames, const uint8_t* d_pyr,
                             int frame_bytes, int mask_words, OrbxFastParams fp, unsigned long long* d_mask,
                             unsigned long long* d_row_stat) {
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  const long long total = (long long)n_tiles * n_frames;
  if (total > 0x7fffffffll) return hipErrorInvalidValue;
  switch (fp.nms_radius) {
    case 0:
      launch_fast4<0>((unsigned)total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_row_stat);
      break;
    case 1:
      launch_fast4<1>((unsigned)total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_row_stat);
      break;
    case 2:
      launch_fast4<2>((unsigned)total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_row_stat);
      break;
    default:
      launch_fast4<3>((unsigned)total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_row_stat);
      break;
  }
  return hipGetLastError();
}
