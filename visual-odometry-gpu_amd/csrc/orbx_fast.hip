// orbx_fast.hip -- FAST-n segment test + score + (2R+1)^2 NMS, fused, gfx950 only.
//   reference semantics: src/orb_cpu.cpp:23-134 (pre-test :39-58, arc test :61-89, score :91-99,
//   NMS + row-major cap :105-134); replaces d_Fast (src/cuda/Fast.cu:30-209) + d_NMS
//   (src/cuda/NMS.cu:21-128).
//
// What bounds it (profiles/r02): the vector unit is ~85 % busy (28 lane-operations per pixel, half of
// them the pre-test walk), and each tile's life is a chain of latencies (tile record, image rows, four
// barriers, mask store) that six resident workgroups per CU only partly hide -- so the design is about
// wave-instructions per pixel AND about taking latencies off that chain:
//   Tiles are 128 pixels wide and up to 49 - 2R rows tall (7 row segments x 7 rows per walk; 8 rows per
//   walk cost the same per pixel, but shorter tile rows make the first pass of the top-rows-first pipeline
//   smaller: orbx_api.cpp, enqueue_batch); the host balances the tile rows of a level (a 218-row level
//   gets 5 tile rows of 44, not 4 of 47 and one of 30), and the walking threads then take 6 instead of 7
//   rows each.  A workgroup handles two consecutive tiles (the same
//   tile of two consecutive frames); the second tile's rows are requested while the first one's
//   candidates are evaluated.
//   phase 1  tile + halo -> LDS, row-coalesced 8-byte loads at addresses clamped into the level
//            (what the tile holds outside the image is never looked at).
//   phase 2  4-point pre-test on EVERY pixel.  A thread owns one dword column (4 pixels) of the
//            score region and walks K = 7 rows of it: each image row is unpacked ONCE into packed
//            16-bit pairs (2 v_perm) and then serves as north row, centre row and south row of three
//            different steps; the pre-test itself is a 8-op v_pk_min/max_u16 network per pixel pair
//            (2nd smallest / 2nd largest of N,E,S,W) + 4 packed ops for both polarities.  The
//            candidate flags of the 7 rows x 4 pixels end up in one register per thread.
//   phase 3  the candidates (~3-6 % of the pixels) are compacted into one LDS queue (wave prefix sums,
//            one LDS atomic per wave) and evaluated 64 at a time: one ring pixel per ds_read_u8, the
//            16 comparisons per polarity shifted into a 16-bit mask per lane, the 9-contiguous-arc
//            test as four shift-and-AND steps on that mask, score = 16 v_sad_u16.
//   phase 4  NMS of the tile-interior candidates that became corners against the dense LDS score tile.
//   phase 5  (first wave) survivor mask rows -> global, one 16-byte store per tile row; tile-row
//            statistics for the early exit.
// Early exit of tiles that cannot reach the first `cap` row-major survivors: see f3_probe_issue.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "orbx_fast_common.h"
#include "orbx_internal.h"
#include "orbx_wave.h"

namespace {

// geometry of a tile (R = NMS radius)
template <int R>
struct F3 {
  static constexpr int TW = ORBX_FAST3_TW;       // 128 output pixels = 2 mask words per row
  static constexpr int NC = TW / 4 + 2;          // dword columns of the score region: x0-4 .. x0+TW+4
  static constexpr int NSEG = 256 / NC;          // 7 row segments -> 238 walking threads
  static constexpr int K = ORBX_FAST3_K;         // rows per walk
  static constexpr int SC_ROWS = NSEG * K;       // score region rows: y0-R .. y0+TH+R
  static constexpr int TH = SC_ROWS - 2 * R;     // output rows of a tile
  static constexpr int IMG_PITCH = TW + 16;      // bytes: x0-8 .. x0+TW+8
  static constexpr int IMG_DW = IMG_PITCH / 4;
  static constexpr int IMG_ROWS = SC_ROWS + 6;   // y0-R-3 .. y0+TH+R+3
  static constexpr int SC_PITCH = NC * 4;        // u16 elements
  static constexpr int QCAP = 736;               // candidate queue entries (more candidates: several passes)
  static constexpr int MASK_DW = TW / 32;        // mask dwords per tile row
};
static_assert(F3<1>::NC == 34 && F3<1>::NSEG == 7 && F3<1>::K <= 8, "walk mapping (tid / 34 by multiply-shift) assumes 34 x 7; a flag byte holds <= 8 rows");
static_assert(F3<0>::TH == orbx_fast3_tile_h(0) && F3<3>::TH == orbx_fast3_tile_h(3), "host tile tables use orbx_fast3_tile_h");

// the walk of one thread: candidate flags of `keff` (<= K) rows x 4 pixels, byte b = pixel b of the
// dword, bit k = row k of the walk (before border masking).  p points at the dword LEFT of the
// thread's column in the first image row of the walk (LDS image row seg*keff = score row
// seg*keff - 3).  keff is uniform over the workgroup: the early break is a scalar branch.
template <int R, bool ZT>
__device__ __forceinline__ uint32_t f3_walk(const uint32_t* p, uint32_t T, int keff) {
  typedef F3<R> G;
  uint32_t E[G::K + 6], O[G::K + 6];
  uint32_t acc = 0;  // bit SET = not a candidate
#pragma unroll
  for (int i = 0; i < G::K + 6; i++) {
    // (no `break`: it would keep the loop rolled, with E/O indexed through M0)
    if (i >= G::K + 5 && keff < G::K) continue;  // keff is K or K-1: only the last step is optional
    const uint32_t c = p[i * G::IMG_DW + 1];
    // even pixels (0,2) and odd pixels (1,3) of the dword as 16-bit lanes
    E[i] = __builtin_amdgcn_perm(c, c, 0x0c020c00u);
    O[i] = __builtin_amdgcn_perm(c, c, 0x0c030c01u);
    if (i >= 6) {
      const int j = i - 3;  // centre row of this step
      const uint32_t Lw = p[j * G::IMG_DW], Cw = p[j * G::IMG_DW + 1], Rw = p[j * G::IMG_DW + 2];
      // perm: bytes 0-3 = 2nd argument, 4-7 = 1st
      // east = x+3: px0 -> C.b3, px1 -> R.b0, px2 -> R.b1, px3 -> R.b2
      const uint32_t ee = __builtin_amdgcn_perm(Rw, Cw, 0x0c050c03u), eo = __builtin_amdgcn_perm(Rw, Cw, 0x0c060c04u);
      // west = x-3: px0 -> L.b1, px1 -> L.b2, px2 -> L.b3, px3 -> C.b0
      const uint32_t we = __builtin_amdgcn_perm(Cw, Lw, 0x0c030c01u), wo = __builtin_amdgcn_perm(Cw, Lw, 0x0c040c02u);
      const uint32_t ze = f3_pretest<ZT>(E[j], E[j - 3], ee, E[j + 3], we, T);
      const uint32_t zo = f3_pretest<ZT>(O[j], O[j - 3], eo, O[j + 3], wo, T);
      // the four sign bytes (px0..px3) side by side; after K steps row k sits in bit k of every byte
      const uint32_t F = __builtin_amdgcn_perm(zo, ze, 0x07030501u);
      acc = (F & 0x80808080u) | (acc >> 1);
    }
  }
  return ~acc >> (8 - keff);  // (fewer than 8 steps: the rows have not travelled all the way down the byte)
}

// queue entry (walking thread, flag index) -> score-region row / column
__device__ __forceinline__ void f3_decode(uint32_t e, int keff, int& sr, int& sc) {
  const uint32_t t = e >> 5, bi = e & 31u;
  const uint32_t seg = (t * 241u) >> 13;  // t / 34 for t < 256
  const uint32_t col = t - seg * 34u;
  sr = (int)(seg * (uint32_t)keff + (bi & 7u));
  sc = (int)(col * 4u + (bi >> 3));
}

// full segment test + score of one candidate per lane (src/orb_cpu.cpp:61-101).
// NARC > 0: arc length known at compile time (the shift-and-AND chain of the run test unrolls);
// NARC == 0: any n.  The 16 comparisons per polarity are collected as a 16-bit mask per lane and the
// run test is per-lane arithmetic: the variant that ran the arc test on the scalar unit (one v_cmp per
// ring pixel into a 64-bit lane mask, AND-doubling over the 16 masks with s_and_b64) needed every SGPR
// there is, and in the persistent kernel -- tile cursor, two tile records and the kernel arguments
// live across it -- paid for that with ~150 v_writelane / v_readlane spill instructions per tile.
template <int R, int NARC>
__device__ __forceinline__ void f3_eval(const uint8_t* s_img, uint16_t* s_score, uint16_t* s_queue, int nq, int thr,
                                        int n, int keff, int tid, const uint8_t* s_lut) {
  typedef F3<R> G;
  typedef F3Ring<G::IMG_PITCH> RG;
  const int narc = NARC > 0 ? NARC : n;
  for (int q0 = (tid & ~63); q0 < nq; q0 += 256) {  // wave-uniform
    const int q = q0 + (tid & 63);
    const bool active = q < nq;
    const uint32_t e = active ? s_queue[q] : 0u;
    int sr, sc;
    f3_decode(e, keff, sr, sc);
    const int pos = sr * G::SC_PITCH + sc;
    if (active) s_queue[q] = (uint16_t)pos;  // the NMS pass reads positions
    const uint8_t* p0 = s_img + sr * G::IMG_PITCH + sc + 1;  // pixel (x-3, y-3)
    const int Ip = p0[3 * G::IMG_PITCH + 3];
    int v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = p0[RG::off(k)];
    // The 16 comparisons per polarity come out of a table (ring pixel - centre + 255 -> bit 0: brighter by >= t,
    // bit 1: darker by >= t; src/orb_cpu.cpp:75-83): ONE add per ring pixel (its address) and one shift-or that
    // moves the pixel's two bits into a 32-bit word -- 16 pixels x 2 bits = the whole circle, so "n contiguous" is
    // rotate-and-AND on that word for both polarities at once (rotations by even amounts keep them apart).
    // 44 vector instructions where sub + v_alignbit per pixel and polarity and two run tests took 84.
    const uint8_t* lut = s_lut + (255 - Ip);
    uint32_t M = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) M = (M << 2) | (uint32_t)lut[v[k]];
    auto rot = [](uint32_t x, int sh) { return __builtin_amdgcn_alignbit(x, x, sh); };
    uint32_t a = M;
    int kk = 1;
#pragma unroll
    for (int st = 0; st < 4; st++)
      if (2 * kk <= narc) {
        a &= rot(a, 2 * kk);
        kk *= 2;
      }
    if (kk < narc) a &= rot(a, 2 * (narc - kk));
    const bool corner = a != 0u;
    uint32_t score = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) score = __builtin_amdgcn_sad_u16((uint32_t)Ip, (uint32_t)v[k], score);
    if (active && corner) s_score[pos] = (uint16_t)score;
  }
}

// NMS of the candidates that became corners (ties survive, src/orb_cpu.cpp:110-133); survivors of
// the tile interior set their bit in the LDS mask.  Only candidates of the tile interior look at their
// neighbours (a halo candidate is never kept), so the score tile needs no guard rows or columns.
template <int R>
__device__ __forceinline__ void f3_nms(const uint16_t* s_score, const uint16_t* s_queue, int nq, uint32_t* s_mask32,
                                       int th, int tid) {
  typedef F3<R> G;
  for (int q0 = (tid & ~63); q0 < nq; q0 += 256) {
    const int q = q0 + (tid & 63);
    if (q < nq) {
      const int pos = s_queue[q];
      const int s = s_score[pos];
      const int sr = pos / G::SC_PITCH, sc = pos - sr * G::SC_PITCH;
      const int iy = sr - R, ix = sc - 4;
      if (s > 0 && iy >= 0 && iy < th && ix >= 0 && ix < G::TW) {
        bool keep = true;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
          for (int dx = -R; dx <= R; dx++)
            if (dy != 0 || dx != 0) keep = keep & !(s_score[pos + dy * G::SC_PITCH + dx] > s);
        if (keep) atomicOr(&s_mask32[iy * G::MASK_DW + (ix >> 5)], 1u << (ix & 31));
      }
    }
  }
}

// A workgroup works through `chunk` CONSECUTIVE linear tile indices (linear index = tile * n_frames + frame:
// the tile table of one frame is in band-major order, so tile row b of every frame comes before tile row
// b+1 of any frame; consecutive indices = the same tile of consecutive frames).  A tile's life is a chain of
// latencies -- tile record, image rows, four barriers, the mask store -- and with every phase's arithmetic
// removed the one-tile-per-workgroup kernel still took half its time.  Here the NEXT tile's image rows (and
// its early-exit statistics) are requested right after the walk of the current one and land while the
// candidates are evaluated.  Measured (256 frames per launch, every tile working / early exit on):
// chunk 1: 315 / 183 us, 2: 305 / 163, 4: 305 / 173, 8: 312 / 220 (long chunks delay the statistics the early
// exit feeds on); fully persistent workgroups (one per resident slot, striding through the whole list)
// were SLOWER, 342 / 231 us: the workgroups of a CU then march through the phases in step and all want the
// vector unit, or all wait for memory, at the same time.
template <int R>
__global__ __launch_bounds__(256, 6) void k_fast3(const OrbxTileDesc* __restrict__ tiles, int n_tiles, int n_frames, int chunk,
                                               const uint8_t* __restrict__ pyr, int frame_bytes, int mask_words,
                                               OrbxFastParams fp, u64* __restrict__ mask,
                                               uint16_t* __restrict__ scores_out, u64* __restrict__ row_stat) {
  typedef F3<R> G;
  __shared__ __attribute__((aligned(16))) uint32_t s_img32[G::IMG_ROWS * G::IMG_DW];
  __shared__ __attribute__((aligned(16))) uint16_t s_score[G::SC_ROWS * G::SC_PITCH];
  static_assert(sizeof(s_score) % 16 == 0, "cleared with 16-byte stores");
  __shared__ uint16_t s_queue[G::QCAP];
  // survivor mask: zero when the kernel starts; the first wave clears each row right after reading it for
  // the store, so the mask is clean again long before the next tile's NMS sets bits
  __shared__ __attribute__((aligned(16))) uint32_t s_mask32[G::TH * G::MASK_DW];
  __shared__ int s_qn;
  __shared__ __attribute__((aligned(16))) uint32_t s_lut32[128];  // f3_eval's comparison table
  __shared__ int s_wtot[4];
  __shared__ int s_skip_next;
  __shared__ int s_srch[2][4];
  uint16_t* const score0 = s_score;
  const uint8_t* s_img = reinterpret_cast<const uint8_t*>(s_img32);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int thr = fp.threshold, n = fp.n;
  if (tid < 128) {  // (read after the first barrier of the tile loop at the earliest)
    uint32_t wv = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int dd = 4 * tid + b - 255;
      wv |= (uint32_t)((dd >= thr ? 1 : 0) | (dd <= -thr ? 2 : 0)) << (8 * b);
    }
    s_lut32[tid] = wv;
  }
  const uint8_t* s_lut = reinterpret_cast<const uint8_t*>(s_lut32);
  // this workgroup's tiles: `chunk` consecutive linear indices (the same tile of consecutive frames)
  const int lin0 = (int)blockIdx.x * chunk, lin_end = min(lin0 + chunk, n_tiles * n_frames);
  auto advance = [&](int& t, int& f) {
    if (++f == n_frames) {
      f = 0;
      t++;
    }
  };
  auto past_end = [&](int t, int f) { return t * n_frames + f >= lin_end; };
  // fill geometry of this thread: 18 8-byte loads per image row, 14 rows per pass
  constexpr int CPR = G::IMG_PITCH / 8, RPP = 256 / CPR, NP = (G::IMG_ROWS + RPP - 1) / RPP;
  const int fr0 = (tid * 3641) >> 16;  // tid / 18
  const int fc = tid - fr0 * CPR;
  // Only pixels whose whole ring lies inside the image can become candidates (the flags of all others
  // are masked after the walk), so what the LDS tile holds OUTSIDE the image is never looked at: the
  // addresses are clamped into the level instead of predicating every load, and LDS rows past the
  // tile's own rows are simply loaded too (the LDS tile always has G::IMG_ROWS rows).  That also makes
  // a speculative load of a tile that turns out to be skipped harmless.
  auto issue_rows = [&](const OrbxTileDesc& d, int f, uint2 (&v)[NP]) {
    const uint8_t* img = pyr + ((size_t)f * (size_t)frame_bytes + d.img_off);
    const int gx = min(max(d.tx * G::TW - 8 + 8 * fc, 0), d.pitch - 8);
    const int gy0 = d.ty * d.f - R - 3 + fr0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int gy = min(max(gy0 + RPP * k, 0), d.h - 1);
      v[k] = *reinterpret_cast<const uint2*>(img + (uint32_t)(gy * d.pitch + gx));
    }
  };
  auto stat_of = [&](const OrbxTileDesc& d, int f) { return row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + d.stat_index; };
  auto dead_of = [&](const OrbxTileDesc& d, int f) {
    return row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + ORBX_MAX_LEVELS * ORBX_MAX_BANDS + d.l;
  };
  // The first candidate at or after (t, f) that has to be worked on -- four candidates per round, one per
  // wave, a barrier per round.  Block-uniform; false: this workgroup is done.
  int srch_round = 0;
  auto search = [&](int& t, int& f) -> bool {
    for (;;) {
      if (past_end(t, f)) return false;
      int ct = t, cf = f;
      for (int k = 0; k < wv; k++) advance(ct, cf);
      int flag = 2;  // past the end
      if (!past_end(ct, cf)) {
        const OrbxTileDesc d = f3_tile(tiles, ct);
        bool skip = false;
        if (d.ty > 0) {
          u64* dead = dead_of(d, cf);
          const u64 st = f3_probe_issue(stat_of(d, cf), dead, d.ty, lane);
          skip = f3_probe_decide(st, dead, d.ty, d.u2, d.u0, lane);
        }
        flag = skip ? 1 : 0;
      }
      if (lane == 0) s_srch[srch_round & 1][wv] = flag;
      __syncthreads();
      int k = 0, fl = 1;
#pragma unroll
      for (int q = 3; q >= 0; q--) {
        const int x = __builtin_amdgcn_readfirstlane(s_srch[srch_round & 1][q]);  // (uniform: keeps the cursor scalar)
        if (x != 1) {
          k = q;
          fl = x;
        }
      }
      srch_round++;
      if (fl == 1) k = 4;  // all four are skipped
      for (int q = 0; q < k; q++) advance(t, f);
      if (fl != 1) return fl == 0;
    }
  };

  int tile = lin0 / n_frames, f = lin0 - tile * n_frames;
  if (row_stat) {
    if (!search(tile, f)) return;
  } else if (past_end(tile, f)) {
    return;
  }
  uint2 v[NP];
  issue_rows(f3_tile(tiles, tile), f, v);
  for (int i = tid; i < G::TH * G::MASK_DW; i += 256) s_mask32[i] = 0u;
  for (;;) {
    // everything about this tile in one 64-byte scalar load; the table is shared by all frames
    const OrbxTileDesc d = f3_tile(tiles, tile);
    const int w = d.w, h = d.h;
    const int tx = d.tx, ty = d.ty;
    // tile height of this level (<= G::TH; the host balances the tile rows of a level) and the rows
    // each walking thread then owns (K, or K-1 for the shorter tiles)
    const int th = d.f;
    const int keff = (th + 2 * R + G::NSEG - 1) / G::NSEG >= G::K ? G::K : G::K - 1;
    const int x0 = tx * G::TW, y0 = ty * th;

    // ---- phase 1: the tile's rows (requested during the previous tile) -> LDS; the score tile (with
    // its guard rows) is cleared
    {
      constexpr int NZ = (int)(sizeof(s_score) / 16);
      for (int i = tid; i < NZ; i += 256) reinterpret_cast<uint4*>(s_score)[i] = make_uint4(0u, 0u, 0u, 0u);
      if (tid == 0) s_qn = 0;
#pragma unroll
      for (int k = 0; k < NP; k++)
        if (fr0 + RPP * k < G::IMG_ROWS)  // (threads 252..255 have fr0 == 14: rows 14, 28, 42, 56 twice -- same data)
          reinterpret_cast<uint2*>(s_img32)[(fr0 + RPP * k) * CPR + fc] = v[k];
    }
    __syncthreads();

    // ---- phase 2: pre-test walk (src/orb_cpu.cpp:39-58)
    uint32_t cand = 0;
    {
      const int seg = (tid * 241) >> 13;  // tid / 34
      const int col = tid - seg * G::NC;
      const int gy_first = y0 - R + seg * keff;  // image row of walk row 0
      // rows that can hold a corner: 3 <= gy < h-3 (src/orb_cpu.cpp:35) and that this tile needs
      const int lo_y = max(3 - gy_first, 0), hi_y = min(min(h - 3, y0 + th + R) - gy_first, keff);
      if (seg < G::NSEG && hi_y > lo_y) {  // (a wave whose rows all lie outside skips the walk)
        const uint32_t* p = s_img32 + (seg * keff) * G::IMG_DW + col;
        const uint32_t T = (uint32_t)thr * 0x00010001u;
        cand = thr == 0 ? f3_walk<R, true>(p, T, keff) : f3_walk<R, false>(p, T, keff);
        // pixels that can hold a corner and that this tile needs: 3 <= gx < w-3, x0-R <= gx < x0+TW+R
        const int gx = x0 - 4 + 4 * col;
        const int lo_x = max(max(3, x0 - R) - gx, 0), hi_x = min(min(w - 3, x0 + G::TW + R) - gx, 4);
        const uint32_t cm = hi_x > lo_x ? ((0xffffffffu >> (32 - 8 * hi_x)) & ~((1u << (8 * lo_x)) - 1u)) : 0u;
        const uint32_t rm = (((1u << hi_y) - 1u) & ~((1u << lo_y) - 1u)) * 0x01010101u;
        cand &= cm & rm;
      }
    }

    // ---- the workgroup's next tile: its rows are requested now (speculatively when the early exit is
    // on: clamped addresses, nothing is lost but the traffic if the tile turns out to be dead), and the
    // first wave asks for its early-exit statistics; both are looked at after the candidates' evaluation
    int ntile = tile, nf = f;
    advance(ntile, nf);
    const bool have_next = !past_end(ntile, nf);  // block-uniform
    u64 pst = 0;
    if (have_next) {
      const OrbxTileDesc dn = f3_tile(tiles, ntile);
      issue_rows(dn, nf, v);
      if (row_stat && wv == 0 && dn.ty > 0) pst = f3_probe_issue(stat_of(dn, nf), dead_of(dn, nf), dn.ty, lane);
    }

    // compaction of the candidate flags into the LDS queue, once per tile: wave prefix sum of the
    // popcounts + one LDS atomic per wave.  A queue entry = (walking thread, flag index).
    const uint32_t my_cands = cand;
    const uint32_t ebase = (uint32_t)tid << 5;
    const int cnt = __popc(cand);
    const int incl = wave_scan_incl(cnt);
    {
      int wbase = 0;
      if (lane == 63) {
        wbase = atomicAdd(&s_qn, incl);
        s_wtot[wv] = incl;
      }
      wbase = __builtin_amdgcn_readlane(wbase, 63);
      int pos = wbase + incl - cnt;
      while (cand) {
        const int bpos = __ffs(cand) - 1;
        cand &= cand - 1;
        if (pos < G::QCAP) s_queue[pos] = (uint16_t)(ebase | (uint32_t)bpos);
        pos++;
      }
    }
    __syncthreads();

    const int ntot = s_qn;
    auto eval = [&](int nq) {
      if (n == 9)
        f3_eval<R, 9>(s_img, score0, s_queue, nq, thr, n, keff, tid, s_lut);
      else if (n == 12)
        f3_eval<R, 12>(s_img, score0, s_queue, nq, thr, n, keff, tid, s_lut);
      else
        f3_eval<R, 0>(s_img, score0, s_queue, nq, thr, n, keff, tid, s_lut);
    };
    if (ntot <= G::QCAP) {  // block-uniform
      // ---- phase 3: full segment test + score of the candidates
      eval(ntot);
      __syncthreads();
      // ---- phase 4: NMS
      f3_nms<R>(score0, s_queue, ntot, s_mask32, th, tid);
    } else {
      // more candidates than the queue holds (noise, tiny thresholds): windows of QCAP candidates in
      // a fixed order (wave bases from the per-wave totals, not from the atomic)
      int wbase = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < wv) wbase += s_wtot[k];
      const int first = wbase + incl - cnt;
      auto fill_window = [&](int base) {
        int pos = first - base;
        for (uint32_t b = my_cands; b; b &= b - 1) {
          if (pos >= 0 && pos < G::QCAP) s_queue[pos] = (uint16_t)(ebase | (uint32_t)(__ffs(b) - 1));
          pos++;
        }
      };
      for (int base = 0; base < ntot; base += G::QCAP) {
        __syncthreads();
        fill_window(base);
        __syncthreads();
        eval(min(G::QCAP, ntot - base));
      }
      for (int base = 0; base < ntot; base += G::QCAP) {
        __syncthreads();
        fill_window(base);
        __syncthreads();
        // (entries are (thread, flag) again: positions are recomputed)
        const int nq = min(G::QCAP, ntot - base);
        for (int q = tid; q < nq; q += 256) {
          int sr, sc;
          f3_decode(s_queue[q], keff, sr, sc);
          s_queue[q] = (uint16_t)(sr * G::SC_PITCH + sc);
        }
        __syncthreads();
        f3_nms<R>(score0, s_queue, nq, s_mask32, th, tid);
      }
    }
    // the next tile's early-exit verdict (its statistics arrived long ago), published by the barrier below
    if (row_stat && have_next && wv == 0) {
      int nt2 = ntile;
      asm volatile("" : "+s"(nt2));  // (a fresh scalar load, see below)
      const OrbxTileDesc dn = f3_tile(tiles, nt2);
      bool skip = false;
      if (dn.ty > 0) skip = f3_probe_decide(pst, dead_of(dn, nf), dn.ty, dn.u2, dn.u0, lane);
      if (lane == 0) s_skip_next = skip;
    }
    __syncthreads();
    // (read here, not after the mask store: a wait for that store would sit in front of the read)
    const bool skip_next = row_stat && have_next ? __builtin_amdgcn_readfirstlane(s_skip_next) != 0 : false;

    // ---- phase 5 (first wave only): lane = tile row, its two mask words leave with one 16-byte store;
    // the tile's survivor count joins the tile-row statistics.  The tile record is loaded AGAIN (one
    // scalar load through a pointer the optimiser cannot see through): keeping its fields alive across
    // the arc test, which wants every SGPR, cost ~50 v_writelane / v_readlane spill instructions per wave.
    int t5 = tile;
    asm volatile("" : "+s"(t5));
    const OrbxTileDesc e = f3_tile(tiles, t5);
    const int th5 = e.f, y05 = e.ty * th5, h5 = e.h;
    if (tid < 64) {
      static_assert(G::TH <= 64 && G::MASK_DW == 4, "one lane per tile row, two 64-bit words per row");
      int surv = 0;
      const int gy = y05 + tid, gw = e.tx * 2, wpr = e.u1;
      if (tid < th5 && gy < h5) {
        const uint4 m = reinterpret_cast<const uint4*>(s_mask32)[tid];
        reinterpret_cast<uint4*>(s_mask32)[tid] = make_uint4(0u, 0u, 0u, 0u);
        u64* dst = mask + ((size_t)f * (size_t)mask_words + e.mask_off) + ((size_t)gy * wpr + gw);
        if (gw + 1 < wpr) {
          typedef uint32_t __attribute__((ext_vector_type(4), aligned(8))) u32x4_a8;
          *reinterpret_cast<u32x4_a8*>(dst) = u32x4_a8{m.x, m.y, m.z, m.w};
          surv = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
        } else {  // the level's last mask word is the tile's first
          *dst = (u64)m.x | ((u64)m.y << 32);
          surv = __popc(m.x) + __popc(m.y);
        }
      }
      if (row_stat) {
        const int ws = wave_sum(surv);
        if (tid == 0)
          __hip_atomic_fetch_add(row_stat + (size_t)f * ORBX_FAST_STAT_WORDS + e.stat_index + e.ty,
                                 (1ull << 32) | (u64)(uint32_t)ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (scores_out) {  // stage operator Fast()/orbx_fast_score: the dense score map (one frame)
      const int w5 = e.w, x05 = e.tx * G::TW;
      for (int i = tid; i < th5 * G::TW; i += 256) {
        const int iy = i / G::TW, ix = i - iy * G::TW;
        const int gy = y05 + iy, gx = x05 + ix;
        if (gy < h5 && gx < w5) scores_out[(size_t)gy * w5 + gx] = score0[(iy + R) * G::SC_PITCH + ix + 4];
      }
      __syncthreads();  // the score tile is cleared at the top of the next tile
    }

    if (!have_next) break;
    tile = ntile;
    f = nf;
    if (skip_next) {  // (rare while the tile rows that matter are being worked on; the rule in the dead zone)
      advance(tile, f);
      if (!search(tile, f)) break;
      issue_rows(f3_tile(tiles, tile), f, v);
    }
  }
}

template <int R>
void launch_fast3(long long total, hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames, const uint8_t* d_pyr,
                  int frame_bytes, int mask_words, OrbxFastParams fp, unsigned long long* d_mask, uint16_t* d_scores,
                  unsigned long long* d_row_stat, int chunk_scale) {
  // tiles per workgroup (ORBX_FAST_CHUNK: A/B timing; results do not depend on it).  Small launches keep
  // one tile per workgroup: the chip is not full anyway.  chunk_scale: the second launch of the
  // top-rows-first pipeline, whose tiles nearly all exit, takes four times as many per workgroup (a
  // workgroup probes four tiles per round trip).
  static const int chunk_env = [] {
    const char* e = getenv("ORBX_FAST_CHUNK");
    return e ? atoi(e) : 2;
  }();
  // (never so many per workgroup that fewer than ~8192 workgroups remain: the chip holds 1536 at once)
  const int wanted = std::min(std::max(chunk_env, 1) * std::max(chunk_scale, 1), 64);
  const int chunk = (int)std::max<long long>(1, std::min<long long>(wanted, total / 8192));
  const int wgs = (int)((total + chunk - 1) / chunk);
  hipLaunchKernelGGL((k_fast3<R>), dim3(wgs), dim3(256), 0, s, d_tiles, n_tiles, n_frames, chunk, d_pyr, frame_bytes,
                     mask_words, fp, d_mask, d_scores, d_row_stat);
}

}  // namespace

// d_tiles: the n_tiles tiles of ONE frame in band-major order (tile = ORBX_FAST3_TW x
// orbx_fast3_tile_h(nms_radius)); 1-D grid, a workgroup per `chunk` consecutive (tile, frame) indices.
// d_row_stat: n_frames * ORBX_FAST_STAT_WORDS zeroed u64 (or NULL: no early exit)
hipError_t orbx_launch_fast_nms(hipStream_t s, const OrbxTileDesc* d_tiles, int n_tiles, int n_frames,
                                const uint8_t* d_pyr, int frame_bytes, int mask_words, OrbxFastParams fp,
                                unsigned long long* d_mask, uint16_t* d_scores, unsigned long long* d_row_stat,
                                int chunk_scale) {
  if (n_tiles <= 0 || n_frames <= 0) return hipSuccess;
  const long long total = (long long)n_tiles * n_frames;
  if (total > 0x7fffffffll) return hipErrorInvalidValue;
  switch (fp.nms_radius) {
    case 0:
      launch_fast3<0>(total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_scores, d_row_stat,
                      chunk_scale);
      break;
    case 1:
      launch_fast3<1>(total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_scores, d_row_stat,
                      chunk_scale);
      break;
    case 2:
      launch_fast3<2>(total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_scores, d_row_stat,
                      chunk_scale);
      break;
    default:
      launch_fast3<3>(total, s, d_tiles, n_tiles, n_frames, d_pyr, frame_bytes, mask_words, fp, d_mask, d_scores, d_row_stat,
                      chunk_scale);
      break;
  }
  return hipGetLastError();
}
