// orbx_fast_common.h -- pieces shared by the two FAST kernels (orbx_fast.hip: LDS tile kernel of the stage
// operators; orbx_fast4.hip: register-streaming kernel of the whole path).  gfx950 only.
//   reference semantics: src/orb_cpu.cpp:23-134
#pragma once
#include <hip/hip_runtime.h>

#include "orbx_internal.h"
#include "orbx_wave.h"

namespace {

// pre-test of the two pixels in the 16-bit lanes of `ip`: sign bit of a lane SET = not a candidate.
//   brighter >= 3  <=>  2nd smallest of (N,E,S,W) >= Ip + t      (src/orb_cpu.cpp:52-53,57)
//   darker   >= 3  <=>  2nd largest  of (N,E,S,W) <= Ip - t      (:54-55; with t == 0 the else-if
//   makes "darker" strict: cp < Ip, ZT)
template <bool ZT>
__device__ __forceinline__ uint32_t f3_pretest(uint32_t ip, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t T) {
  const uint32_t m1 = pk_min_u16(a, b), M1 = pk_max_u16(a, b);
  const uint32_t m2 = pk_min_u16(c, d), M2 = pk_max_u16(c, d);
  const uint32_t X = pk_max_u16(m1, m2), Y = pk_min_u16(M1, M2);
  const uint32_t lo2 = pk_min_u16(X, Y), hi2 = pk_max_u16(X, Y);
  const uint32_t br = pk_sub(lo2, ip);  // >= t  <=> at least 3 brighter
  uint32_t dk = pk_sub(ip, hi2);        // >= t  <=> at least 3 darker
  if (ZT) dk = pk_sub(dk, 0x00010001u);
  const uint32_t r = pk_max_i16(br, dk);
  return ZT ? r : pk_sub(r, T);
}

// run of >= n set bits in the circular 16-bit mask (any n; per-lane arithmetic)
__device__ __forceinline__ bool f3_has_run16(uint32_t m, int n) {
  const uint32_t x = m | (m << 16);
  uint32_t acc = x;
  int k = 1;
  while (2 * k <= n) {
    acc &= acc >> k;
    k *= 2;
  }
  if (k < n) acc &= acc >> (n - k);
  return (acc & 0xffffu) != 0;
}

// byte offsets of the 16 ring pixels (circle_offsets, src/orb_cpu.cpp:8-13) from pixel (x-3, y-3)
template <int PITCH>
struct F3Ring {
  static constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  static constexpr int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
  static constexpr int off(int k) { return (dy[k] + 3) * PITCH + dx[k] + 3; }
};

// ---- early exit of a tile, probed by ONE wave (all 64 lanes take part) --------------------------------
// Keypoints are kept in ROW-MAJOR order up to `cap` (src/orb_cpu.cpp:108-110, src/orb.cpp:63), so once the
// tile rows strictly above a tile are complete and already hold >= cap survivors, nothing in that tile can
// be among the first cap.  Such a tile stores nothing: its mask words are never looked at (the row-major
// walk of the selection kernel ignores everything after the first cap survivors).  The test reads completed
// statistics only, so it never depends on the order in which tiles run (a stale read just means "do the
// work"); the first tile that proves row b dead publishes 64-b in dead_from (monotone max) so that later
// tiles decide with one load.
// issue: lane < ty loads the statistic of tile row `lane`, lane 63 the level's "dead from" word
__device__ __forceinline__ u64 f3_probe_issue(const u64* stat, const u64* dead_from, int ty, int lane) {
  // ONE load instruction (two loads into the same register pair would be separated by a wait for every
  // outstanding load of the wave, the prefetched image rows included)
  u64 st = 0;
  const u64* p = lane == 63 ? dead_from : stat + lane;
  if (lane < ty || lane == 63) st = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return st;
}
__device__ __forceinline__ bool f3_probe_decide(u64 st, u64* dead_from, int ty, int tiles_x, int cap, int lane) {
  const int known = (int)__builtin_amdgcn_readlane((uint32_t)st, 63);  // 64 - (first dead row), 0: unknown
  bool skip = known >= 64 - ty;
  if (!skip) {
    const bool complete = lane >= ty || (int)(st >> 32) == tiles_x;
    const u64 inc = __ballot(!complete);                   // rows not yet complete
    const int k = inc ? __ffsll((long long)inc) - 1 : 64;  // first incomplete row
    const int surv = wave_sum(lane < k && lane < ty ? (int)(uint32_t)st : 0);
    skip = surv >= cap;
    if (skip && lane == 0)
      __hip_atomic_fetch_max(dead_from, (u64)(64 - ty), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return skip;
}

// A tile record, fetched with SCALAR loads: the table is never written on the device, which the constant
// address space tells the compiler (inside the tile loop a load through the plain pointer follows global
// stores of the previous tile and became a VECTOR load, with a wait for every outstanding load -- the
// prefetched image rows included -- in front of its first use).
__device__ __forceinline__ OrbxTileDesc f3_tile(const OrbxTileDesc* tiles, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const OrbxTileDesc __attribute__((address_space(4))) * CPtr;
  return *(CPtr)(uintptr_t)(tiles + i);
#else
  return tiles[i];  // (host pass of the single-source compile; never called)
#endif
}

}  // namespace
