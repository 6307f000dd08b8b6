// orbx_wave.h -- wave64 cross-lane helpers shared by the kernel files (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short ss2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide integer sum, result in every lane.  DPP within the 16-lane rows
// (no LDS-crossbar round trips), then the four row sums are combined on the SALU.
__device__ __forceinline__ int wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141 /*row_half_mirror*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140 /*row_mirror*/, 0xf, 0xf, true);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// wave-wide inclusive prefix sum (lane i gets v_0 + ... + v_i) with DPP only
// (GCN cross-lane scan: row_shr 1/2/3, row_shr 4 and 8 with bank masks, then
// row_bcast 15 / 31 across the 16-lane rows)
__device__ __forceinline__ int wave_scan_incl(int v) {
  const int v0 = v;
  v += __builtin_amdgcn_update_dpp(0, v0, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v0, 0x112 /*row_shr:2*/, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v0, 0x113 /*row_shr:3*/, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xe, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xc, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}

// packed 16-bit lane arithmetic on a 32-bit register (v_pk_*_u16 / _i16)
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) + __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) - __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ss2_t, a), __builtin_bit_cast(ss2_t, b)));
}

// LDS hand-over between the lanes of ONE wave: its LDS operations execute in order, this only pins the compiler
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
