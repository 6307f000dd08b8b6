// bw_probe.hip -- what this MI355X's memory system gives to the access shapes the ORB kernels use.
// Stand-alone (not part of liborbx): hipcc --offload-arch=gfx950 -O3 -o bw_probe tools/bw_probe.hip
//   ./bw_probe [MB]        (default 97 MB = the 64-frame KITTI pyramid; try 1600 to leave the 256 MiB Infinity Cache)
// Every kernel moves the SAME number of bytes: `bytes` read and/or `bytes` written, rows of 1280
// bytes (the KITTI level-0 pitch).  Also used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE per
// access width (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`): the true byte counts
// are printed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int PITCH = 1280;
constexpr int RH = 64;  // rows per wave, as the streaming blur

// a wave owns a column strip: lane -> W bytes of a row, RH rows top to bottom
template <int W, bool LOAD, bool STORE, int LANE_LO, int LANE_HI>
__global__ __launch_bounds__(256) void k_strip(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows) {
  typedef uint32_t vec_t __attribute__((ext_vector_type(W / 4)));
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int STRIP = (LANE_HI - LANE_LO + 1) * W;  // productive bytes per wave row
  constexpr int SPR = (PITCH + STRIP - 1) / STRIP;    // strips per row
  const int band = wave / SPR, strip = wave - band * SPR;
  const int y0 = band * RH;
  if (y0 >= rows) return;
  const int x = strip * STRIP + (lane - LANE_LO) * W;
  const bool ok = lane >= LANE_LO && lane <= LANE_HI && x >= 0 && x + W <= PITCH;
  vec_t acc = 0;
  for (int y = y0; y < min(y0 + RH, rows); y++) {
    const size_t off = (size_t)y * PITCH + x;
    vec_t v = 0;
    if (LOAD && (x >= 0 && x + W <= PITCH)) v = *reinterpret_cast<const vec_t*>(src + off);
    if (STORE) {
      if (ok) *reinterpret_cast<vec_t*>(dst + off) = v;
    } else {
      acc ^= v;
    }
  }
  if (!STORE && acc[0] == 0x12345678u) dst[0] = 1;  // keep the loads alive
}

template <int W, bool LOAD, bool STORE, int LANE_LO, int LANE_HI>
double run(const char* name, const uint8_t* src, uint8_t* dst, size_t bytes, hipStream_t s) {
  const int rows = (int)(bytes / PITCH);
  constexpr int STRIP = (LANE_HI - LANE_LO + 1) * W;
  constexpr int SPR = (PITCH + STRIP - 1) / STRIP;
  const int waves = SPR * ((rows + RH - 1) / RH);
  dim3 grid((waves + 3) / 4);
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_strip<W, LOAD, STORE, LANE_LO, LANE_HI>), grid, dim3(256), 0, s, src, dst, rows);
  const int reps = 20;
  CHK(hipEventRecord(a, s));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_strip<W, LOAD, STORE, LANE_LO, LANE_HI>), grid, dim3(256), 0, s, src, dst, rows);
  CHK(hipEventRecord(b, s));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  const double moved = (double)rows * PITCH * ((LOAD ? 1 : 0) + (STORE ? 1 : 0));
  printf("%-34s %8.1f us  %7.2f TB/s  (%.1f MB read, %.1f MB written per launch)\n", name, us, moved / us / 1e6,
         LOAD ? rows * (double)PITCH / 1e6 : 0.0, STORE ? rows * (double)PITCH / 1e6 : 0.0);
  return us;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 97;
  const size_t bytes = mb * 1000000 / PITCH * PITCH;
  uint8_t *src, *dst;
  CHK(hipMalloc(&src, bytes + 4096));
  CHK(hipMalloc(&dst, bytes + 4096));
  CHK(hipMemset(src, 1, bytes + 4096));
  CHK(hipMemset(dst, 0, bytes + 4096));
  hipStream_t s;
  CHK(hipStreamCreate(&s));
  printf("buffers: %zu MB each (Infinity Cache is 256 MiB), rows of %d bytes, %d rows per wave\n", mb, PITCH, RH);
  run<4, true, true, 0, 63>("copy  4 B/lane, 256-B wave rows", src, dst, bytes, s);
  run<4, true, true, 1, 62>("copy  4 B/lane, 248-B wave rows", src, dst, bytes, s);
  run<8, true, true, 0, 63>("copy  8 B/lane", src, dst, bytes, s);
  run<16, true, true, 0, 63>("copy 16 B/lane", src, dst, bytes, s);
  run<4, true, false, 0, 63>("read  4 B/lane", src, dst, bytes, s);
  run<8, true, false, 0, 63>("read  8 B/lane", src, dst, bytes, s);
  run<16, true, false, 0, 63>("read 16 B/lane", src, dst, bytes, s);
  run<4, false, true, 0, 63>("write 4 B/lane, 256-B wave rows", src, dst, bytes, s);
  run<4, false, true, 1, 62>("write 4 B/lane, 248-B wave rows", src, dst, bytes, s);
  run<16, false, true, 0, 63>("write 16 B/lane", src, dst, bytes, s);
  return 0;
}
