// orbx_math.h -- bit-reproducible float transcendentals for the ORB hot path.
//
// Why this exists: the reference computes keypoint angles and the BRIEF
// rotation on the HOST with glibc (std::atan2 -> atan2f, orb_cpu.cpp:178;
// std::cos/std::sin -> cosf/sinf, orb_cpu.cpp:217-218; std::lround,
// orb_cpu.cpp:228-232).  Descriptor bits depend on lround(c*x - s*y), so a
// device libm that differs from glibc in the last ulp can flip bits.  The
// functions below restate the *published algorithms* glibc 2.35 (the libc of
// this image, Ubuntu 22.04) uses for these three functions, using only IEEE
// add/mul/div/convert operations, so that the same source compiled for
// gfx950 (with -ffp-contract=off) and for x86-64 produces identical bits:
//
//   * atan2f/atanf : Sun fdlibm float version (sysdeps/ieee754/flt-32/
//                    e_atan2f.c, s_atanf.c) -- pure binary32 arithmetic.
//   * sinf/cosf    : Arm "optimized-routines" single-precision sin/cos
//                    (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h)
//                    -- binary64 polynomial, rounded once to binary32.
//
// tests/test_math_vs_glibc.py proves (on the CPU, exhaustively over the
// reachable domain for sin/cos and on 10^8 integer moment pairs for atan2)
// that these functions return the same bits as the libm the oracle links.
//
// Domain: finite inputs; atan2 arguments are exact integers (|m| < 2^24),
// sin/cos arguments are atan2 results (|a| <= pi).  Outside that domain the
// functions still return sensible values but parity with glibc is untested.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ inline
#else
#define ORBX_HD static inline
#endif

#if defined(__clang__)
#define ORBX_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define ORBX_NO_CONTRACT
#endif

ORBX_HD uint32_t orbx_f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
ORBX_HD float orbx_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---- atanf (fdlibm float) -------------------------------------------------
ORBX_HD float orbx_atanf(float x) {
  ORBX_NO_CONTRACT
  // Constants are written as the decimal literals of the published source
  // (the hex values in its comments are not all exact), converted
  // double -> float exactly as a C compiler does for `static const float`.
  const float atanhi[4] = {(float)4.6364760399e-01, (float)7.8539812565e-01,
                           (float)9.8279368877e-01, (float)1.5707962513e+00};
  const float atanlo[4] = {(float)5.0121582440e-09, (float)3.7748947079e-08,
                           (float)3.4473217170e-08, (float)7.5497894159e-08};
  const float aT0 = (float)3.3333334327e-01, aT1 = (float)-2.0000000298e-01,
              aT2 = (float)1.4285714924e-01, aT3 = (float)-1.1111110449e-01,
              aT4 = (float)9.0908870101e-02, aT5 = (float)-7.6918758452e-02,
              aT6 = (float)6.6610731184e-02, aT7 = (float)-5.8335702866e-02,
              aT8 = (float)4.9768779427e-02, aT9 = (float)-3.6531571299e-02,
              aT10 = (float)1.6285819933e-02;
  const int32_t hx = (int32_t)orbx_f2u(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c800000) {  // |x| >= 2^26 (or NaN)
    if (ix > 0x7f800000) return x + x;
    float r = atanhi[3] + atanlo[3];
    return hx > 0 ? r : -r;
  }
  if (ix < 0x3ee00000) {    // |x| < 0.4375
    if (ix < 0x31000000) {  // |x| < 2^-29
      return x;
    }
    id = -1;
  } else {
    x = orbx_u2f((uint32_t)ix);  // fabsf
    if (ix < 0x3f980000) {       // |x| < 1.1875
      if (ix < 0x3f300000) {     // 7/16 <= |x| < 11/16
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {  // 11/16 <= |x| < 19/16
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else {
      if (ix < 0x401c0000) {  // |x| < 2.4375
        id = 2;
        x = (x - 1.5f) / (1.0f + 1.5f * x);
      } else {
        id = 3;
        x = -1.0f / x;
      }
    }
  }
  float z = x * x;
  float w = z * z;
  float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -z : z;
}

// ---- atan2f (fdlibm float), finite arguments ------------------------------
ORBX_HD float orbx_atan2f(float y, float x) {
  ORBX_NO_CONTRACT
  const float tiny = 1.0e-30f;
  const float pi_o_2 = (float)1.5707963705e+00;
  const float pi = (float)3.1415927410e+00;
  const float pi_lo = (float)-8.7422776573e-08;
  const int32_t hx = (int32_t)orbx_f2u(x);
  const int32_t hy = (int32_t)orbx_f2u(y);
  const int32_t ix = hx & 0x7fffffff;
  const int32_t iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
  if (hx == 0x3f800000) return orbx_atanf(y);            // x == 1.0
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     // 2*sign(x)+sign(y)
  if (iy == 0) {
    switch (m) {
      case 0:
      case 1:
        return y;
      case 2:
        return pi + tiny;
      default:
        return -pi - tiny;
    }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  // (infinite arguments are outside this path's domain; fdlibm's special
  //  cases for them are omitted)
  float z;
  const int32_t k = (iy - ix) >> 23;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60)
    z = 0.0f;
  else {
    float q = y / x;
    z = orbx_atanf(orbx_u2f(orbx_f2u(q) & 0x7fffffffu));
  }
  switch (m) {
    case 0:
      return z;
    case 1:
      return orbx_u2f(orbx_f2u(z) ^ 0x80000000u);
    case 2:
      return pi - (z - pi_lo);
    default:
      return (z - pi_lo) - pi;
  }
}

// ---- sinf / cosf (Arm optimized-routines, |x| < 120) ----------------------
// Evaluates sin (n even) or cos (n odd) of the reduced argument with the
// binary64 polynomial; `neg_cos` selects the table with negated cosine
// coefficients (quadrants 2,3).
ORBX_HD float orbx_sincos_poly(double x, double x2, int n, int neg_cos) {
  ORBX_NO_CONTRACT
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
               S3 = -0x1.994eb3774cf24p-13;
  double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
         C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
  if (neg_cos) {
    C0 = -C0;
    C1 = -C1;
    C2 = -C2;
    C3 = -C3;
    C4 = -C4;
  }
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = S2 + x2 * S3;
    double x7 = x3 * x2;
    double s = x + x3 * S1;
    return (float)(s + x7 * s1);
  } else {
    double x4 = x2 * x2;
    double c2 = C3 + x2 * C4;
    double c1 = C0 + x2 * C1;
    double x6 = x4 * x2;
    double c = c1 + x4 * C2;
    return (float)(c + x6 * c2);
  }
}

ORBX_HD uint32_t orbx_abstop12(float x) { return (orbx_f2u(x) >> 20) & 0x7ff; }

// Shared body: want_cos = 0 -> sinf, 1 -> cosf.  Valid for |y| < 120.
ORBX_HD float orbx_sincosf_impl(float y, int want_cos) {
  ORBX_NO_CONTRACT
  const double HPI_INV = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  const double HPI = 0x1.921FB54442D18p0;        // pi/2
  const double x = (double)y;
  const uint32_t top = orbx_abstop12(y);
  if (top < orbx_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
    const double x2 = x * x;
    if (top < orbx_abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
    return orbx_sincos_poly(x, x2, want_cos, 0);
  }
  // fast range reduction (|y| < 120)
  const double r = x * HPI_INV;
  const int n = ((int32_t)r + 0x800000) >> 24;
  const double xr = x - (double)n * HPI;
  const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  return orbx_sincos_poly(xr * sign, xr * xr, n ^ want_cos, (n & 2) != 0);
}

ORBX_HD float orbx_sinf(float y) { return orbx_sincosf_impl(y, 0); }
ORBX_HD float orbx_cosf(float y) { return orbx_sincosf_impl(y, 1); }

// lroundf semantics (round half away from zero) for |v| < 2^22, exact.
ORBX_HD int orbx_lroundf(float v) {
  ORBX_NO_CONTRACT
  float a = orbx_u2f(orbx_f2u(v) & 0x7fffffffu);
  int t = (int)a;                 // trunc
  float frac = a - (float)t;      // exact for |v| < 2^23
  if (frac >= 0.5f) t += 1;
  return ((int32_t)orbx_f2u(v) < 0) ? -t : t;
}
