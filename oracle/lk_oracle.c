/* lk_oracle.c -- CPU restatement of the pyramidal Lucas-Kanade tracker the reference
 * calls:
 *
 *   cv::calcOpticalFlowPyrLK(img1, img2, pts1, pts2, status, err, cv::Size(21,21), 3,
 *       cv::TermCriteria(COUNT + EPS, 30, 0.01))           src/feature_tracking.cpp:175-181
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h).
 *
 * PARITY UNPINNED.  The algorithm lives in OpenCV (imgproc/video: lkpyramid.cpp,
 * pyramids.cpp), which is absent from this image and from /root/reference; the
 * reference pins no OpenCV version (CMakeLists.txt: find_package(OpenCV REQUIRED)).
 * This file restates OpenCV 4.x's published algorithm:
 *   - buildOpticalFlowPyramid: level l+1 = pyrDown(level l) (5x5 [1 4 6 4 1]^2 / 256,
 *     rounded, BORDER_REFLECT_101, size (w+1)/2); levels stop when one would not be
 *     larger than the window; Scharr derivatives (3,10,3) as int16 pairs, REFLECT_101
 *     inside the image, zero outside it; images REFLECT_101-extended;
 *   - LKTrackerInvoker: 14-bit fixed-point bilinear weights, 5 fractional bits kept
 *     for the interpolated intensities, the 2x2 gradient matrix, minimum-eigenvalue
 *     test (1e-4), at most `max_iters` Newton steps per level, the epsilon and the
 *     oscillation (|delta + prev_delta| < 0.01) stopping rules, status and the
 *     L1 error at level 0.
 * One deliberate difference: OpenCV accumulates the window sums (A11, A12, A22, b1,
 * b2) in float, in an order that depends on its SIMD width; here (and in the HIP
 * kernel) they are accumulated EXACTLY as integers and rounded to float once, which
 * is within float rounding of any OpenCV summation order and makes the result
 * independent of the order.  The reference's own tests hold no vectors for this call.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

#define LK_W_BITS 14

static int lk_reflect(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

void oracle_lk_pyr_down(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst) {
  static const int k[5] = {1, 4, 6, 4, 1};
  const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++) {
      int sum = 0;
      for (int i = 0; i < 5; i++) {
        const uint8_t* row = src + (size_t)lk_reflect(2 * y + i - 2, sh) * sstride;
        int hs = 0;
        for (int j = 0; j < 5; j++) hs += k[j] * row[lk_reflect(2 * x + j - 2, sw)];
        sum += k[i] * hs;
      }
      dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
    }
}

void oracle_lk_scharr(const uint8_t* img, int w, int h, int stride, int16_t* deriv) {
  for (int y = 0; y < h; y++) {
    const uint8_t* r0 = img + (size_t)lk_reflect(y - 1, h) * stride;
    const uint8_t* r1 = img + (size_t)y * stride;
    const uint8_t* r2 = img + (size_t)lk_reflect(y + 1, h) * stride;
    for (int x = 0; x < w; x++) {
      const int xl = lk_reflect(x - 1, w), xr = lk_reflect(x + 1, w);
      const int t0l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, t0r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
      const int t1l = r2[xl] - r0[xl], t1c = r2[x] - r0[x], t1r = r2[xr] - r0[xr];
      deriv[((size_t)y * w + x) * 2] = (int16_t)(t0r - t0l);
      deriv[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1r + t1l) * 3 + t1c * 10);
    }
  }
}

typedef struct {
  int w, h;
  uint8_t* img;   /* tight, pitch w */
  int16_t* deriv; /* (dx, dy) pairs, tight; NULL for the `next` pyramid */
} lk_level;

static int lk_img(const lk_level* L, int x, int y) {
  return L->img[(size_t)lk_reflect(y, L->h) * L->w + lk_reflect(x, L->w)];
}
static int lk_d(const lk_level* L, int x, int y, int c) {
  if (x < 0 || y < 0 || x >= L->w || y >= L->h) return 0;
  return L->deriv[((size_t)y * L->w + x) * 2 + c];
}
static int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

static int lk_build(const uint8_t* img, int w, int h, int stride, int win, int max_level, int with_deriv,
                    lk_level* lv) {
  lv[0].w = w;
  lv[0].h = h;
  lv[0].img = (uint8_t*)malloc((size_t)w * h);
  for (int y = 0; y < h; y++) memcpy(lv[0].img + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  int top = 0;
  for (int l = 1; l <= max_level; l++) {
    const int nw = (lv[l - 1].w + 1) / 2, nh = (lv[l - 1].h + 1) / 2;
    if (nw <= win || nh <= win) break; /* buildOpticalFlowPyramid stops here */
    lv[l].w = nw;
    lv[l].h = nh;
    lv[l].img = (uint8_t*)malloc((size_t)nw * nh);
    oracle_lk_pyr_down(lv[l - 1].img, lv[l - 1].w, lv[l - 1].h, lv[l - 1].w, lv[l].img);
    top = l;
  }
  for (int l = 0; l <= top; l++) {
    lv[l].deriv = NULL;
    if (with_deriv) {
      lv[l].deriv = (int16_t*)malloc((size_t)lv[l].w * lv[l].h * 4);
      oracle_lk_scharr(lv[l].img, lv[l].w, lv[l].h, lv[l].w, lv[l].deriv);
    }
  }
  return top;
}

static void lk_weights(float a, float b, int* iw00, int* iw01, int* iw10, int* iw11) {
  *iw00 = (int)lrintf((1.f - a) * (1.f - b) * (float)(1 << LK_W_BITS));
  *iw01 = (int)lrintf(a * (1.f - b) * (float)(1 << LK_W_BITS));
  *iw10 = (int)lrintf((1.f - a) * b * (float)(1 << LK_W_BITS));
  *iw11 = (1 << LK_W_BITS) - *iw00 - *iw01 - *iw10;
}

int oracle_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, int stride_prev, int stride_next,
                    const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int win,
                    int max_level, int max_iters, double epsilon) {
  if (win < 3 || win > 31 || max_level < 0 || max_level > 7) return -1;
  if (max_iters < 0) max_iters = 0;
  if (max_iters > 100) max_iters = 100;
  if (epsilon < 0) epsilon = 0;
  if (epsilon > 10) epsilon = 10;
  epsilon *= epsilon;
  lk_level P[8], N[8];
  const int top = lk_build(prev, w, h, stride_prev, win, max_level, 1, P);
  lk_build(next, w, h, stride_next, win, max_level, 0, N);
  const float flt_scale = 1.f / (float)(1 << 20);
  const float half = (float)(win - 1) * 0.5f;
  int16_t* Iw = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)win * win);
  for (int i = 0; i < n; i++) {
    status[i] = 1;
    if (err) err[i] = 0.f;
  }
  for (int level = top; level >= 0; level--) {
    const lk_level* I = &P[level];
    const lk_level* J = &N[level];
    for (int i = 0; i < n; i++) {
      const float sc = (float)(1. / (double)(1 << level));
      float px = prev_pts[2 * i] * sc, py = prev_pts[2 * i + 1] * sc;
      float nx, ny;
      if (level == top) {
        nx = px;
        ny = py;
      } else {
        nx = next_pts[2 * i] * 2.f;
        ny = next_pts[2 * i + 1] * 2.f;
      }
      next_pts[2 * i] = nx;
      next_pts[2 * i + 1] = ny;
      px -= half;
      py -= half;
      const int ipx = (int)floorf(px), ipy = (int)floorf(py);
      if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
        if (level == 0) {
          status[i] = 0;
          if (err) err[i] = 0.f;
        }
        continue;
      }
      int iw00, iw01, iw10, iw11;
      lk_weights(px - (float)ipx, py - (float)ipy, &iw00, &iw01, &iw10, &iw11);
      int64_t sA11 = 0, sA12 = 0, sA22 = 0;
      for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
          const int X = ipx + x, Y = ipy + y;
          const int iv = lk_descale(lk_img(I, X, Y) * iw00 + lk_img(I, X + 1, Y) * iw01 + lk_img(I, X, Y + 1) * iw10 +
                                        lk_img(I, X + 1, Y + 1) * iw11,
                                    LK_W_BITS - 5);
          const int ix = lk_descale(lk_d(I, X, Y, 0) * iw00 + lk_d(I, X + 1, Y, 0) * iw01 + lk_d(I, X, Y + 1, 0) * iw10 +
                                        lk_d(I, X + 1, Y + 1, 0) * iw11,
                                    LK_W_BITS);
          const int iy = lk_descale(lk_d(I, X, Y, 1) * iw00 + lk_d(I, X + 1, Y, 1) * iw01 + lk_d(I, X, Y + 1, 1) * iw10 +
                                        lk_d(I, X + 1, Y + 1, 1) * iw11,
                                    LK_W_BITS);
          int16_t* o = Iw + 3 * ((size_t)y * win + x);
          o[0] = (int16_t)iv;
          o[1] = (int16_t)ix;
          o[2] = (int16_t)iy;
          sA11 += (int64_t)ix * ix;
          sA12 += (int64_t)ix * iy;
          sA22 += (int64_t)iy * iy;
        }
      const float A11 = (float)(double)sA11 * flt_scale, A12 = (float)(double)sA12 * flt_scale,
                  A22 = (float)(double)sA22 * flt_scale;
      float D = A11 * A22 - A12 * A12;
      const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if (min_eig < 1e-4f || D < FLT_EPSILON) {
        if (level == 0) status[i] = 0;
        continue;
      }
      D = 1.f / D;
      nx -= half;
      ny -= half;
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < max_iters; j++) {
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
          if (level == 0) status[i] = 0;
          break;
        }
        lk_weights(nx - (float)inx, ny - (float)iny, &iw00, &iw01, &iw10, &iw11);
        int64_t sb1 = 0, sb2 = 0;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            const int X = inx + x, Y = iny + y;
            const int16_t* o = Iw + 3 * ((size_t)y * win + x);
            const int diff = lk_descale(lk_img(J, X, Y) * iw00 + lk_img(J, X + 1, Y) * iw01 +
                                            lk_img(J, X, Y + 1) * iw10 + lk_img(J, X + 1, Y + 1) * iw11,
                                        LK_W_BITS - 5) -
                             o[0];
            sb1 += (int64_t)diff * o[1];
            sb2 += (int64_t)diff * o[2];
          }
        const float b1 = (float)(double)sb1 * flt_scale, b2 = (float)(double)sb2 * flt_scale;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nx += dx;
        ny += dy;
        next_pts[2 * i] = nx + half;
        next_pts[2 * i + 1] = ny + half;
        if ((double)dx * dx + (double)dy * dy <= epsilon) break;
        if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
          next_pts[2 * i] -= dx * 0.5f;
          next_pts[2 * i + 1] -= dy * 0.5f;
          break;
        }
        pdx = dx;
        pdy = dy;
      }
      if (status[i] && err && level == 0) {
        const float ex = next_pts[2 * i] - half, ey = next_pts[2 * i + 1] - half;
        const int inx = (int)floorf(ex), iny = (int)floorf(ey);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
          status[i] = 0;
          continue;
        }
        lk_weights(ex - (float)inx, ey - (float)iny, &iw00, &iw01, &iw10, &iw11);
        int64_t e = 0;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            const int X = inx + x, Y = iny + y;
            const int diff = lk_descale(lk_img(J, X, Y) * iw00 + lk_img(J, X + 1, Y) * iw01 +
                                            lk_img(J, X, Y + 1) * iw10 + lk_img(J, X + 1, Y + 1) * iw11,
                                        LK_W_BITS - 5) -
                             Iw[3 * ((size_t)y * win + x)];
            e += diff < 0 ? -diff : diff;
          }
        err[i] = (float)(double)e * (1.f / (float)(32 * win * win));
      }
    }
  }
  for (int l = 0; l <= top; l++) {
    free(P[l].img);
    free(P[l].deriv);
    free(N[l].img);
  }
  free(Iw);
  return top;
}
