/* orb_oracle.c -- CPU restatement of the reference ORB front-end (plain C).
 * TEST INFRASTRUCTURE ONLY; see orb_oracle.h for scope and pinning status.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, no FMA, glibc libm --
 * the same arithmetic the reference's g++ -O2 build performs).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

const int8_t oracle_pattern_31[1024] = {
#include "pattern_31.inc"
};

/* Bresenham ring of radius 3, clockwise from 12 o'clock  [orb_cpu.cpp:8-13] */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};

/* ------------------------------------------------------------------------ */
/* FAST + score  [orb_cpu.cpp:23-103]                                       */
void oracle_fast_score(const uint8_t* img, int w, int h, int stride, int threshold, int n, float* scores,
                       int64_t* n_pretest, int64_t* n_corners) {
  int64_t npre = 0, ncor = 0;
  memset(scores, 0, sizeof(float) * (size_t)w * (size_t)h); /* Mat::zeros, :29 */
  for (int y = 3; y < h - 3; y++) {
    for (int x = 3; x < w - 3; x++) { /* :34-35 */
      const int Ip = img[(size_t)y * stride + x];
      /* early rejection on ring pixels 0,4,8,12: need >=3 brighter or >=3 darker  :39-58 */
      static const int check_idx[4] = {0, 4, 8, 12};
      int brighter = 0, darker = 0;
      for (int k = 0; k < 4; k++) {
        const int i = check_idx[k];
        const int cp = img[(size_t)(y + RING_DY[i]) * stride + (x + RING_DX[i])];
        if (cp >= Ip + threshold)
          brighter++;
        else if (cp <= Ip - threshold)
          darker++;
      }
      if ((brighter > darker ? brighter : darker) < 3) continue;
      npre++;
      int cv[32]; /* :61-69 */
      for (int i = 0; i < 16; i++) {
        const int v = img[(size_t)(y + RING_DY[i]) * stride + (x + RING_DX[i])];
        cv[i] = v;
        cv[i + 16] = v;
      }
      for (int i = 0; i < 16; i++) { /* :73-101 */
        int all_b = 1, all_d = 1;
        for (int j = 0; j < n; j++) {
          const int v = cv[i + j];
          if (v < Ip + threshold) all_b = 0;
          if (v > Ip - threshold) all_d = 0;
        }
        if (all_b || all_d) {
          float score = 0.0f; /* :90-96 */
          for (int q = 0; q < 16; q++) score += (float)abs(Ip - cv[q]);
          scores[(size_t)y * w + x] = score;
          ncor++;
          break;
        }
      }
    }
  }
  if (n_pretest) *n_pretest = npre;
  if (n_corners) *n_corners = ncor;
}

/* NMS + cap  [orb_cpu.cpp:105-134] */
int oracle_nms(const float* scores, int w, int h, int nms_window, int nfeatures, int32_t* kps_xy,
               int64_t* total_out) {
  const int r = nms_window / 2; /* :105 */
  int count = 0;
  int64_t total = 0;
  for (int y = 3; y < h - 3; y++) {
    for (int x = 3; x < w - 3; x++) {
      const float s = scores[(size_t)y * w + x];
      if (s <= 0.0f) continue; /* :110 (cap handled below so the total can be reported) */
      int keep;
      if (r != 0) {
        /* cv::minMaxLoc over the (2r+1)^2 ROI  :114-124 */
        double maxv = scores[(size_t)(y - r) * w + (x - r)];
        for (int dy = -r; dy <= r; dy++)
          for (int dx = -r; dx <= r; dx++) {
            const double v = scores[(size_t)(y + dy) * w + (x + dx)];
            if (v > maxv) maxv = v;
          }
        keep = fabs((double)s - maxv) < 1e-6f; /* :126 */
      } else {
        keep = 1; /* :130-132 */
      }
      if (keep) {
        total++;
        if (count < nfeatures) { /* :110 `keypoints.size() >= nfeatures` */
          kps_xy[2 * count] = x;
          kps_xy[2 * count + 1] = y;
          count++;
        }
      }
    }
  }
  if (total_out) *total_out = total;
  return count;
}

int oracle_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int n, int nms_window,
                       int nfeatures, int32_t* kps_xy) {
  float* scores = (float*)malloc(sizeof(float) * (size_t)w * (size_t)h);
  oracle_fast_score(img, w, h, stride, threshold, n, scores, NULL, NULL);
  const int c = oracle_nms(scores, w, h, nms_window, nfeatures, kps_xy, NULL);
  free(scores);
  return c;
}

/* orientation  [orb_cpu.cpp:139-183] */
void oracle_orientations(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, int nkp,
                         int patch_size, float* angles) {
  const int pr = patch_size / 2; /* :142 */
  for (int i = 0; i < nkp; i++) {
    const int x = kps_xy[2 * i], y = kps_xy[2 * i + 1];
    if (x - pr < 0 || x + pr >= w || y - pr < 0 || y + pr >= h) { /* :152-156 */
      angles[i] = 0.0f;
      continue;
    }
    float m10 = 0.0f, m01 = 0.0f, m00 = 0.0f;
    for (int r = -pr; r <= pr; ++r)
      for (int c = -pr; c <= pr; ++c) { /* :162-176 */
        const float I = (float)img[(size_t)(y + r) * stride + (x + c)];
        m10 += (float)c * I;
        m01 += (float)r * I;
        m00 += I;
      }
    (void)m00;
    angles[i] = atan2f(m01, m10); /* :178 */
  }
}

/* cv::integral  [orb_cpu.cpp:207-208] */
void oracle_integral(const uint8_t* img, int w, int h, int stride, int32_t* integral) {
  const int W = w + 1;
  memset(integral, 0, sizeof(int32_t) * (size_t)W);
  for (int y = 0; y < h; y++) {
    int32_t row = 0;
    integral[(size_t)(y + 1) * W] = 0;
    for (int x = 0; x < w; x++) {
      row += img[(size_t)y * stride + x];
      integral[(size_t)(y + 1) * W + (x + 1)] = integral[(size_t)y * W + (x + 1)] + row;
    }
  }
}

/* sum5x5  [orb_cpu.cpp:190-201]; only called when the box is inside the image */
static int sum5x5_integral(const int32_t* I, int W, int x, int y) {
  const int x0 = x - 2, y0 = y - 2, x1 = x + 3, y1 = y + 3;
  return I[(size_t)y1 * W + x1] + I[(size_t)y0 * W + x0] - I[(size_t)y0 * W + x1] - I[(size_t)y1 * W + x0];
}

/* box sum over the zero-extended image (defined replacement for the D15 reads) */
static int sum5x5_zero_ext(const uint8_t* img, int w, int h, int stride, int x, int y) {
  int s = 0;
  for (int yy = y - 2; yy <= y + 2; yy++)
    for (int xx = x - 2; xx <= x + 2; xx++)
      if (xx >= 0 && yy >= 0 && xx < w && yy < h) s += img[(size_t)yy * stride + xx];
  return s;
}

/* rotated BRIEF  [orb_cpu.cpp:203-258] */
void oracle_brief(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, const float* angles,
                  int nkp, uint8_t* desc, uint8_t* valid, int64_t* n_skipped, int64_t* n_oob) {
  int32_t* I = (int32_t*)malloc(sizeof(int32_t) * (size_t)(w + 1) * (size_t)(h + 1));
  oracle_integral(img, w, h, stride, I);
  const int height = h + 1, width = w + 1; /* integral dims  :210-211 */
  int64_t nskip = 0, noob = 0;
  memset(desc, 0, (size_t)nkp * 32);
  if (valid) memset(valid, 0, (size_t)nkp * 32);
  for (int idx = 0; idx < nkp; idx++) {
    const int kx = kps_xy[2 * idx], ky = kps_xy[2 * idx + 1];
    const float angle = angles[idx];
    const float c = cosf(angle); /* :217-218 */
    const float s = sinf(angle);
    uint8_t* d = desc + (size_t)idx * 32;
    for (int i = 0; i < 256; i++) {
      const int x1 = oracle_pattern_31[i * 4], y1 = oracle_pattern_31[i * 4 + 1];
      const int x2 = oracle_pattern_31[i * 4 + 2], y2 = oracle_pattern_31[i * 4 + 3];
      const int dx1 = (int)lroundf(c * (float)x1 - s * (float)y1); /* :228-232 */
      const int dy1 = (int)lroundf(s * (float)x1 + c * (float)y1);
      const int dx2 = (int)lroundf(c * (float)x2 - s * (float)y2);
      const int dy2 = (int)lroundf(s * (float)x2 + c * (float)y2);
      const int cx1 = kx + dx1, cy1 = ky + dy1, cx2 = kx + dx2, cy2 = ky + dy2;
      const int sr = 5 / 2; /* :240-245 */
      if (cx1 < sr || cy1 < sr || cx1 > width - sr || cy1 > height - sr || cx2 < sr || cy2 < sr ||
          cx2 > width - sr || cy2 > height - sr) {
        nskip++;
        if (valid) valid[(size_t)idx * 32 + (i >> 3)] |= (uint8_t)(1u << (i & 7)); /* defined: bit stays 0 */
        continue;
      }
      const int in1 = (cx1 + 2 <= w - 1) && (cy1 + 2 <= h - 1);
      const int in2 = (cx2 + 2 <= w - 1) && (cy2 + 2 <= h - 1);
      int s1, s2;
      if (in1 && in2) {
        s1 = sum5x5_integral(I, width, cx1, cy1); /* :247-248 */
        s2 = sum5x5_integral(I, width, cx2, cy2);
        if (valid) valid[(size_t)idx * 32 + (i >> 3)] |= (uint8_t)(1u << (i & 7));
      } else {
        /* D15: the reference reads outside the integral image here */
        s1 = sum5x5_zero_ext(img, w, h, stride, cx1, cy1);
        s2 = sum5x5_zero_ext(img, w, h, stride, cx2, cy2);
        noob++;
      }
      if (s1 < s2) d[i >> 3] |= (uint8_t)(1u << (i & 7)); /* :250-252 */
    }
  }
  free(I);
  if (n_skipped) *n_skipped = nskip;
  if (n_oob) *n_oob = noob;
}

/* ORBCPU::detectAndCompute  [orb_cpu.cpp:271-276] */
int oracle_detect_and_compute_cpu(const uint8_t* img, int w, int h, int stride, int nfeatures, int threshold,
                                  int n, int nms_window, int patch_size, int32_t* kps_xy, float* angles,
                                  uint8_t* desc, uint8_t* valid) {
  const int c = oracle_fast_detect(img, w, h, stride, threshold, n, nms_window, nfeatures, kps_xy);
  oracle_orientations(img, w, h, stride, kps_xy, c, patch_size, angles);
  oracle_brief(img, w, h, stride, kps_xy, angles, c, desc, valid, NULL, NULL);
  return c;
}

/* ------------------------------------------------------------------------ */
/* stage arithmetic of the GPU flavour                                       */

int oracle_reflect101(int p, int len) { /* GaussianBlur1D.cu:27-32 */
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - p - 2;
  }
  return p;
}

/* convertTo(CV_8U): saturate_cast<uchar>(cvRound(v)), round-half-even */
static uint8_t to_u8_rne(float v) {
  long r = lrintf(v);
  if (r < 0) r = 0;
  if (r > 255) r = 255;
  return (uint8_t)r;
}

/* GaussianBlur1D  [GaussianBlur1D.cu:34-163].  Intent restated: the kernel's
 * own reflect101 is applied at every border (the reference's partial blocks
 * read zero-filled centre cells instead when W%32 or H%8 != 0 -- defect,
 * DESIGN.md D18). */
void oracle_blur5_sep(const uint8_t* img, int w, int h, int stride, uint8_t* dst, int dst_stride) {
  static const float k5[5] = {1, 4, 6, 4, 1}; /* :20 */
  float* tmp = (float*)malloc(sizeof(float) * (size_t)w * (size_t)h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float sum = 0.0f;
      for (int k = 0; k < 5; k++) sum += k5[k] * (float)img[(size_t)y * stride + oracle_reflect101(x - 2 + k, w)];
      tmp[(size_t)y * w + x] = sum / 16.0f; /* :67 */
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float sum = 0.0f;
      for (int k = 0; k < 5; k++) sum += k5[k] * tmp[(size_t)oracle_reflect101(y - 2 + k, h) * w + x];
      dst[(size_t)y * dst_stride + x] = to_u8_rne(sum / 16.0f); /* :104, :162 */
    }
  free(tmp);
}

/* GaussianBlur 5x5 /273  [GaussianBlur.cu:21-130] */
void oracle_blur5_273(const uint8_t* img, int w, int h, int stride, uint8_t* dst, int dst_stride) {
  static const float kf[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float sum = 0;
      for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++)
          sum += (float)img[(size_t)oracle_reflect101(y - 2 + i, h) * stride + oracle_reflect101(x - 2 + j, w)] *
                 kf[i * 5 + j];
      dst[(size_t)y * dst_stride + x] = to_u8_rne(sum / 273.0f); /* :67, :129 */
    }
}

/* createGaussianKernel  [GaussianBlur.cpp:7-37] */
void oracle_gaussian_kernel(int K, float sigma, float* kernel) {
  if (sigma <= 0.0f) sigma = 0.3f * ((K - 1) * 0.5f) + 0.8f; /* :15 */
  const int half = K / 2;
  float sum = 0.0f;
  for (int y = -half; y <= half; ++y)
    for (int x = -half; x <= half; ++x) {
      const float value = expf(-(float)(x * x + y * y) / (2 * sigma * sigma)); /* :25 */
      kernel[(y + half) * K + (x + half)] = value;
      sum += value;
    }
  for (int i = 0; i < K * K; ++i) kernel[i] /= sum; /* :32-34 */
}

/* d_conv2d arithmetic  [Convolution.cu:40-53] */
void oracle_conv2d_f32(const float* in, int w, int h, const float* kernel, int K, float* out) {
  const int wo = w - K + 1, ho = h - K + 1;
  for (int y = 0; y < ho; y++)
    for (int x = 0; x < wo; x++) {
      float sum = 0;
      for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) sum += in[(size_t)(y + i) * w + (x + j)] * kernel[i * K + j];
      out[(size_t)y * wo + x] = sum;
    }
}

/* conv2d wrapper  [Convolution.cu:57-101] */
void oracle_conv2d_u8(const uint8_t* in, int w, int h, int stride, const float* kernel, int K, uint8_t* out) {
  const int wo = w - K + 1, ho = h - K + 1;
  float* f = (float*)malloc(sizeof(float) * (size_t)w * (size_t)h);
  float* o = (float*)malloc(sizeof(float) * (size_t)wo * (size_t)ho);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) f[(size_t)y * w + x] = (float)in[(size_t)y * stride + x]; /* :62 */
  oracle_conv2d_f32(f, w, h, kernel, K, o);
  for (size_t i = 0; i < (size_t)wo * (size_t)ho; i++) out[i] = to_u8_rne(o[i]); /* :100 */
  free(f);
  free(o);
}

static uint8_t* pad_reflect101(const uint8_t* img, int w, int h, int stride, int r) {
  const int wp = w + 2 * r, hp = h + 2 * r;
  uint8_t* p = (uint8_t*)malloc((size_t)wp * (size_t)hp);
  for (int y = 0; y < hp; y++)
    for (int x = 0; x < wp; x++)
      p[(size_t)y * wp + x] = img[(size_t)oracle_reflect101(y - r, h) * stride + oracle_reflect101(x - r, w)];
  return p;
}

/* GaussianBlurCUDA  [GaussianBlur.cpp:39-49] */
void oracle_gaussian_blur_conv(const uint8_t* img, int w, int h, int stride, int K, uint8_t* dst) {
  float* kern = (float*)malloc(sizeof(float) * (size_t)K * (size_t)K);
  oracle_gaussian_kernel(K, -1.0f, kern);
  const int r = K / 2;
  uint8_t* p = pad_reflect101(img, w, h, stride, r); /* :44 */
  oracle_conv2d_u8(p, w + 2 * r, h + 2 * r, w + 2 * r, kern, K, dst);
  free(p);
  free(kern);
}

static const float SOBEL_X[9] = {-1.f, 0.f, 1.f, -2.f, 0.f, 2.f, -1.f, 0.f, 1.f}; /* Sobel.cpp:6-10 */
static const float SOBEL_Y[9] = {-1.f, -2.f, -1.f, 0.f, 0.f, 0.f, 1.f, 2.f, 1.f}; /* Sobel.cpp:12-16 */

/* SobelCUDA  [Sobel.cpp:18-32] */
void oracle_sobel_u8(const uint8_t* img, int w, int h, int stride, int dir, uint8_t* dst) {
  uint8_t* p = pad_reflect101(img, w, h, stride, 1);
  oracle_conv2d_u8(p, w + 2, h + 2, w + 2, dir == 0 ? SOBEL_X : SOBEL_Y, 3, dst);
  free(p);
}

/* float Sobel at one pixel (same accumulation order as d_conv2d) */
static void sobel_at(const uint8_t* img, int w, int h, int stride, int x, int y, float* gx, float* gy) {
  float sx = 0, sy = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const float v = (float)img[(size_t)oracle_reflect101(y - 1 + i, h) * stride + oracle_reflect101(x - 1 + j, w)];
      sx += v * SOBEL_X[i * 3 + j];
      sy += v * SOBEL_Y[i * 3 + j];
    }
  *gx = sx;
  *gy = sy;
}

/* Harris at keypoints -- intent of HarrisScore.cu:23-89 (see header) */
void oracle_harris(const uint8_t* img, int w, int h, int stride, const int32_t* kps_xy, int nkp, int window,
                   float k, float* out) {
  const int K = window, r = K / 2;
  float* g = (float*)malloc(sizeof(float) * (size_t)K * (size_t)K);
  oracle_gaussian_kernel(K, -1.0f, g); /* GaussianBlurCUDA -> createGaussianKernel(K) */
  for (int idx = 0; idx < nkp; idx++) {
    const int x = kps_xy[2 * idx], y = kps_xy[2 * idx + 1];
    float a = 0, b = 0, c = 0;
    for (int i = 0; i < K; i++)
      for (int j = 0; j < K; j++) {
        /* product images are REFLECT_101-padded before the blur  [GaussianBlur.cpp:44] */
        const int yy = oracle_reflect101(y - r + i, h), xx = oracle_reflect101(x - r + j, w);
        float gx, gy;
        sobel_at(img, w, h, stride, xx, yy, &gx, &gy);
        const float wgt = g[i * K + j];
        a += (gx * gx) * wgt; /* Sx2  :50,:55 */
        c += (gy * gy) * wgt; /* Sy2  :51,:56 */
        b += (gx * gy) * wgt; /* Sxy  :52,:57 (intent: from Ixy) */
      }
    const float det = a * c - b * b; /* :35 */
    const float trace = a + c;       /* :36 */
    out[idx] = det - k * trace * trace; /* :38 */
  }
  free(g);
}

/* ------------------------------------------------------------------------ */
/* pyramid + orchestrator                                                    */

float oracle_level_scale(float scale_factor, int level) {
  return (float)pow((double)scale_factor, (double)level); /* orb.cpp:95,117: float scale = pow(sf, i) */
}

void oracle_level_size(int w0, int h0, float scale_factor, int level, int* wl, int* hl) {
  if (level == 0) {
    *wl = w0;
    *hl = h0;
    return;
  }
  const float scale = oracle_level_scale(scale_factor, level);
  *wl = (int)round((double)((float)w0 / scale)); /* orb.cpp:118 */
  *hl = (int)round((double)((float)h0 / scale));
}

int oracle_level_quota(int nfeatures, float scale_factor, int nlevels, int level) {
  /* orb.cpp:62 with C++ promotion rules: 1/sf and 1-1/sf in float, pow in double */
  const float inv = 1 / scale_factor;
  const float num = 1 - inv;
  const double den = 1 - pow((double)inv, (double)nlevels);
  const double v = nfeatures * ((double)num / den) * pow((double)inv, (double)level);
  return (int)v;
}

void oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  int* yofs = (int*)malloc(sizeof(int) * (size_t)dh);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
  short* ibeta = (short*)malloc(sizeof(short) * 2 * (size_t)dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= sw - 1) {
      fx = 0;
      sx = sw - 1;
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)lrintf((1.f - fx) * 2048.f);
    ialpha[2 * dx + 1] = (short)lrintf(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = (short)lrintf((1.f - fy) * 2048.f);
    ibeta[2 * dy + 1] = (short)lrintf(fy * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 > sh - 1 ? sh - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 > sh - 1 ? sh - 1 : sy1);
    const uint8_t* S0 = src + (size_t)sy0 * sstride;
    const uint8_t* S1 = src + (size_t)sy1 * sstride;
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; dx++) {
      const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
      const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
      const int r0 = S0[sx] * a0 + S0[sx1] * a1;
      const int r1 = S1[sx] * a0 + S1[sx1] * a1;
      dst[(size_t)dy * dstride + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
  }
  free(xofs);
  free(yofs);
  free(ialpha);
  free(ibeta);
}

int oracle_select_top(const float* resp, int n, int keep, int32_t* idx_out) {
  /* rank by counting: deterministic total order (R desc, index asc) */
  const int m = n < keep ? n : keep;
  for (int i = 0; i < n; i++) {
    int rank = 0;
    for (int j = 0; j < n; j++)
      if (resp[j] > resp[i] || (resp[j] == resp[i] && j < i)) rank++;
    if (rank < m) idx_out[rank] = i;
  }
  return m;
}

void oracle_build_level(const uint8_t* img, int w, int h, int stride, const oracle_orb_params* p, int level,
                        uint8_t* dst) {
  int wl, hl;
  oracle_level_size(w, h, p->scale_factor, level, &wl, &hl);
  uint8_t* raw = (uint8_t*)malloc((size_t)wl * (size_t)hl);
  if (level == 0) {
    for (int y = 0; y < h; y++) memcpy(raw + (size_t)y * w, img + (size_t)y * stride, (size_t)w); /* orb.cpp:112 */
  } else {
    oracle_resize_linear(img, w, h, stride, raw, wl, hl, wl); /* orb.cpp:119 */
  }
  const int do_blur = (p->blur_levels == 2) || (p->blur_levels == 1 && level >= 1);
  if (do_blur) {
    if (p->blur_kind == 0)
      oracle_blur5_sep(raw, wl, hl, wl, dst, wl);
    else
      oracle_blur5_273(raw, wl, hl, wl, dst, wl);
  } else {
    memcpy(dst, raw, (size_t)wl * (size_t)hl);
  }
  free(raw);
}

int oracle_detect_and_compute_gpu(const uint8_t* img, int w, int h, int stride, const oracle_orb_params* p,
                                  int32_t* kps_xy, int32_t* kps_level_xy, int32_t* levels, float* angles,
                                  float* responses, uint8_t* desc, uint8_t* valid, int capacity) {
  int total = 0;
  for (int l = 0; l < p->nlevels; l++) { /* orb.cpp:61 */
    int wl, hl;
    oracle_level_size(w, h, p->scale_factor, l, &wl, &hl);
    uint8_t* lvl = (uint8_t*)malloc((size_t)wl * (size_t)hl);
    oracle_build_level(img, w, h, stride, p, l, lvl);
    const int quota = oracle_level_quota(p->nfeatures, p->scale_factor, p->nlevels, l); /* :62 */
    const int cap = 2 * quota;                                                         /* :63 */
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(cap > 0 ? cap : 1));
    const int nc = (wl >= 7 && hl >= 7 && cap > 0)
                       ? oracle_fast_detect(lvl, wl, hl, wl, p->threshold, p->n, p->nms_window, cap, cand)
                       : 0;
    float* R = (float*)malloc(sizeof(float) * (size_t)(nc > 0 ? nc : 1));
    oracle_harris(lvl, wl, hl, wl, cand, nc, p->harris_window, p->harris_k, R); /* :65 */
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int keep = oracle_select_top(R, nc, quota, sel); /* :67-86 intent */
    if (total + keep > capacity) keep = capacity - total;
    int32_t* kl = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(keep > 0 ? keep : 1));
    for (int i = 0; i < keep; i++) {
      kl[2 * i] = cand[2 * sel[i]];
      kl[2 * i + 1] = cand[2 * sel[i] + 1];
      responses[total + i] = R[sel[i]];
    }
    oracle_orientations(lvl, wl, hl, wl, kl, keep, p->patch_size, angles + total); /* :90 */
    oracle_brief(lvl, wl, hl, wl, kl, angles + total, keep, desc + (size_t)total * 32,
                 valid ? valid + (size_t)total * 32 : NULL, NULL, NULL); /* :91 (intent: pyramid[l], D10) */
    const float scale = oracle_level_scale(p->scale_factor, l); /* :95 */
    for (int i = 0; i < keep; i++) {
      if (kps_level_xy) {
        kps_level_xy[2 * (total + i)] = kl[2 * i];
        kps_level_xy[2 * (total + i) + 1] = kl[2 * i + 1];
      }
      if (levels) levels[total + i] = l;
      kps_xy[2 * (total + i)] = (int)((float)kl[2 * i] * scale); /* :96-97 */
      kps_xy[2 * (total + i) + 1] = (int)((float)kl[2 * i + 1] * scale);
    }
    total += keep;
    free(kl);
    free(sel);
    free(R);
    free(cand);
    free(lvl);
  }
  return total;
}

/* ------------------------------------------------------------------------ */
/* descriptor matching (next row)                      */

static int hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

void oracle_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* idx, int32_t* dist) {
  for (int i = 0; i < nq; i++) {
    int d1 = 1 << 30, d2 = 1 << 30, j1 = -1, j2 = -1;
    for (int j = 0; j < nt; j++) {
      const int d = hamming256(query + (size_t)i * 32, train + (size_t)j * 32);
      if (d < d1) {
        d2 = d1;
        j2 = j1;
        d1 = d;
        j1 = j;
      } else if (d < d2) {
        d2 = d;
        j2 = j;
      }
    }
    idx[2 * i] = j1;
    idx[2 * i + 1] = j2;
    dist[2 * i] = j1 >= 0 ? d1 : -1;
    dist[2 * i + 1] = j2 >= 0 ? d2 : -1;
  }
}

int oracle_match_ratio(const uint8_t* query, int nq, const uint8_t* train, int nt, double ratio,
                       int32_t* query_idx, int32_t* train_idx, int32_t* dist1) {
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nq > 0 ? nq : 1));
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nq > 0 ? nq : 1));
  oracle_knn2(query, nq, train, nt, idx, dist);
  int n = 0;
  for (int i = 0; i < nq; i++) {
    if (idx[2 * i + 1] < 0) continue; /* matches[i].size() < 2  (feature_matching.cpp:174) */
    const float m = (float)dist[2 * i], nn = (float)dist[2 * i + 1]; /* DMatch::distance is float */
    if (m < ratio * nn) { /* :177 */
      query_idx[n] = i;
      train_idx[n] = idx[2 * i];
      if (dist1) dist1[n] = dist[2 * i];
      n++;
    }
  }
  free(idx);
  free(dist);
  return n;
}
