/*
 * droid_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic of the reference's `droid_backends`
 * CUDA extension (ToniRV/NeRF-SLAM, the .cu files under /root/reference/src).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (nerf-slam_amd/) never does.
 *
 * Every function cites the reference lines it follows.  Per-pixel arithmetic
 * mirrors the reference expression by expression (float, with the same
 * float->double promotions the reference's literals cause); sums over pixels
 * are accumulated in double (the reference uses a 256-thread tree in float,
 * whose order a different device cannot reproduce anyway).
 *
 * Parity status: the reference ships no tests / golden vectors,
 * and its CUDA sources cannot be built here (Eigen + CUDA absent).  This file
 * is pinned by (tests/test_oracle_pins.py):
 *   - F.grid_sample identity for the lookups,
 *   - golden vectors generated from the reference's own PYTHON modules
 *     networks/geom/projective_ops.py, networks/geom/chol.py and
 *     networks/modules/corr.py (tools/gen_golden.py, fixtures in tests/golden),
 *   - finite differences of the projection for the Jacobians.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept.        */
/* The reference computes in c10::Half = (float op) then round to half        */
/* (correlation_kernels.cu:53-65 with scalar_t = at::Half).                   */
/* ------------------------------------------------------------------------- */
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t f;
  if (exp == 0) {
    if (man == 0) {
      f = sign;
    } else { /* subnormal */
      int e = -1;
      do { e++; man <<= 1; } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    f = sign | 0x7f800000u | (man << 13);
  } else {
    f = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &f, 4);
  return out;
}

static uint16_t f2h(float x) {
  uint32_t f;
  memcpy(&f, &x, 4);
  uint32_t sign = (f >> 16) & 0x8000u;
  uint32_t fexp = (f >> 23) & 0xff;
  uint32_t man = f & 0x7fffffu;
  if (fexp == 255) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  }
  int e = (int)fexp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (e <= 0) { /* subnormal or zero */
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    int shift = 14 - e; /* 14..24 */
    uint32_t hm = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13;
  uint32_t rem = man & 0x1fffu;
  uint16_t out = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) out++; /* may carry into exp: correct */
  return out;
}

uint16_t orc_f2h(float x) { return f2h(x); }
float orc_h2f(uint16_t h) { return h2f(h); }

static uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }
static uint16_t hadd(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); }

static int within(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

/* ------------------------------------------------------------------------- */
/* K12  corr_index_forward_kernel<at::Half>  (correlation_kernels.cu:20-70).  */
/* volume [B,h1,w1,h2,w2] half, coords [B,2,h1,w1] float,                     */
/* corr [B,2r+1,2r+1,h1,w1] half, zero-initialised as at :142-143.            */
/* The global `+=` sequence is kept in the reference's loop order.            */
/* ------------------------------------------------------------------------- */
void orc_corr_index_forward_f16(const uint16_t* volume, const float* coords, uint16_t* corr,
                                int B, int h1, int w1, int h2, int w2, int r) {
  const int rd = 2 * r + 1;
  const long HW1 = (long)h1 * w1;
  memset(corr, 0, sizeof(uint16_t) * (size_t)B * rd * rd * HW1);
  /* every (n, y, x) owns its outputs: threads change nothing in the arithmetic (bench.py times this on all host cores) */
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < B; n++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        float x0 = coords[((long)n * 2 + 0) * HW1 + (long)y * w1 + x];
        float y0 = coords[((long)n * 2 + 1) * HW1 + (long)y * w1 + x];
        float dx = x0 - floorf(x0);
        float dy = y0 - floorf(y0);
        const uint16_t* vol = volume + (((long)n * h1 + y) * w1 + x) * (long)h2 * w2;
        uint16_t* out = corr + (long)n * rd * rd * HW1 + (long)y * w1 + x;
#define CORR(i, j) out[((long)(i) * rd + (j)) * HW1]
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            int x1 = (int)floorf(x0) - r + i;
            int y1 = (int)floorf(y0) - r + j;
            if (within(y1, x1, h2, w2)) {
              uint16_t s = vol[(long)y1 * w2 + x1];
              if (i > 0 && j > 0) CORR(i - 1, j - 1) = hadd(CORR(i - 1, j - 1), hmul(s, f2h(dx * dy)));
              if (i > 0 && j < rd) CORR(i - 1, j) = hadd(CORR(i - 1, j), hmul(s, f2h(dx * (1.0f - dy))));
              if (i < rd && j > 0) CORR(i, j - 1) = hadd(CORR(i, j - 1), hmul(s, f2h((1.0f - dx) * dy)));
              if (i < rd && j < rd) CORR(i, j) = hadd(CORR(i, j), hmul(s, f2h((1.0f - dx) * (1.0f - dy))));
            }
          }
#undef CORR
      }
}

/* Same kernel, scalar_t = float (AT_DISPATCH_FLOATING_TYPES_AND_HALF, :145).
 * `corr += s * w` is a*b+c in float; nvcc contracts it to an FMA by default,
 * so fmaf is used here (unverifiable: no reference binary).                  */
void orc_corr_index_forward_f32(const float* volume, const float* coords, float* corr,
                                int B, int h1, int w1, int h2, int w2, int r) {
  const int rd = 2 * r + 1;
  const long HW1 = (long)h1 * w1;
  memset(corr, 0, sizeof(float) * (size_t)B * rd * rd * HW1);
  for (int n = 0; n < B; n++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        float x0 = coords[((long)n * 2 + 0) * HW1 + (long)y * w1 + x];
        float y0 = coords[((long)n * 2 + 1) * HW1 + (long)y * w1 + x];
        float dx = x0 - floorf(x0);
        float dy = y0 - floorf(y0);
        const float* vol = volume + (((long)n * h1 + y) * w1 + x) * (long)h2 * w2;
        float* out = corr + (long)n * rd * rd * HW1 + (long)y * w1 + x;
#define CORR(i, j) out[((long)(i) * rd + (j)) * HW1]
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            int x1 = (int)floorf(x0) - r + i;
            int y1 = (int)floorf(y0) - r + j;
            if (within(y1, x1, h2, w2)) {
              float s = vol[(long)y1 * w2 + x1];
              if (i > 0 && j > 0) CORR(i - 1, j - 1) = fmaf(s, dx * dy, CORR(i - 1, j - 1));
              if (i > 0 && j < rd) CORR(i - 1, j) = fmaf(s, dx * (1.0f - dy), CORR(i - 1, j));
              if (i < rd && j > 0) CORR(i, j - 1) = fmaf(s, (1.0f - dx) * dy, CORR(i, j - 1));
              if (i < rd && j < rd) CORR(i, j) = fmaf(s, (1.0f - dx) * (1.0f - dy), CORR(i, j));
            }
          }
#undef CORR
      }
}

/* ------------------------------------------------------------------------- */
/* A1  CorrBlock.corr + pyramid (networks/modules/corr.py:23-38,63-72).       */
/* fmap1,fmap2 [n,C,HW] half (channel-major as the reference's reshape :67-68) */
/* Each fmap is divided by 4.0 in half, the product matrix is rounded to half  */
/* (torch.matmul under autocast: fp32 accumulate, fp16 output; accumulation    */
/* order inside the BLAS is unspecified -> this oracle accumulates in double   */
/* and rounds once), then 3x avg_pool2d(2,2) in half with float accumulation   */
/* in (row, col) order and one rounding (ATen avg_pool2d: accscalar_t=float).  */
/* pyr0 [n,HW,h,w], pyr1 [n,HW,h/2,w/2], ...                                   */
/* ------------------------------------------------------------------------- */
void orc_corr_pool_f16(const uint16_t* in, uint16_t* out, long nslices, int h, int w) {
  int ho = h / 2, wo = w / 2;
#pragma omp parallel for schedule(static)
  for (long s = 0; s < nslices; s++) {
    const uint16_t* I = in + s * (long)h * w;
    uint16_t* O = out + s * (long)ho * wo;
    for (int y = 0; y < ho; y++)
      for (int x = 0; x < wo; x++) {
        float acc = 0.0f;
        acc += h2f(I[(2 * y) * w + 2 * x]);
        acc += h2f(I[(2 * y) * w + 2 * x + 1]);
        acc += h2f(I[(2 * y + 1) * w + 2 * x]);
        acc += h2f(I[(2 * y + 1) * w + 2 * x + 1]);
        O[y * wo + x] = f2h(acc / 4.0f);
      }
  }
}

void orc_corr_volume_f16(const uint16_t* fmap1, const uint16_t* fmap2, uint16_t* vol,
                         int n, int C, int HW) {
  float* a = (float*)malloc(sizeof(float) * (size_t)C * HW);
  float* b = (float*)malloc(sizeof(float) * (size_t)C * HW);
  for (int e = 0; e < n; e++) {
    for (long k = 0; k < (long)C * HW; k++) {
      a[k] = h2f(f2h(h2f(fmap1[(long)e * C * HW + k]) / 4.0f));
      b[k] = h2f(f2h(h2f(fmap2[(long)e * C * HW + k]) / 4.0f));
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < HW; p++)
      for (int q = 0; q < HW; q++) {
        double acc = 0.0;
        for (int c = 0; c < C; c++) acc += (double)a[(long)c * HW + p] * (double)b[(long)c * HW + q];
        vol[((long)e * HW + p) * HW + q] = f2h((float)acc);
      }
  }
  free(a);
  free(b);
}

/* ------------------------------------------------------------------------- */
/* K14  altcorr_forward_kernel<float>  (altcorr_kernel.cu:28-149).            */
/* fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last float, coords           */
/* [B,N,H1,W1,2], corr [B,N,rd*rd,H1,W1] zero-init (:299-301).                */
/* 32-channel slabs (:19,53); per slab a 32-long sequential dot (:98-100)     */
/* then four weighted `+=` (:112-142).  Output channel = iy + rd*ix (:102-105).*/
/* ------------------------------------------------------------------------- */
void orc_altcorr_forward_f32(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                             int B, int H1, int W1, int H2, int W2, int C, int N, int r) {
  const int rd = 2 * r + 1;
  const long HW1 = (long)H1 * W1;
  memset(corr, 0, sizeof(float) * (size_t)B * N * rd * rd * HW1);
  for (int b = 0; b < B; b++)
    for (int c0 = 0; c0 < C; c0 += 32)
      for (int n = 0; n < N; n++)
        for (int h1 = 0; h1 < H1; h1++)
          for (int w1 = 0; w1 < W1; w1++) {
            const float* cp = coords + ((((long)b * N + n) * H1 + h1) * W1 + w1) * 2;
            float x2 = cp[0], y2 = cp[1];
            float dx = x2 - floorf(x2);
            float dy = y2 - floorf(y2);
            const float* f1 = fmap1 + (((long)b * H1 + h1) * W1 + w1) * C + c0;
            float* out = corr + ((long)b * N + n) * rd * rd * HW1 + (long)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                int h2 = (int)floorf(y2) - r + iy;
                int w2 = (int)floorf(x2) - r + ix;
                float s = 0.0f;
                if (within(h2, w2, H2, W2)) {
                  const float* f2 = fmap2 + (((long)b * H2 + h2) * W2 + w2) * C + c0;
                  int kmax = (C - c0 < 32) ? C - c0 : 32;
                  for (int k = 0; k < kmax; k++) s = fmaf(f1[k], f2[k], s);
                }
                float nw = s * (dy * dx);
                float ne = s * (dy * (1 - dx));
                float sw = s * ((1 - dy) * dx);
                float se = s * ((1 - dy) * (1 - dx));
                if (iy > 0 && ix > 0) out[((long)(iy - 1) + rd * (ix - 1)) * HW1] += nw;
                if (iy > 0 && ix < rd) out[((long)(iy - 1) + rd * ix) * HW1] += ne;
                if (iy < rd && ix > 0) out[((long)iy + rd * (ix - 1)) * HW1] += sw;
                if (iy < rd && ix < rd) out[((long)iy + rd * ix) * HW1] += se;
              }
          }
}

/* ------------------------------------------------------------------------- */
/* SE3 helpers  (droid_kernels.cu:66-188, 994-1012).  q = [x,y,z,w].          */
/* ------------------------------------------------------------------------- */
static void actSO3(const float* q, const float* X, float* Y) {
  float uv[3];
  uv[0] = (float)(2.0 * (q[1] * X[2] - q[2] * X[1]));
  uv[1] = (float)(2.0 * (q[2] * X[0] - q[0] * X[2]));
  uv[2] = (float)(2.0 * (q[0] * X[1] - q[1] * X[0]));
  float y0 = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  float y1 = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  float y2 = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

static void actSE3(const float* t, const float* q, const float* X, float* Y) {
  float x3 = X[3];
  actSO3(q, X, Y);
  Y[3] = x3;
  Y[0] += x3 * t[0];
  Y[1] += x3 * t[1];
  Y[2] += x3 * t[2];
}

/* droid_kernels.cu:88-105.  X and Y may alias; the reference reads X[0..2]
 * (already overwritten when X==Y) for the cross product -- reproduced.       */
static void adjSE3(const float* t, const float* q, const float* X, float* Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  actSO3(qinv, &X[0], &Y[0]);
  actSO3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  actSO3(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}

static void relSE3(const float* ti, const float* qi, const float* tj, const float* qj, float* tij, float* qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}

static void expSO3(const float* phi, float* q) {
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq;
  float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8) {
    imag = (float)(0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4);
    real = (float)(1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4);
  } else {
    imag = sinf((float)(0.5 * theta)) / theta;
    real = cosf((float)(0.5 * theta));
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

static void crossInplace(const float* a, float* b) {
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}

static void expSE3(const float* xi, float* t, float* q) {
  expSO3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4) {
    float a = (1 - cosf(theta)) / theta_sq;
    crossInplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    float b = (theta - sinf(theta)) / (theta * theta_sq);
    crossInplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}

/* retrSE3 (droid_kernels.cu:994-1012): T1 = Exp(xi) * T, xi = [tau, phi].    */
static void retrSE3(const float* xi, const float* t, const float* q, float* t1, float* q1) {
  float dt[3] = {0, 0, 0};
  float dq[4] = {0, 0, 0, 1};
  expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  actSO3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

/* exported thin wrappers so tests can pin the group algebra directly */
void orc_relSE3(const float* pi, const float* pj, float* out7) { relSE3(pi, pi + 3, pj, pj + 3, out7, out7 + 3); }
void orc_actSE3(const float* p, const float* X, float* Y) { actSE3(p, p + 3, X, Y); }
void orc_adjSE3(const float* p, const float* X, float* Y) { adjSE3(p, p + 3, X, Y); }
void orc_expSE3(const float* xi, float* out7) { expSE3(xi, out7, out7 + 3); }

/* pose_retr_kernel (droid_kernels.cu:1015-1048) */
void orc_pose_retr(float* poses, const float* dx, int kf0, int kf1) {
  for (int k = kf0; k < kf1; k++) {
    float t1[3], q1[4];
    retrSE3(dx + (long)(k - kf0) * 6, poses + (long)k * 7, poses + (long)k * 7 + 3, t1, q1);
    memcpy(poses + (long)k * 7, t1, 12);
    memcpy(poses + (long)k * 7 + 3, q1, 16);
  }
}

/* ------------------------------------------------------------------------- */
/* K3  frame_distance_kernel (droid_kernels.cu:630-769).                      */
/* ------------------------------------------------------------------------- */
void orc_frame_distance(const float* poses, const float* disps, const float* intr,
                        const int64_t* ii, const int64_t* jj, float* dist,
                        int num, int ht, int wd, float beta) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  for (int b = 0; b < num; b++) {
    int ix = (int)ii[b], jx = (int)jj[b];
    float tij[3], qij[4];
    relSE3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij, qij);
    double accum = 0, valid = 0, total = 0;
    for (int k = 0; k < ht * wd; k++) {
      const int i = k / wd, j = k % wd;
      const float u = (float)j, v = (float)i;
      float Xi[4], Xj[4];
      Xi[0] = (u - cx) / fx;
      Xi[1] = (v - cy) / fy;
      Xi[2] = 1;
      Xi[3] = disps[((long)ix * ht + i) * wd + j];
      actSE3(tij, qij, Xi, Xj);
      float du = fx * (Xj[0] / Xj[2]) + cx - u;
      float dv = fy * (Xj[1] / Xj[2]) + cy - v;
      float d = sqrtf(du * du + dv * dv);
      total += beta;
      if (Xj[2] > 0.25) {
        accum += beta * d;
        valid += beta;
      }
      Xj[0] = Xi[0] + Xi[3] * tij[0];
      Xj[1] = Xi[1] + Xi[3] * tij[1];
      Xj[2] = Xi[2] + Xi[3] * tij[2];
      du = fx * (Xj[0] / Xj[2]) + cx - u;
      dv = fy * (Xj[1] / Xj[2]) + cy - v;
      d = sqrtf(du * du + dv * dv);
      total += (1 - beta);
      if (Xj[2] > 0.25) {
        accum += (1 - beta) * d;
        valid += (1 - beta);
      }
    }
    float accf = (float)accum, validf = (float)valid, totalf = (float)total;
    dist[b] = (validf / (totalf + 1e-8) < 0.75) ? 1000.0f : accf / validf;
  }
}

/* ------------------------------------------------------------------------- */
/* K1  projective_transform_kernel (droid_kernels.cu:192-536).                */
/* targets, weights [M,2,ht,wd]; poses [*,7]; disps [*,ht,wd]; intr[4];       */
/* extr[7]; outputs Hs[4,M,6,6], vs[2,M,6], Eiz,Ejz [M,6,HW], Cii,bz [M,HW].  */
/* ------------------------------------------------------------------------- */
static void reorder_wt(float* J) { /* [t,w] -> [w,t]  (:387-403) */
  float c[6];
  memcpy(c, J, 24);
  J[0] = c[3]; J[1] = c[4]; J[2] = c[5]; J[3] = c[0]; J[4] = c[1]; J[5] = c[2];
}

void orc_projective_transform(const float* target, const float* weight, const float* poses,
                              const float* disps, const float* intr, const float* extr,
                              const int64_t* ii, const int64_t* jj, int M, int ht, int wd,
                              float* Hs, float* vs, float* Eiz, float* Ejz, float* Cii, float* bz) {
  const int HW = ht * wd;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float* cTb_t = extr;
  const float* cTb_q = extr + 3;
#pragma omp parallel for schedule(static) /* per-edge outputs are disjoint */
  for (int e = 0; e < M; e++) {
    const int ix = (int)ii[e], jx = (int)jj[e];
    float tij[3], qij[4];
    if (ix == jx) { /* stereo (:249-259) */
      tij[0] = -0.1f; tij[1] = 0; tij[2] = 0;
      qij[0] = 0; qij[1] = 0; qij[2] = 0; qij[3] = 1;
    } else {
      relSE3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij, qij);
    }
    double hij[78];
    double vi[6], vj[6];
    for (int l = 0; l < 78; l++) hij[l] = 0;
    for (int n = 0; n < 6; n++) { vi[n] = 0; vj[n] = 0; }

    for (int k = 0; k < HW; k++) {
      const int i = k / wd, j = k % wd;
      const float u = (float)j, v = (float)i;
      float Xi[4], Xj[4], Jx[12], Jz;
      float* Ji = &Jx[0];
      float* Jj = &Jx[6];
      Xi[0] = (u - cx) / fx;
      Xi[1] = (v - cy) / fy;
      Xi[2] = 1;
      Xi[3] = disps[(long)ix * HW + k];
      actSE3(tij, qij, Xi, Xj);
      const float x = Xj[0], y = Xj[1], h = Xj[3];
      const float d = (Xj[2] < 0.25) ? 0.0f : (float)(1.0 / Xj[2]);
      const float d2 = d * d;
      float weight_u = (Xj[2] < 0.25) ? 0.0f : (float)(.001 * weight[((long)e * 2 + 0) * HW + k]);
      float weight_v = (Xj[2] < 0.25) ? 0.0f : (float)(.001 * weight[((long)e * 2 + 1) * HW + k]);
      const float residual_u = target[((long)e * 2 + 0) * HW + k] - (fx * d * x + cx);
      const float residual_v = target[((long)e * 2 + 1) * HW + k] - (fy * d * y + cy);

      /* ---- u row (:361-421) ---- */
      Jz = fx * (tij[0] * d - tij[2] * (x * d2));
      float C = weight_u * Jz * Jz;
      float b = weight_u * residual_u * Jz;
      if (ix == jx) weight_u = 0;
      Jj[0] = fx * (h * d);
      Jj[1] = (float)(fx * 0.0);
      Jj[2] = fx * (-x * h * d2);
      Jj[3] = fx * (-x * y * d2);
      Jj[4] = (float)(fx * (1.0 + x * x * d2));
      Jj[5] = fx * (-y * d);
      adjSE3(tij, qij, Jj, Ji);
      for (int n = 0; n < 6; n++) Ji[n] = (float)(Ji[n] * -1.0);
      adjSE3(cTb_t, cTb_q, Jj, Jj);
      adjSE3(cTb_t, cTb_q, Ji, Ji);
      for (int n = 0; n < 6; n++) Jj[n] = (float)(Jj[n] * -1.0);
      for (int n = 0; n < 6; n++) Ji[n] = (float)(Ji[n] * -1.0);
      reorder_wt(Jj);
      reorder_wt(Ji);
      int l = 0;
      for (int n = 0; n < 12; n++)
        for (int m = 0; m <= n; m++) { hij[l] += (double)(weight_u * Jx[n] * Jx[m]); l++; }
      float Ei_[6], Ej_[6];
      for (int n = 0; n < 6; n++) {
        vi[n] += (double)(weight_u * residual_u * Ji[n]);
        vj[n] += (double)(weight_u * residual_u * Jj[n]);
        Ei_[n] = weight_u * Jz * Ji[n];
        Ej_[n] = weight_u * Jz * Jj[n];
      }

      /* ---- v row (:428-487) ---- */
      Jz = fy * (tij[1] * d - tij[2] * (y * d2));
      C += weight_v * Jz * Jz;
      b += weight_v * residual_v * Jz;
      if (ix == jx) weight_v = 0;
      Jj[0] = fy * 0;
      Jj[1] = fy * (h * d);
      Jj[2] = fy * (-y * h * d2);
      Jj[3] = fy * (-1 - y * y * d2);
      Jj[4] = fy * (x * y * d2);
      Jj[5] = fy * (x * d);
      adjSE3(tij, qij, Jj, Ji);
      for (int n = 0; n < 6; n++) Ji[n] = (float)(Ji[n] * -1.0);
      adjSE3(cTb_t, cTb_q, Jj, Jj);
      adjSE3(cTb_t, cTb_q, Ji, Ji);
      for (int n = 0; n < 6; n++) Jj[n] = (float)(Jj[n] * -1.0);
      for (int n = 0; n < 6; n++) Ji[n] = (float)(Ji[n] * -1.0);
      reorder_wt(Jj);
      reorder_wt(Ji);
      l = 0;
      for (int n = 0; n < 12; n++)
        for (int m = 0; m <= n; m++) { hij[l] += (double)(weight_v * Jx[n] * Jx[m]); l++; }
      for (int n = 0; n < 6; n++) {
        vi[n] += (double)(weight_v * residual_v * Ji[n]);
        vj[n] += (double)(weight_v * residual_v * Jj[n]);
        Ei_[n] += weight_v * Jz * Ji[n];
        Ej_[n] += weight_v * Jz * Jj[n];
      }
      Cii[(long)e * HW + k] = C;
      bz[(long)e * HW + k] = b;
      for (int n = 0; n < 6; n++) {
        Eiz[((long)e * 6 + n) * HW + k] = Ei_[n];
        Ejz[((long)e * 6 + n) * HW + k] = Ej_[n];
      }
    }
    for (int n = 0; n < 6; n++) {
      vs[((long)0 * M + e) * 6 + n] = (float)vi[n];
      vs[((long)1 * M + e) * 6 + n] = (float)vj[n];
    }
    int l = 0;
#define HS(b4, r, c) Hs[(((long)(b4) * M + e) * 6 + (r)) * 6 + (c)]
    for (int n = 0; n < 12; n++)
      for (int m = 0; m <= n; m++) {
        float s = (float)hij[l];
        if (n < 6 && m < 6) { HS(0, n, m) = s; HS(0, m, n) = s; }
        else if (n >= 6 && m < 6) { HS(1, m, n - 6) = s; HS(2, n - 6, m) = s; }
        else { HS(3, n - 6, m - 6) = s; HS(3, m - 6, n - 6) = s; }
        l++;
      }
#undef HS
  }
}

/* debug export: per-pixel reprojection + Jacobians of one edge, in the SAME
 * convention K1 uses after its sign flip/reorder (used to pin against the
 * reference's networks/geom/projective_ops.py:98-145).                        */
void orc_edge_jacobians(const float* pose_i, const float* pose_j, const float* disp, const float* intr,
                        const float* extr, int ht, int wd, float* coords, float* Ji_out, float* Jj_out,
                        float* Jz_out) {
  const int HW = ht * wd;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float tij[3], qij[4];
  relSE3(pose_i, pose_i + 3, pose_j, pose_j + 3, tij, qij);
  for (int k = 0; k < HW; k++) {
    const int i = k / wd, j = k % wd;
    float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1, disp[k]}, Xj[4];
    actSE3(tij, qij, Xi, Xj);
    const float x = Xj[0], y = Xj[1], h = Xj[3];
    const float d = (Xj[2] < 0.25) ? 0.0f : (float)(1.0 / Xj[2]);
    const float d2 = d * d;
    coords[(long)k * 2 + 0] = fx * d * x + cx;
    coords[(long)k * 2 + 1] = fy * d * y + cy;
    for (int row = 0; row < 2; row++) {
      float Ji[6], Jj[6];
      if (row == 0) {
        Jz_out[(long)k * 2 + 0] = fx * (tij[0] * d - tij[2] * (x * d2));
        Jj[0] = fx * (h * d); Jj[1] = 0; Jj[2] = fx * (-x * h * d2);
        Jj[3] = fx * (-x * y * d2); Jj[4] = (float)(fx * (1.0 + x * x * d2)); Jj[5] = fx * (-y * d);
      } else {
        Jz_out[(long)k * 2 + 1] = fy * (tij[1] * d - tij[2] * (y * d2));
        Jj[0] = 0; Jj[1] = fy * (h * d); Jj[2] = fy * (-y * h * d2);
        Jj[3] = fy * (-1 - y * y * d2); Jj[4] = fy * (x * y * d2); Jj[5] = fy * (x * d);
      }
      adjSE3(tij, qij, Jj, Ji);
      for (int n = 0; n < 6; n++) Ji[n] = -Ji[n];
      adjSE3(extr, extr + 3, Jj, Jj);
      adjSE3(extr, extr + 3, Ji, Ji);
      for (int n = 0; n < 6; n++) { Jj[n] = -Jj[n]; Ji[n] = -Ji[n]; }
      reorder_wt(Jj);
      reorder_wt(Ji);
      memcpy(Ji_out + ((long)k * 2 + row) * 6, Ji, 24);
      memcpy(Jj_out + ((long)k * 2 + row) * 6, Jj, 24);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* K6  accum_cuda + accum_kernel (droid_kernels.cu:971-991,1065-1115).        */
/* out[j,:] = sum_{i: ix[i]==jx[j]} data[i,:]   (jx ascending).               */
/* ------------------------------------------------------------------------- */
static void accum(const float* data, const int64_t* ix, int nrows, const int64_t* jx, int count, long D, float* out) {
  for (int j = 0; j < count; j++) {
    float* o = out + (long)j * D;
    for (long k = 0; k < D; k++) o[k] = 0;
    for (int i = 0; i < nrows; i++)
      if (ix[i] == jx[j]) {
        const float* s = data + (long)i * D;
        for (long k = 0; k < D; k++) o[k] += s[k];
      }
  }
}
void orc_accum(const float* data, const int64_t* ix, int nrows, const int64_t* jx, int count, long D, float* out) {
  accum(data, ix, nrows, jx, count, D, out);
}

static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* torch::_unique(x, sorted=true, return_inverse=true) */
static int unique_inverse(const int64_t* x, int n, int64_t* uniq, int64_t* inv) {
  int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * n);
  memcpy(tmp, x, sizeof(int64_t) * n);
  qsort(tmp, n, sizeof(int64_t), cmp_i64);
  int K = 0;
  for (int i = 0; i < n; i++)
    if (i == 0 || tmp[i] != tmp[i - 1]) uniq[K++] = tmp[i];
  for (int i = 0; i < n; i++)
    for (int k = 0; k < K; k++)
      if (uniq[k] == x[i]) { inv[i] = k; break; }
  free(tmp);
  return K;
}

/* ------------------------------------------------------------------------- */
/* A5  reduced_camera_matrix_cuda (droid_kernels.cu:1681-1768) including      */
/* SparseBlock (:1240-1316, fp64 accumulate, column-major get_dense read as   */
/* row-major => the returned H is the transpose of A-S) and schur_block       */
/* (:1349-1438) with EEt6x6 (:1118-1173) and Ev6x1 (:1176-1210).              */
/* Outputs: H[6P*6P], v[6P], Q[K'*HW], E[(P+M)*6*HW], w[K'*HW], kx[K'].       */
/* eta has K' rows.  Returns K'.                                              */
/* ------------------------------------------------------------------------- */
int orc_reduced_camera_matrix(const float* poses, const float* disps, const float* intr, const float* extr,
                              const float* disps_sens, const float* targets, const float* weights,
                              const float* eta, const int64_t* ii, const int64_t* jj, int M, int ht, int wd,
                              int kf0, int kf1, float* H, float* v, float* Q, float* E, float* w,
                              int64_t* kx_out) {
  const int HW = ht * wd;
  const int P = kf1 - kf0;
  const int NE = P + M;
  int64_t* ii_exp = (int64_t*)malloc(sizeof(int64_t) * NE);
  int64_t* jj_exp = (int64_t*)malloc(sizeof(int64_t) * NE);
  int64_t* ts = (int64_t*)malloc(sizeof(int64_t) * (P > 0 ? P : 1));
  for (int t = 0; t < P; t++) { ts[t] = kf0 + t; ii_exp[t] = kf0 + t; jj_exp[t] = kf0 + t; }
  for (int e = 0; e < M; e++) { ii_exp[P + e] = ii[e]; jj_exp[P + e] = jj[e]; }
  int64_t* kx = (int64_t*)malloc(sizeof(int64_t) * NE);
  int64_t* kk = (int64_t*)malloc(sizeof(int64_t) * NE);
  const int K = unique_inverse(ii_exp, NE, kx, kk);
  memcpy(kx_out, kx, sizeof(int64_t) * K);

  float* Hs = (float*)calloc((size_t)4 * M * 36, 4);
  float* vs = (float*)calloc((size_t)2 * M * 6, 4);
  float* Eiz = (float*)calloc((size_t)M * 6 * HW, 4);
  float* Ejz = E + (long)P * 6 * HW; /* E = cat(Ei, Ejz) (:1758) */
  float* Cii = (float*)calloc((size_t)M * HW, 4);
  float* wi = (float*)calloc((size_t)M * HW, 4);
  orc_projective_transform(targets, weights, poses, disps, intr, extr, ii, jj, M, ht, wd, Hs, vs, Eiz, Ejz, Cii, wi);

  const int n6 = 6 * P;
  double* A = (double*)calloc((size_t)n6 * n6 + 1, 8);
  double* a = (double*)calloc((size_t)n6 + 1, 8);
  /* A.update_lhs(Hs, cat(ii,ii,jj,jj)-kf0, cat(ii,jj,ii,jj)-kf0)  (:1742-1744, 1254-1282) */
  for (int blk = 0; blk < 4; blk++)
    for (int e = 0; e < M; e++) {
      long i = ((blk < 2) ? ii[e] : jj[e]) - kf0;
      long j = ((blk % 2 == 0) ? ii[e] : jj[e]) - kf0;
      if (i >= 0 && j >= 0 && i < P && j < P)
        for (int k = 0; k < 6; k++)
          for (int l = 0; l < 6; l++)
            A[(6 * i + k) * n6 + (6 * j + l)] += (double)Hs[(((long)blk * M + e) * 6 + k) * 6 + l];
    }
  /* A.update_rhs(vs, cat(ii,jj)-kf0)  (:1746-1747, 1284-1299) */
  for (int blk = 0; blk < 2; blk++)
    for (int e = 0; e < M; e++) {
      long i = ((blk == 0) ? ii[e] : jj[e]) - kf0;
      if (i >= 0 && i < P)
        for (int j = 0; j < 6; j++) a[i * 6 + j] += (double)vs[((long)blk * M + e) * 6 + j];
    }

  /* depth block (:1750-1754) */
  float* Cacc = (float*)malloc(sizeof(float) * (size_t)K * HW);
  accum(Cii, ii, M, kx, K, HW, Cacc);
  accum(wi, ii, M, kx, K, HW, w);
  const float alpha = 0.05f;
  for (int k = 0; k < K; k++)
    for (int p = 0; p < HW; p++) {
      float ds = disps_sens[kx[k] * HW + p];
      float m = (ds > 0) ? 1.0f : 0.0f;
      float C = Cacc[(long)k * HW + p] + m * alpha + (1 - m) * eta[(long)k * HW + p];
      w[(long)k * HW + p] = w[(long)k * HW + p] - m * alpha * (disps[kx[k] * HW + p] - ds);
      Q[(long)k * HW + p] = (float)(1.0 / C);
    }
  /* Ei = accum(Eiz, ii, ts) (:1757) */
  accum(Eiz, ii, M, ts, P, (long)6 * HW, E);

  /* schur_block (:1349-1438).  Three passes so that the pixel sums (the only expensive part: one per ordered row pair of a
   * depth slot, ~68 k of them at P = 256 / M = 3912) can run on all host cores while the accumulation into S keeps the
   * serial enumeration order of the reference's pair list (:1368-1402) -- the result is bit-identical to the plain double
   * loop this replaces, for any thread count. */
  double* S = (double*)calloc((size_t)n6 * n6 + 1, 8);
  double* s = (double*)calloc((size_t)n6 + 1, 8);
  long n_pairs = 0;
  for (int n = 0; n < NE; n++) {
    if (!(jj_exp[n] >= kf0 && jj_exp[n] < kf1)) continue; /* (:1375; j==kf1 would overflow graph[P]) */
    for (int m = 0; m < NE; m++)
      if (jj_exp[m] >= kf0 && jj_exp[m] < kf1 && kk[n] == kk[m]) n_pairs++;
  }
  int* pn = (int*)malloc(sizeof(int) * (size_t)(n_pairs + 1));
  int* pm = (int*)malloc(sizeof(int) * (size_t)(n_pairs + 1));
  float* dSf = (float*)malloc(sizeof(float) * (size_t)(n_pairs + 1) * 36);
  float* bbf = (float*)calloc((size_t)NE * 6 + 1, 4);
  n_pairs = 0;
  for (int n = 0; n < NE; n++) {
    if (!(jj_exp[n] >= kf0 && jj_exp[n] < kf1)) continue;
    for (int m = 0; m < NE; m++)
      if (jj_exp[m] >= kf0 && jj_exp[m] < kf1 && kk[n] == kk[m]) { pn[n_pairs] = n; pm[n_pairs] = m; n_pairs++; }
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (long t = 0; t < n_pairs; t++) {
    const int n = pn[t], m = pm[t];
    /* EEt6x6: dS[a][b] = sum_px (E[n][a]*q) * E[m][b]  (:1142-1157), float products, reduced here in double */
    double dS[36];
    for (int q = 0; q < 36; q++) dS[q] = 0;
    const float* Qk = Q + kk[n] * HW;
    for (int p = 0; p < HW; p++) {
      float ei[6], ej[6];
      for (int c = 0; c < 6; c++) {
        ei[c] = E[((long)n * 6 + c) * HW + p] * Qk[p];
        ej[c] = E[((long)m * 6 + c) * HW + p];
      }
      for (int c = 0; c < 6; c++)
        for (int d = 0; d < 6; d++) dS[c * 6 + d] += (double)(ei[c] * ej[d]);
    }
    for (int q = 0; q < 36; q++) dSf[t * 36 + q] = (float)dS[q];
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (int n = 0; n < NE; n++) {
    if (!(jj_exp[n] >= kf0 && jj_exp[n] < kf1)) continue;
    /* Ev6x1 (:1191-1197) */
    double bb[6] = {0, 0, 0, 0, 0, 0};
    const float* Qk = Q + kk[n] * HW;
    const float* wk = w + kk[n] * HW;
    for (int p = 0; p < HW; p++) {
      float q_w = Qk[p] * wk[p];
      for (int c = 0; c < 6; c++) bb[c] += (double)(q_w * E[((long)n * 6 + c) * HW + p]);
    }
    for (int c = 0; c < 6; c++) bbf[(long)n * 6 + c] = (float)bb[c];
  }
  for (long t = 0; t < n_pairs; t++) {
    const long tn = jj_exp[pn[t]] - kf0, tm = jj_exp[pm[t]] - kf0;
    for (int c = 0; c < 6; c++)
      for (int d = 0; d < 6; d++) S[(6 * tn + c) * n6 + (6 * tm + d)] += (double)dSf[t * 36 + c * 6 + d];
  }
  for (int n = 0; n < NE; n++) { /* update_rhs(v, jj_exp-kf0) (:1435) */
    if (!(jj_exp[n] >= kf0 && jj_exp[n] < kf1)) continue;
    const long tn = jj_exp[n] - kf0;
    for (int c = 0; c < 6; c++) s[tn * 6 + c] += (double)bbf[(long)n * 6 + c];
  }
  free(pn); free(pm); free(dSf); free(bbf);
  /* rcm = A - S; get_dense(): column-major data read row-major (:1305-1316) => transpose */
  for (int r = 0; r < n6; r++)
    for (int c = 0; c < n6; c++) H[(long)r * n6 + c] = (float)(A[(long)c * n6 + r] - S[(long)c * n6 + r]);
  for (int r = 0; r < n6; r++) v[r] = (float)(a[r] - s[r]);

  free(ii_exp); free(jj_exp); free(ts); free(kx); free(kk); free(Hs); free(vs); free(Eiz);
  free(Cii); free(wi); free(A); free(a); free(Cacc); free(S); free(s);
  return K;
}

/* ------------------------------------------------------------------------- */
/* A11  solve_depth_cuda (droid_kernels.cu:1772-1825): EvT6x1 (:1213-1238,    */
/* rows whose pose index <=0 or >=P are skipped), accum, dz = Q*(w - .),      */
/* disp_retr (:1050-1063).  disps is updated in place.                        */
/* ------------------------------------------------------------------------- */
void orc_solve_depth(const float* dx, float* disps, const float* Q, const float* E, const float* w,
                     const int64_t* ii, const int64_t* jj, int M, int ht, int wd, int kf0, int kf1) {
  const int HW = ht * wd;
  const int P = kf1 - kf0;
  const int NE = P + M;
  int64_t* ii_exp = (int64_t*)malloc(sizeof(int64_t) * NE);
  int64_t* jj_exp = (int64_t*)malloc(sizeof(int64_t) * NE);
  for (int t = 0; t < P; t++) { ii_exp[t] = kf0 + t; jj_exp[t] = kf0 + t; }
  for (int e = 0; e < M; e++) { ii_exp[P + e] = ii[e]; jj_exp[P + e] = jj[e]; }
  int64_t* kx = (int64_t*)malloc(sizeof(int64_t) * NE);
  int64_t* kk = (int64_t*)malloc(sizeof(int64_t) * NE);
  const int K = unique_inverse(ii_exp, NE, kx, kk);
  float* dw = (float*)calloc((size_t)NE * HW, 4);
  for (int n = 0; n < NE; n++) {
    long ix = jj_exp[n] - kf0;
    if (ix <= 0 || ix >= P) continue; /* (:1225) x.size(0) == P */
    for (int p = 0; p < HW; p++) {
      float acc = 0;
      for (int c = 0; c < 6; c++) acc += E[((long)n * 6 + c) * HW + p] * dx[ix * 6 + c];
      dw[(long)n * HW + p] = acc;
    }
  }
  float* dws = (float*)malloc(sizeof(float) * (size_t)K * HW);
  accum(dw, ii_exp, NE, kx, K, HW, dws);
  for (int k = 0; k < K; k++)
    for (int p = 0; p < HW; p++) {
      float dz = Q[(long)k * HW + p] * (w[(long)k * HW + p] - dws[(long)k * HW + p]);
      disps[kx[k] * HW + p] = disps[kx[k] * HW + p] + dz;
    }
  free(ii_exp); free(jj_exp); free(kx); free(kk); free(dw); free(dws);
}

/* ------------------------------------------------------------------------- */
/* Dead-but-exported ops restated for API parity tests.                       */
/* projmap_kernel (:539-628): coords [num,ht,wd,3] (only [..,0:2] written),   */
/* valid [num,ht,wd,1].                                                       */
/* ------------------------------------------------------------------------- */
void orc_projmap(const float* poses, const float* disps, const float* intr, const int64_t* ii,
                 const int64_t* jj, int num, int ht, int wd, float* coords, float* valid) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  memset(coords, 0, sizeof(float) * (size_t)num * HW * 3);
  for (int b = 0; b < num; b++) {
    int ix = (int)ii[b], jx = (int)jj[b];
    float tij[3], qij[4];
    relSE3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij, qij);
    for (int k = 0; k < HW; k++) {
      const float u = (float)(k % wd), v = (float)(k / wd);
      float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1, disps[(long)ix * HW + k]}, Xj[4];
      actSE3(tij, qij, Xi, Xj);
      float* c = coords + ((long)b * HW + k) * 3;
      c[0] = u; c[1] = v;
      if (Xj[2] > 0.01) {
        c[0] = fx * (Xj[0] / Xj[2]) + cx;
        c[1] = fy * (Xj[1] / Xj[2]) + cy;
      }
      valid[(long)b * HW + k] = (Xj[2] > 0.25) ? 1.0f : 0.0f;
    }
  }
}

/* Torch-path reprojection: networks/geom/projective_ops.py:98-145 with jacobian=False
 * (iproj :20-39, actp :69-71, proj :41-53, validity :116-118).  coords [num,HW,2], valid [num,HW].   */
void orc_reproject(const float* poses, const float* disps, const float* intr, const int64_t* ii, const int64_t* jj,
                   int num, int ht, int wd, float* coords, float* valid) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  for (int b = 0; b < num; b++) {
    int ix = (int)ii[b], jx = (int)jj[b];
    float tij[3], qij[4];
    if (ix == jx) {
      tij[0] = -0.1f; tij[1] = 0; tij[2] = 0; qij[0] = qij[1] = qij[2] = 0; qij[3] = 1;
    } else {
      relSE3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij, qij);
    }
    for (int k = 0; k < HW; k++) {
      float Xi[4] = {((float)(k % wd) - cx) / fx, ((float)(k / wd) - cy) / fy, 1, disps[(long)ix * HW + k]}, Xj[4];
      actSE3(tij, qij, Xi, Xj);
      const float Z = (Xj[2] < 0.5f * 0.2f) ? 1.0f : Xj[2];
      const float d = 1.0f / Z;
      coords[((long)b * HW + k) * 2 + 0] = fx * (Xj[0] * d) + cx;
      coords[((long)b * HW + k) * 2 + 1] = fy * (Xj[1] * d) + cy;
      valid[(long)b * HW + k] = (Xj[2] > 0.2f) ? 1.0f : 0.0f;
    }
  }
}

/* iproj_kernel (:896-967): points [nm,ht,wd,3] */
void orc_iproj(const float* poses, const float* disps, const float* intr, int nm, int ht, int wd, float* points) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  for (int b = 0; b < nm; b++)
    for (int k = 0; k < HW; k++) {
      const float u = (float)(k % wd), v = (float)(k / wd);
      float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1, disps[(long)b * HW + k]}, Xj[4];
      actSE3(poses + (long)b * 7, poses + (long)b * 7 + 3, Xi, Xj);
      float* p = points + ((long)b * HW + k) * 3;
      p[0] = Xj[0] / Xj[3];
      p[1] = Xj[1] / Xj[3];
      p[2] = Xj[2] / Xj[3];
    }
}

/* depth_filter_kernel (:773-892): counter [num,ht,wd] */
void orc_depth_filter(const float* poses, const float* disps, const float* intr, const int64_t* inds,
                      const float* thresh, int num, int nframes, int ht, int wd, float* counter) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  memset(counter, 0, sizeof(float) * (size_t)num * HW);
  for (int b = 0; b < num; b++)
    for (int nb = 0; nb < 6; nb++) {
      int ix = (int)inds[b];
      int jx = (nb < 3) ? ix - nb - 1 : ix + nb;
      if (jx < 0 || jx >= nframes) continue;
      const float t = thresh[b];
      float tij[3], qij[4];
      relSE3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij, qij);
      for (int k = 0; k < HW; k++) {
        const int i = k / wd, j = k % wd;
        const float di = disps[(long)ix * HW + k];
        float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1, di}, Xj[4];
        actSE3(tij, qij, Xi, Xj);
        const float uj = fx * (Xj[0] / Xj[2]) + cx;
        const float vj = fy * (Xj[1] / Xj[2]) + cy;
        const float dj = Xj[3] / Xj[2];
        const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
        if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
          const float d00 = disps[(long)jx * HW + (v0 + 0) * wd + u0 + 0];
          const float d01 = disps[(long)jx * HW + (v0 + 0) * wd + u0 + 1];
          const float d10 = disps[(long)jx * HW + (v0 + 1) * wd + u0 + 0];
          const float d11 = disps[(long)jx * HW + (v0 + 1) * wd + u0 + 1];
          if (fabs(1.0 / dj - 1.0 / d00) < t) counter[(long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d01) < t) counter[(long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d10) < t) counter[(long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d11) < t) counter[(long)b * HW + k] += 1.0f;
        }
      }
    }
}

/* corr_index_backward_kernel<float> (correlation_kernels.cu:73-124) */
void orc_corr_index_backward_f32(const float* coords, const float* corr_grad, float* volume_grad,
                                 int B, int h1, int w1, int h2, int w2, int r) {
  const int rd = 2 * r + 1;
  const long HW1 = (long)h1 * w1;
  memset(volume_grad, 0, sizeof(float) * (size_t)B * HW1 * h2 * w2);
  for (int n = 0; n < B; n++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        float x0 = coords[((long)n * 2 + 0) * HW1 + (long)y * w1 + x];
        float y0 = coords[((long)n * 2 + 1) * HW1 + (long)y * w1 + x];
        float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
        const float* cg = corr_grad + (long)n * rd * rd * HW1 + (long)y * w1 + x;
        float* vg = volume_grad + (((long)n * h1 + y) * w1 + x) * (long)h2 * w2;
#define CG(i, j) cg[((long)(i) * rd + (j)) * HW1]
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            int x1 = (int)floorf(x0) - r + i;
            int y1 = (int)floorf(y0) - r + j;
            if (within(y1, x1, h2, w2)) {
              float g = 0.0f;
              if (i > 0 && j > 0) g += CG(i - 1, j - 1) * (dx * dy);
              if (i > 0 && j < rd) g += CG(i - 1, j) * (dx * (1.0f - dy));
              if (i < rd && j > 0) g += CG(i, j - 1) * ((1.0f - dx) * dy);
              if (i < rd && j < rd) g += CG(i, j) * ((1.0f - dx) * (1.0f - dy));
              vg[(long)y1 * w2 + x1] += g;
            }
          }
#undef CG
      }
}
