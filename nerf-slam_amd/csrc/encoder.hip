// encoder.hip -- the kernels around conv.hip that put the tracker's feature / context encoders on the MFMA convolution
// (networks/modules/extractor.py:118-198 `BasicEncoder`: 7x7/2 stem, six residual blocks of 3x3
// convolutions with instance norm (feature net) or no norm (context net), two of them stride 2, 1x1 output layer).
//
// torch + MIOpen run one encoder call as ~90 launches (convolution, three kernels per InstanceNorm2d, relu, add, layout
// transposes): 1.2 ms per 640x480 frame although the arithmetic is 16 GFLOP and every activation fits in the L2 / MALL.
// Here an encoder call is 18 MFMA convolutions (csrc/conv.hip, bias and -- for the context net -- relu in the epilogue) and
//   * enc_stem_im2col:   u8 / f32 image [N,3,H,W] -> normalised f16 patches [N,Ho,Wo,160] (147 = 7x7x3 taps, tap-major),
//                        so that the stem is a 1x1 convolution over 10 chunks of 16 channels;
//   * enc_im2col_3x3s2:  channels-last activation -> [N,Ho,Wo,9C] patches of a stride-2 3x3 convolution (a 1x1 convolution
//                        over 9C channels); the block's 1x1 stride-2 shortcut reads the CENTRE TAP slice of the same buffer;
//   * enc_in_stats:      per (image, channel) sums / sums of squares of a raw convolution output, P partials per image;
//   * enc_in_apply:      finishes the statistics (every workgroup reduces the P partials itself: <= 64 KB out of L2) and
//                        writes relu(IN(y)), or the whole block tail relu(x + relu(IN(y))) / relu(IN(d) + relu(IN(y))).
// All activations are f16 channels-last, statistics f32 partials combined in f64, eps and the biased variance are
// InstanceNorm2d's defaults (extractor.py:133-144 builds it without affine parameters).
#include "common.h"

typedef _Float16 en_f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// stem patches.  k = (ky * 7 + kx) * 3 + c  (the order of weight.permute(0, 2, 3, 1).reshape(32, 147)), 147..159 zero.
// One thread per 16-byte piece, 20 pieces per output pixel.
// ---------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void enc_stem_im2col_kernel(const TI* __restrict__ img, _Float16* __restrict__ out, long npix, int H,
                                                              int W, int Ho, int Wo, float m0, float m1, float m2, float s0, float s1,
                                                              float s2) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * 20) return;
  const long pix = i / 20;
  const int piece = (int)(i - pix * 20);
  const long hw = (long)Ho * Wo;
  const long n = pix / hw;
  const int p = (int)(pix - n * hw), oy = p / Wo, ox = p - oy * Wo;
  const TI* im = img + n * 3 * (long)H * W;
  const float mean[3] = {m0, m1, m2}, inv[3] = {s0, s1, s2};
  en_f16x8 v;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int k = piece * 8 + q;
    const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
    const int iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
    const bool ok = k < 147 && iy >= 0 && iy < H && ix >= 0 && ix < W;
    // (x / 255 - mean) / std in f32, as the reference's normalisation in front of the encoders; the zero padding is of
    // the NORMALISED image
    v[q] = ok ? (_Float16)(((float)im[((long)c * H + iy) * W + ix] / 255.0f - mean[c]) * inv[c]) : (_Float16)0.0f;
  }
  *reinterpret_cast<en_f16x8*>(out + pix * 160 + piece * 8) = v;
}

// patches of a 3x3 / stride 2 / pad 1 convolution: out[n, oy, ox, (ky * 3 + kx) * C + c] = x[n, 2 oy + ky - 1, 2 ox + kx - 1, c]
__global__ __launch_bounds__(256) void enc_im2col_3x3s2_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ out, long npix,
                                                               int H, int W, int C8, int Ho, int Wo) {
  const int ppp = 9 * C8;  // pieces per output pixel
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * ppp) return;
  const long pix = i / ppp;
  const int piece = (int)(i - pix * ppp);
  const int tap = piece / C8, c8 = piece - tap * C8, ky = tap / 3, kx = tap - ky * 3;
  const long hw = (long)Ho * Wo;
  const long n = pix / hw;
  const int p = (int)(pix - n * hw), oy = p / Wo, ox = p - oy * Wo;
  const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (iy >= 0 && iy < H && ix >= 0 && ix < W)
    v = *reinterpret_cast<const uint4*>(x + (((n * H + iy) * (long)W + ix) * C8 + c8) * 8);
  *reinterpret_cast<uint4*>(out + (pix * ppp + piece) * 8) = v;
}

// ---------------------------------------------------------------------------------------------
// instance-norm statistics: partial[n][p][0][c] = sum, [n][p][1][c] = sum of squares over the pixels of part p; the LAST
// workgroup of an image to arrive adds the P rows up (fixed order, double) and leaves the totals in row 0, which is all the
// apply kernel reads.  (Round 2 stopped at the partial rows, at most 64 of them: a quarter of the CUs at 240x320, 19 dependent
// load rounds per workgroup -- and every one of the apply kernel's ~600 workgroups added the 64 rows up again.)
// grid (P, N); thread = (pixel group g = tid / C8, piece = tid % C8): 16-byte loads, whole rows coalesced.
// ---------------------------------------------------------------------------------------------
#define ENC_IN_MAXN 4096
// Arrivals per image are counted in a row of N counters that the CALLER owns (zeroed once; the last arrival resets its entry):
// nerfslam/encoder_op.py keeps one row per normalised layer of an encoder instance, so two statistics launches can only meet
// on a counter if the same layer of the same encoder ran concurrently with itself -- which stream order (and the
// serialisation of replays of one graph) rules out.  (Round 4 handed out rows of a global table round-robin at LAUNCH time; a
// captured graph then baked one row into every replay)

__global__ __launch_bounds__(256) void enc_in_stats_kernel(const _Float16* __restrict__ x, float* __restrict__ partial, int HW, int C8,
                                                           int P, unsigned int* __restrict__ ticket) {
  __shared__ float red[4][2][128];
  __shared__ double ps[256], pq[256];
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int p = blockIdx.x, n = blockIdx.y;
  const int C = C8 * 8, G = 256 / C8;
  const int chunk = (HW + P - 1) / P;
  const int lo = p * chunk, hi = min(HW, lo + chunk);
  const int piece = tid % C8, g = tid / C8;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = q[e] = 0.0f;
  const _Float16* xb = x + (long)n * HW * C + piece * 8;
  // four loads in flight per lane (a workgroup is a dozen dependent load rounds otherwise)
  for (int pix = lo + g; pix < hi; pix += 4 * G) {
    en_f16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int px = pix + u * G;
      v[u] = px < hi ? *reinterpret_cast<const en_f16x8*>(xb + (long)px * C) : (en_f16x8)(_Float16)0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float f = (float)v[u][e];
        s[e] += f;
        q[e] += f * f;
      }
  }
  // lanes of one wave that hold the same piece are C8 apart (C8 = 4, 8 or 16 divides 64)
  for (int off = C8; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      s[e] += __shfl_xor(s[e], off, 64);
      q[e] += __shfl_xor(q[e], off, 64);
    }
  }
  if (lane < C8) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      red[wv][0][lane * 8 + e] = s[e];
      red[wv][1][lane * 8 + e] = q[e];
    }
  }
  __syncthreads();
  float* rows = partial + (long)n * P * 2 * C;
  if (tid < 2 * C) {
    const int k = tid / C, c = tid - k * C;
    rows[((long)p * 2 + k) * C + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
  }
  if (P == 1) return;
  // publish the row, take a ticket; the last arrival sees every row (agent-scope release / acquire around the counter)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    s_ticket = atomicAdd(&ticket[n], 1u);
    __threadfence();
  }
  __syncthreads();
  if (s_ticket != (unsigned)(P - 1)) return;
  {
    const int ncol = 2 * C, RG = 256 / ncol;
    const int col = tid % ncol, rg = tid / ncol;
    double ds = 0.0;                            // (row r = [sum C | sum of squares C]; 16 loads in flight, fixed order)
    for (int r = rg; r < P; r += 16 * RG) {
      float f[16];
#pragma unroll
      for (int u = 0; u < 16; u++) f[u] = r + u * RG < P ? rows[(long)(r + u * RG) * ncol + col] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; u++) ds += (double)f[u];
    }
    ps[tid] = ds;
    __syncthreads();
    if (tid < ncol) {
      for (int gq = 1; gq < RG; gq++) ds += ps[tid + gq * ncol];
      pq[tid] = ds;
    }
    __syncthreads();
    if (tid < ncol) rows[tid] = (float)pq[tid];     // totals -> row 0 (every row has been read)
    if (tid == 0) ticket[n] = 0u;
  }
}

struct InApplyArgs {
  const _Float16* y;      // [N,HW,C] raw convolution output (or, without ystats, an activation that is used as it is)
  const float* ystats;    // partials of y, or nullptr: no normalisation of y
  const _Float16* x;      // residual input [N,HW,C], or nullptr
  const float* xstats;    // partials of x (the shortcut convolution's raw output: normalised, no relu), or nullptr
  _Float16* out;
  int HW, C8, P;
  float eps;
};

// out = relu(x' + relu(y')), y' = IN(y) or y, x' = IN(x) or x or nothing.  grid (ceil(HW C8 / 1024), N).
__global__ __launch_bounds__(256) void enc_in_apply_kernel(InApplyArgs a) {
  __shared__ float mu[2][128], rs[2][128];
  const int tid = threadIdx.x, n = blockIdx.y;
  const int C = a.C8 * 8;
  // mean / 1 / sigma of the 2C columns (y | x, channel) from the totals the statistics kernel left in row 0 of the image
  if (tid < 2 * C) {
    const int k = tid / C, c = tid - k * C;
    const float* st = k == 0 ? a.ystats : a.xstats;
    float m = 0.0f, r = 1.0f;
    if (st) {
      const float* b = st + (long)n * a.P * 2 * C;
      const double s = (double)b[c], q = (double)b[C + c];
      const double mean = s / a.HW, var = fmax(q / a.HW - mean * mean, 0.0);   // biased variance (InstanceNorm2d)
      m = (float)mean;
      r = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    mu[k][c] = m;
    rs[k][c] = r;
  }
  __syncthreads();
  const long total = (long)a.HW * a.C8;
  const long base = (long)n * total;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const long i = ((long)blockIdx.x * 4 + it) * 256 + tid;
    if (i >= total) break;
    const int c0 = (int)(i % a.C8) * 8;
    en_f16x8 v = *reinterpret_cast<const en_f16x8*>(a.y + (base + i) * 8);
    en_f16x8 xv;
    if (a.x) xv = *reinterpret_cast<const en_f16x8*>(a.x + (base + i) * 8);
    en_f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float f = fmaxf(((float)v[e] - mu[0][c0 + e]) * rs[0][c0 + e], 0.0f);
      if (a.x) f = fmaxf(f + ((float)xv[e] - mu[1][c0 + e]) * rs[1][c0 + e], 0.0f);
      o[e] = (_Float16)f;
    }
    *reinterpret_cast<en_f16x8*>(a.out + (base + i) * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int ns_enc_stem_im2col(const void* img, int img_is_u8, void* out, int N, int H, int W, const float* mean, const float* std,
                                  void* stream) {
  if (N == 0) return NS_OK;
  NS_REQUIRE(img && out && mean && std, "ns_enc_stem_im2col: null pointer");
  NS_REQUIRE(N > 0 && H > 0 && W > 0, "ns_enc_stem_im2col: bad shape");
  NS_REQUIRE(std[0] != 0.0f && std[1] != 0.0f && std[2] != 0.0f, "ns_enc_stem_im2col: zero std");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long npix = (long)N * Ho * Wo;
  const dim3 grid(ns_cdiv(npix * 20, 256));
  if (img_is_u8)
    hipLaunchKernelGGL(enc_stem_im2col_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img,
                       (_Float16*)out, npix, H, W, Ho, Wo, mean[0], mean[1], mean[2], 1.0f / std[0], 1.0f / std[1], 1.0f / std[2]);
  else
    hipLaunchKernelGGL(enc_stem_im2col_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)img, (_Float16*)out, npix,
                       H, W, Ho, Wo, mean[0], mean[1], mean[2], 1.0f / std[0], 1.0f / std[1], 1.0f / std[2]);
  NS_CHECK_LAUNCH("enc_stem_im2col_kernel");
  return NS_OK;
}

extern "C" int ns_enc_im2col_3x3s2(const void* x, void* out, int N, int H, int W, int C, void* stream) {
  if (N == 0) return NS_OK;
  NS_REQUIRE(x && out, "ns_enc_im2col_3x3s2: null pointer");
  NS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ns_enc_im2col_3x3s2: bad shape (C = %d must be a multiple of 8)", C);
  NS_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0, "ns_enc_im2col_3x3s2: 16-byte alignment");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long npix = (long)N * Ho * Wo;
  hipLaunchKernelGGL(enc_im2col_3x3s2_kernel, dim3(ns_cdiv(npix * 9 * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x, (_Float16*)out, npix, H, W, C / 8, Ho, Wo);
  NS_CHECK_LAUNCH("enc_im2col_3x3s2_kernel");
  return NS_OK;
}

extern "C" int ns_enc_in_parts(int HW) {
  const int p = (HW + 511) / 512;
  return p < 1 ? 1 : (p > 64 ? 64 : p);
}

extern "C" int ns_enc_in_stats(const void* x, float* partial, unsigned int* ticket, int N, int HW, int C, void* stream) {
  if (N == 0) return NS_OK;
  NS_REQUIRE(x && partial, "ns_enc_in_stats: null pointer");
  NS_REQUIRE(N > 0 && N <= ENC_IN_MAXN && HW > 0 && (C == 32 || C == 64 || C == 128),
             "ns_enc_in_stats: bad shape (N %d <= 4096, HW %d, C %d: 32, 64 or 128 channels)", N, HW, C);
  NS_REQUIRE(((uintptr_t)x % 16) == 0, "ns_enc_in_stats: 16-byte alignment");
  const int P = ns_enc_in_parts(HW);
  NS_REQUIRE(P == 1 || ticket, "ns_enc_in_stats: %d partial rows per image need the caller's arrival counters (N zeroed uint32)", P);
  hipLaunchKernelGGL(enc_in_stats_kernel, dim3(P, N), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, partial, HW, C / 8, P,
                     ticket);
  NS_CHECK_LAUNCH("enc_in_stats_kernel");
  return NS_OK;
}

extern "C" int ns_enc_in_apply(const void* y, const float* ystats, const void* x, const float* xstats, void* out, int N, int HW, int C,
                               float eps, void* stream) {
  if (N == 0) return NS_OK;
  NS_REQUIRE(y && out, "ns_enc_in_apply: null pointer");
  NS_REQUIRE(N > 0 && N <= 65535 && HW > 0 && (C == 32 || C == 64 || C == 128),
             "ns_enc_in_apply: bad shape (N %d, HW %d, C %d: 32, 64 or 128 channels)", N, HW, C);
  NS_REQUIRE(!(xstats && !x), "ns_enc_in_apply: statistics of a residual that is not there");
  NS_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)x % 16) == 0, "ns_enc_in_apply: 16-byte alignment");
  InApplyArgs a;
  a.y = (const _Float16*)y;
  a.ystats = ystats;
  a.x = (const _Float16*)x;
  a.xstats = xstats;
  a.out = (_Float16*)out;
  a.HW = HW;
  a.C8 = C / 8;
  a.P = ns_enc_in_parts(HW);
  a.eps = eps;
  hipLaunchKernelGGL(enc_in_apply_kernel, dim3(ns_cdiv((long)HW * a.C8, 1024), N), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("enc_in_apply_kernel");
  return NS_OK;
}
