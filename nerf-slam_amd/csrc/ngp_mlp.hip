// ngp_mlp.hip -- the two tiny MLPs of the NeRF (density 32->64->16, colour 32->64->64->16) for gfx950.
//
// Boundary: what instant-ngp's "FullyFusedMLP" does inside `Testbed.frame()` (reference call site
// /root/reference/fusion/nerf_fusion.py:299).  Parity unpinned (see ngp.hip).
//
// Round-1 structure:
//   forward / activation backward: one lane per sample, weights broadcast out of LDS, f16 storage with
//       f32 accumulation, activations written UNIT-MAJOR ([unit][sample]) so every store of a wave is one
//       contiguous 128-byte line and the weight-gradient GEMM can read 8 consecutive samples per lane;
//   weight gradients: dW = dY^T X is the only place where a reduction over the 2^18 samples happens
//       and it IS GEMM-shaped (M,N <= 64, K = samples) -> v_mfma_f32_32x32x16_f16, split-K over
//       workgroups, rows staged through LDS with coalesced loads, deterministic two-stage reduction.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// weights (f16, row-major [out][in]) packed back to back: W1[64,32] W2[16,64] W3[64,32] W4[64,64] W5[16,64]
#define W1_OFF 0
#define W2_OFF 2048
#define W3_OFF 3072
#define W4_OFF 5120
#define W5_OFF 9216
#define W_TOTAL 10240

__device__ __forceinline__ void sh16(float x, float y, float z, float* o) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// y[o] = f16( sum_k W[o][k] x[k] ), optional ReLU; W in LDS (wave-uniform address -> broadcast).
template <int NIN, int NOUT, bool RELU>
__device__ __forceinline__ void dense(const _Float16* __restrict__ W, const _Float16* x, _Float16* y) {
#pragma unroll 4
  for (int o = 0; o < NOUT; o++) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NIN; k += 8) {
      const f16x8 w = *reinterpret_cast<const f16x8*>(W + o * NIN + k);
#pragma unroll
      for (int q = 0; q < 8; q++) acc = fmaf((float)w[q], (float)x[k + q], acc);
    }
    if (RELU) acc = fmaxf(acc, 0.0f);
    y[o] = (_Float16)acc;
  }
}

// dx[k] = f16( sum_o W[o][k] dy[o] ) with the transposed copy WT[k][o] in LDS
template <int NIN, int NOUT>
__device__ __forceinline__ void dense_t(const _Float16* __restrict__ WT, const _Float16* dy, _Float16* dx) {
#pragma unroll 4
  for (int k = 0; k < NIN; k++) {
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < NOUT; o += 8) {
      const f16x8 w = *reinterpret_cast<const f16x8*>(WT + k * NOUT + o);
#pragma unroll
      for (int q = 0; q < 8; q++) acc = fmaf((float)w[q], (float)dy[o + q], acc);
    }
    dx[k] = (_Float16)acc;
  }
}

struct MlpFwdArgs {
  const _Float16* W;      // packed weights
  const _Float16* feat;   // [N,32]
  const float* dirs;      // [N,3]
  _Float16* out;          // [N,4] (r,g,b raw, log-density)
  // unit-major activations for the backward pass (all null in inference)
  _Float16* featT;        // [32,N]
  _Float16* h1T;          // [64,N]
  _Float16* cinT;         // [32,N]
  _Float16* h3T;          // [64,N]
  _Float16* h4T;          // [64,N]
  long N;
};

__global__ __launch_bounds__(256) void ngp_mlp_fwd_kernel(MlpFwdArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ws[W_TOTAL];
  for (int i = threadIdx.x; i < W_TOTAL / 8; i += 256)
    reinterpret_cast<f16x8*>(Ws)[i] = reinterpret_cast<const f16x8*>(a.W)[i];
  __syncthreads();
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  const long N = a.N;
  const bool save = a.h1T != nullptr;
  _Float16 x[32], h[64], g[64];
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(a.feat + n * 32 + k);
#pragma unroll
    for (int q = 0; q < 8; q++) x[k + q] = v[q];
  }
  if (save)
#pragma unroll
    for (int k = 0; k < 32; k++) a.featT[(long)k * N + n] = x[k];
  dense<32, 64, true>(Ws + W1_OFF, x, h);
  if (save)
#pragma unroll
    for (int k = 0; k < 64; k++) a.h1T[(long)k * N + n] = h[k];
  _Float16 cin[32];
  dense<64, 16, false>(Ws + W2_OFF, h, cin);
  const _Float16 logdens = cin[0];
  float sh[16];
  sh16(a.dirs[n * 3], a.dirs[n * 3 + 1], a.dirs[n * 3 + 2], sh);
#pragma unroll
  for (int k = 0; k < 16; k++) cin[16 + k] = (_Float16)sh[k];
  if (save)
#pragma unroll
    for (int k = 0; k < 32; k++) a.cinT[(long)k * N + n] = cin[k];
  dense<32, 64, true>(Ws + W3_OFF, cin, h);
  if (save)
#pragma unroll
    for (int k = 0; k < 64; k++) a.h3T[(long)k * N + n] = h[k];
  dense<64, 64, true>(Ws + W4_OFF, h, g);
  if (save)
#pragma unroll
    for (int k = 0; k < 64; k++) a.h4T[(long)k * N + n] = g[k];
  _Float16 rgb[16];
  dense<64, 16, false>(Ws + W5_OFF, g, rgb);
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 o = {rgb[0], rgb[1], rgb[2], logdens};
  *reinterpret_cast<f16x4*>(a.out + n * 4) = o;
}

struct MlpBwdArgs {
  const _Float16* WT;     // packed TRANSPOSED weights: W1T[32,64] W2T[64,16] W3T[32,64] W4T[64,64] W5T[64,16]
  const _Float16* dLdout; // [N,4]
  const _Float16 *h1T, *h3T, *h4T;  // saved activations (ReLU masks)
  _Float16* dLdfeat;      // [N,32]
  _Float16 *d5T, *d4T, *d3T, *ddT, *d1T;  // [16,N] [64,N] [64,N] [16,N] [64,N] unit-major output gradients
  long N;
};

__global__ __launch_bounds__(256) void ngp_mlp_bwd_kernel(MlpBwdArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ws[W_TOTAL];
  for (int i = threadIdx.x; i < W_TOTAL / 8; i += 256)
    reinterpret_cast<f16x8*>(Ws)[i] = reinterpret_cast<const f16x8*>(a.WT)[i];
  __syncthreads();
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  const long N = a.N;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 go = *reinterpret_cast<const f16x4*>(a.dLdout + n * 4);
  _Float16 dy[64], dx[64];
  // layer 5 (no activation): dY5 = (dr, dg, db, 0, ...)
#pragma unroll
  for (int k = 0; k < 16; k++) dy[k] = (k < 3) ? go[k] : (_Float16)0;
#pragma unroll
  for (int k = 0; k < 16; k++) a.d5T[(long)k * N + n] = dy[k];
  dense_t<64, 16>(Ws + W5_OFF, dy, dx);  // W5T [64][16]
#pragma unroll
  for (int k = 0; k < 64; k++) {  // ReLU' of layer 4
    dy[k] = ((float)a.h4T[(long)k * N + n] > 0.0f) ? dx[k] : (_Float16)0;
    a.d4T[(long)k * N + n] = dy[k];
  }
  dense_t<64, 64>(Ws + W4_OFF, dy, dx);  // W4T [64][64]
#pragma unroll
  for (int k = 0; k < 64; k++) {  // ReLU' of layer 3
    dy[k] = ((float)a.h3T[(long)k * N + n] > 0.0f) ? dx[k] : (_Float16)0;
    a.d3T[(long)k * N + n] = dy[k];
  }
  dense_t<32, 64>(Ws + W3_OFF, dy, dx);  // W3T [32][64] -> d(cin); only the density half flows further
#pragma unroll
  for (int k = 0; k < 16; k++) {
    float v = (float)dx[k];
    if (k == 0) v += (float)go[3];
    dy[k] = (_Float16)v;
    a.ddT[(long)k * N + n] = dy[k];
  }
  dense_t<64, 16>(Ws + W2_OFF, dy, dx);  // W2T [64][16]
#pragma unroll
  for (int k = 0; k < 64; k++) {  // ReLU' of layer 1
    dy[k] = ((float)a.h1T[(long)k * N + n] > 0.0f) ? dx[k] : (_Float16)0;
    a.d1T[(long)k * N + n] = dy[k];
  }
  dense_t<32, 64>(Ws + W1_OFF, dy, dx);  // W1T [32][64]
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    f16x8 o;
#pragma unroll
    for (int q = 0; q < 8; q++) o[q] = dx[k + q];
    *reinterpret_cast<f16x8*>(a.dLdfeat + n * 32 + k) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradients: dW[o][i] = sum_n dYT[o][n] * XT[i][n]     (MFMA, split-K over workgroups)
// grid (ksplit, 5 layers).  Per step a workgroup stages 64 samples of every row of dYT and XT
// (<= 128 rows x 128 B) through LDS with fully coalesced loads; each of the 4 waves owns one 32x32
// tile of the (<= 64 x 64) output.  partial[layer][ksplit][out*in] f32, summed by the second kernel
// in split order (deterministic).
// ---------------------------------------------------------------------------------------------
struct WgradLayer {
  const _Float16* dYT;  // [nout][N]
  const _Float16* XT;   // [nin][N]
  int nout, nin, woff;
};
struct WgradArgs {
  WgradLayer layer[5];
  float* partial;  // [ksplit][W_TOTAL]
  long N;
  int ksplit;
};

#define WG_ROWB 144  // 64 samples * 2 B + 16 B pad: conflict-free 16-byte slots

__global__ __launch_bounds__(256) void ngp_mlp_wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) char tile[128 * WG_ROWB];
  const WgradLayer L = a.layer[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int nrows = L.nout + L.nin;
  // wave -> output tile (to, ti); layers with 16 outputs use one row tile
  const int tiles_o = (L.nout + 31) / 32, tiles_i = (L.nin + 31) / 32;
  const int to = wave / tiles_i, ti = wave % tiles_i;
  const bool active = wave < tiles_o * tiles_i;
  f32x16 acc = (f32x16)0.0f;
  const long per = ((a.N + a.ksplit - 1) / a.ksplit + 63) / 64 * 64;
  const long n0 = (long)blockIdx.x * per, n1 = min(a.N, n0 + per);
  for (long nb = n0; nb < n1; nb += 64) {
    __syncthreads();
    // stage: row r (0..nout-1: dYT, then XT) x 8 pieces of 16 B
    for (int piece = tid; piece < nrows * 8; piece += 256) {
      const int r = piece >> 3, s = piece & 7;
      const _Float16* src = (r < L.nout ? L.dYT + (long)r * a.N : L.XT + (long)(r - L.nout) * a.N) + nb + s * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (nb + s * 8 + 8 <= a.N) {
        v = *reinterpret_cast<const uint4*>(src);
      } else {
        _Float16 t[8];
        for (int q = 0; q < 8; q++) t[q] = (nb + s * 8 + q < a.N) ? src[q] : (_Float16)0;
        v = *reinterpret_cast<const uint4*>(t);
      }
      *reinterpret_cast<uint4*>(tile + r * WG_ROWB + s * 16) = v;
    }
    __syncthreads();
    if (active) {
      const int ro = to * 32 + col, ri = L.nout + ti * 32 + col;
      const bool oko = ro < L.nout, oki = (ti * 32 + col) < L.nin;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {  // 4 x 16 samples
        f16x8 av = *reinterpret_cast<const f16x8*>(tile + (oko ? ro : 0) * WG_ROWB + ks * 32 + half * 16);
        f16x8 bv = *reinterpret_cast<const f16x8*>(tile + (oki ? ri : 0) * WG_ROWB + ks * 32 + half * 16);
        if (!oko) av = (f16x8)(_Float16)0;
        if (!oki) bv = (f16x8)(_Float16)0;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
      }
    }
  }
  if (active) {
    // D[i = out row][j = in col]: lane holds col j = lane&31, rows (r&3) + 8*(r>>2) + 4*half
    float* P = a.partial + (long)blockIdx.x * W_TOTAL + L.woff;
    const int j = ti * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int i = to * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < L.nout && j < L.nin) P[i * L.nin + j] = acc[r];
    }
  }
}

__global__ __launch_bounds__(256) void ngp_mlp_wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                                   float* __restrict__ grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W_TOTAL) return;
  float s = 0.0f;
  for (int k = 0; k < ksplit; k++) s += partial[(long)k * W_TOTAL + i];
  grad[i] += s;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int ns_ngp_mlp_forward(const void* weights, const void* feat, const float* dirs, void* out, void* featT,
                                  void* h1T, void* cinT, void* h3T, void* h4T, long N, void* stream) {
  NS_REQUIRE(weights && feat && dirs && out, "ns_ngp_mlp_forward: null pointer");
  NS_REQUIRE((h1T == nullptr) == (featT == nullptr) && (h1T == nullptr) == (cinT == nullptr) &&
                 (h1T == nullptr) == (h3T == nullptr) && (h1T == nullptr) == (h4T == nullptr),
             "ns_ngp_mlp_forward: pass all activation buffers (training) or none (inference)");
  if (N <= 0) return NS_OK;
  MlpFwdArgs a{(const _Float16*)weights, (const _Float16*)feat, dirs, (_Float16*)out, (_Float16*)featT,
               (_Float16*)h1T, (_Float16*)cinT, (_Float16*)h3T, (_Float16*)h4T, N};
  hipLaunchKernelGGL(ngp_mlp_fwd_kernel, dim3(ns_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_fwd_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_mlp_backward(const void* weightsT, const void* dLdout, const void* featT, const void* h1T,
                                   const void* cinT, const void* h3T, const void* h4T, void* dLdfeat, void* d5T,
                                   void* d4T, void* d3T, void* ddT, void* d1T, float* partial_ws, int ksplit,
                                   float* grad_weights, long N, void* stream) {
  NS_REQUIRE(weightsT && dLdout && featT && h1T && cinT && h3T && h4T && dLdfeat && d5T && d4T && d3T && ddT && d1T &&
                 partial_ws && grad_weights,
             "ns_ngp_mlp_backward: null pointer");
  NS_REQUIRE(ksplit >= 1 && N % 8 == 0, "ns_ngp_mlp_backward: ksplit >= 1 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  hipStream_t st = (hipStream_t)stream;
  MlpBwdArgs b{(const _Float16*)weightsT, (const _Float16*)dLdout, (const _Float16*)h1T, (const _Float16*)h3T,
               (const _Float16*)h4T,      (_Float16*)dLdfeat,      (_Float16*)d5T,       (_Float16*)d4T,
               (_Float16*)d3T,            (_Float16*)ddT,          (_Float16*)d1T,       N};
  hipLaunchKernelGGL(ngp_mlp_bwd_kernel, dim3(ns_cdiv(N, 256)), dim3(256), 0, st, b);
  NS_CHECK_LAUNCH("ngp_mlp_bwd_kernel");
  WgradArgs w;
  w.layer[0] = WgradLayer{(const _Float16*)d1T, (const _Float16*)featT, 64, 32, W1_OFF};
  w.layer[1] = WgradLayer{(const _Float16*)ddT, (const _Float16*)h1T, 16, 64, W2_OFF};
  w.layer[2] = WgradLayer{(const _Float16*)d3T, (const _Float16*)cinT, 64, 32, W3_OFF};
  w.layer[3] = WgradLayer{(const _Float16*)d4T, (const _Float16*)h3T, 64, 64, W4_OFF};
  w.layer[4] = WgradLayer{(const _Float16*)d5T, (const _Float16*)h4T, 16, 64, W5_OFF};
  w.partial = partial_ws;
  w.N = N;
  w.ksplit = ksplit;
  hipLaunchKernelGGL(ngp_mlp_wgrad_kernel, dim3(ksplit, 5), dim3(256), 0, st, w);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_kernel");
  hipLaunchKernelGGL(ngp_mlp_wgrad_reduce_kernel, dim3(ns_cdiv(W_TOTAL, 256)), dim3(256), 0, st, partial_ws, ksplit,
                     grad_weights);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_reduce_kernel");
  return NS_OK;
}
