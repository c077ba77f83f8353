// ba_solve_large.hip -- the GTSAM round trip of RaftVisualFrontend.ba() (slam/visual_frontends/visual_frontend.py:
// 1123-1158) for systems that do not fit one workgroup's LDS: the global BA over the whole keyframe buffer
// (backend(), :1255-1295; 6P = 1536 for config #5's 256 keyframes) and frontend windows with covariances above 18
// poses.  Same semantics as ns_ba_solve (ba_solve.hip), same bordered formulation, blocked through HBM/L2:
//
//   W = [ A (lower triangle, n x n) ; b^T (1 x n) ; I (n x n, only when L^-1 is wanted) ]        f64, row-major, ld = n
//
// Right-looking blocked Cholesky, block NB = 32, two launches per block column:
//   panel    : every workgroup factors the 32 x 32 diagonal block in LDS (redundantly: cheaper than a launch and a
//              dependency), then solves X L_kk^T = W[rows, k-block] for its 256 rows, one row per lane in registers.
//              The UNFACTORED block in W is read by every workgroup of the launch and never written: workgroup 0 leaves
//              the factor in a separate array of diagonal blocks (Dfac, behind W in the workspace), which the back
//              substitution reads.  (Writing it back in place would race with late-dispatched workgroups that still
//              have to read the input block -- nothing orders the workgroups of one launch.)
//   trailing : W[i, j] -= P_i . P_j for the rows below and the columns right of the block, 64 x 64 tiles, 4 x 4 per lane,
//              operands staged through LDS.  Plain v_fma_f64: on CDNA4 the vector and matrix f64 peaks are the same
//              (78.6 TFLOP/s), so there is nothing for MFMA to win here, and n^3/3 = 1.2 GFLOP at 6P = 1536.
// The border row leaves y = L^-1 b (no separate forward substitution), the identity rows leave (L^-1)^T.  One
// workgroup does the back substitution L^T x = y (n^2/2 FMAs, L streamed once), then the retraction kernel of
// ba_solve.hip runs under the device-side info flag.  Nothing returns to the host.
#include "common.h"
#include "se3.h"

#define LNB 32
#define LPAD 33

struct LargeArgs {
  double* W;
  int n, nrows;      // system size; rows of W (n + 1 or 2n + 1)
  int32_t* info;     // device flag: first non-positive pivot + 1
  double* Dfac;      // [ceil(n / 32)][32 x 32] factored diagonal blocks (lower triangle, row-major, ld = 32)
};

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bsl_load_kernel(const float* __restrict__ H, const float* __restrict__ v, float ep,
                                                       float lm, LargeArgs a, int want_inv) {
  const int n = a.n;
  const long total = (long)a.nrows * n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / n), c = (int)(idx - (long)r * n);
    double out;
    if (r < n) {
      // the HessianFactors keep the upper triangle of what they are given (:1127-1134): entry (r, c), c <= r, is H[c][r]
      if (c <= r) {
        out = (double)H[(long)c * n + r];
        if (c == r) out += (double)ep + (double)lm * out;
      } else {
        out = 0.0;
      }
    } else if (r == n) {
      out = (double)v[c];
    } else {
      out = (r - n - 1 == c) ? 1.0 : 0.0;
    }
    a.W[idx] = out;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.info = 0;
}

__global__ void bsl_prior_kernel(LargeArgs a, const float* __restrict__ prior, const float* __restrict__ wTb, int kf0,
                                 float prior_sigma) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double pr[7], x0[7], pinv[7], rel[7], e[6];
  for (int k = 0; k < 7; k++) {
    pr[k] = (double)prior[k];
    x0[k] = (double)wTb[(long)kf0 * 7 + k];
  }
  se3::inv(pr, pinv);
  se3::mul(pinv, x0, rel);
  se3::log_wv(rel, e);
  const double info = 1.0 / ((double)prior_sigma * (double)prior_sigma);
  for (int k = 0; k < 6; k++) {
    a.W[(long)k * a.n + k] += info;
    a.W[(long)a.n * a.n + k] += -e[k] * info;
  }
}

__global__ __launch_bounds__(256) void bsl_hfull_kernel(LargeArgs a, double* __restrict__ Hfull) {
  const int n = a.n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)n * n; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / n), c = (int)(idx - (long)r * n);
    Hfull[idx] = (c <= r) ? a.W[idx] : a.W[(long)c * n + r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// panel: block column k0 .. k0+nb.  Workgroup b owns rows k0 + nb + 256 b ... (one per lane); workgroup 0 also writes the
// factored diagonal block to a.Dfac (never into W: see the header).
__global__ __launch_bounds__(256) void bsl_panel_kernel(LargeArgs a, int k0) {
  __shared__ double D[LNB * LPAD];
  __shared__ double rdiag[LNB];
  const int n = a.n, tid = threadIdx.x;
  const int nb = min(LNB, n - k0);
  for (int idx = tid; idx < LNB * LNB; idx += 256) {
    const int r = idx / LNB, c = idx % LNB;
    D[r * LPAD + c] = (r < nb && c <= r) ? a.W[(long)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  // unblocked right-looking factorisation of the 32 x 32 block: column j scaled by lanes i > j, then the rank-1 update
  for (int j = 0; j < nb; j++) {
    const double s = D[j * LPAD + j];
    const bool bad = !(s > 0.0);
    const double sp = bad ? 1.0 : s;
    double di = __builtin_amdgcn_rsq(sp);
    di = di * (1.5 - 0.5 * sp * di * di);
    di = di * (1.5 - 0.5 * sp * di * di);
    __syncthreads();            // every lane has read the pivot before lane 0 overwrites it
    if (tid == 0) {
      D[j * LPAD + j] = sp * di;
      rdiag[j] = di;
      if (bad && blockIdx.x == 0) atomicCAS(a.info, 0, k0 + j + 1);
    }
    if (tid > j && tid < nb) D[tid * LPAD + j] *= di;
    __syncthreads();
    // trailing entries (r, c), j < c <= r < nb
    for (int idx = tid; idx < LNB * LNB; idx += 256) {
      const int r = idx / LNB, c = idx % LNB;
      if (c > j && c <= r && r < nb) D[r * LPAD + c] -= D[r * LPAD + j] * D[c * LPAD + j];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    double* df = a.Dfac + (long)(k0 / LNB) * (LNB * LNB);
    for (int idx = tid; idx < LNB * LNB; idx += 256) {
      const int r = idx / LNB, c = idx % LNB;
      df[idx] = (r < nb && c <= r) ? D[r * LPAD + c] : 0.0;
    }
  }
  const int row = k0 + nb + blockIdx.x * 256 + tid;
  if (row >= a.nrows) return;
  double X[LNB];
  double* wr = a.W + (long)row * n + k0;
#pragma unroll
  for (int c = 0; c < LNB; c++) X[c] = (c < nb) ? wr[c] : 0.0;
#pragma unroll
  for (int c = 0; c < LNB; c++) {
    if (c < nb) {
      double t = X[c];
#pragma unroll
      for (int k = 0; k < c; k++) t -= X[k] * D[c * LPAD + k];   // wave-uniform LDS address: broadcast
      X[c] = t * rdiag[c];
    }
  }
#pragma unroll
  for (int c = 0; c < LNB; c++)
    if (c < nb) wr[c] = X[c];
}

// trailing update with the panel of block column k0 (width nb): 64 x 64 output tiles, origin s = k0 + nb
__global__ __launch_bounds__(256) void bsl_trailing_kernel(LargeArgs a, int k0) {
  __shared__ double As[64 * LPAD];
  __shared__ double Bs[64 * LPAD];
  const int n = a.n, tid = threadIdx.x;
  const int nb = min(LNB, n - k0), s = k0 + nb;
  const int r0 = s + blockIdx.y * 64, c0 = s + blockIdx.x * 64;
  if (c0 >= n) return;
  const int rlast = min(r0 + 63, a.nrows - 1);
  if (rlast < n && c0 > rlast) return;          // tile strictly above the diagonal of the system rows
  for (int idx = tid; idx < 64 * LNB; idx += 256) {
    const int r = idx / LNB, c = idx % LNB;
    As[r * LPAD + c] = (r0 + r < a.nrows && c < nb) ? a.W[(long)(r0 + r) * n + k0 + c] : 0.0;
    Bs[r * LPAD + c] = (c0 + r < n && c < nb) ? a.W[(long)(c0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;      // lane owns rows ty + 16 u, columns tx + 16 v  (u, v < 4)
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int v = 0; v < 4; v++) acc[u][v] = 0.0;
  for (int c = 0; c < nb; c++) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) av[u] = As[(ty + 16 * u) * LPAD + c];
#pragma unroll
    for (int v = 0; v < 4; v++) bv[v] = Bs[(tx + 16 * v) * LPAD + c];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[u][v] += av[u] * bv[v];
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int r = r0 + ty + 16 * u;
    if (r >= a.nrows) continue;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int c = c0 + tx + 16 * v;
      if (c < n && (r >= n || c <= r)) a.W[(long)r * n + c] -= acc[u][v];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// back substitution L^T x = y by one workgroup; x and y live in LDS
__global__ __launch_bounds__(1024) void bsl_backsolve_kernel(LargeArgs a, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* y = sm;                    // n
  double* D = y + a.n;               // LNB x LPAD
  const int n = a.n, tid = threadIdx.x;
  for (int i = tid; i < n; i += 1024) y[i] = a.W[(long)n * n + i];
  __syncthreads();
  const int nblk = (n + LNB - 1) / LNB;
  for (int jb = nblk - 1; jb >= 0; jb--) {
    const int j0 = jb * LNB, nb = min(LNB, n - j0);
    for (int idx = tid; idx < LNB * LNB; idx += 1024) {
      const int r = idx / LNB, c = idx % LNB;
      if (r < nb && c <= r) D[r * LPAD + c] = a.Dfac[(long)jb * (LNB * LNB) + idx];
    }
    __syncthreads();
    if (tid < 64) {                  // wave 0: column-oriented substitution, lane c owns x[j0 + c]
      double xc = (tid < nb) ? y[j0 + tid] : 0.0;
      for (int c = nb - 1; c >= 0; c--) {
        const double xv = __shfl(xc, c, 64) / D[c * LPAD + c];
        if (tid == c) xc = xv;
        if (tid < c) xc -= D[c * LPAD + tid] * xv;
      }
      if (tid < nb) y[j0 + tid] = xc;
    }
    __syncthreads();
    for (int i = tid; i < j0; i += 1024) {
      double sacc = 0.0;
      for (int c = 0; c < nb; c++) sacc += a.W[(long)(j0 + c) * n + i] * y[j0 + c];
      y[i] -= sacc;
    }
    __syncthreads();
  }
  const bool failed = *a.info != 0;
  for (int i = tid; i < n; i += 1024) dx[i] = failed ? 0.0f : (float)y[i];
}

// L^-1 (f32, lower triangular) out of the identity border rows, which hold its transpose
__global__ __launch_bounds__(256) void bsl_linv_kernel(LargeArgs a, float* __restrict__ Linv) {
  __shared__ double T[32][33];
  const int n = a.n;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;   // output tile rows r0.., columns c0..
  const bool failed = *a.info != 0;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int u = ty; u < 32; u += 8) {                     // read border row (c0 + u), columns r0 + tx
    const int c = c0 + u, r = r0 + tx;
    T[u][tx] = (c < n && r < n) ? a.W[(long)(n + 1 + c) * n + r] : 0.0;
  }
  __syncthreads();
  for (int u = ty; u < 32; u += 8) {
    const int r = r0 + u, c = c0 + tx;
    if (r < n && c < n) Linv[(long)r * n + c] = (failed || c > r) ? 0.0f : (float)T[tx][u];
  }
}

// pose marginals: the 6 x 6 diagonal blocks of (L L^T)^-1 = L^-T L^-1  (:1178-1189); one workgroup per pose
__global__ __launch_bounds__(256) void bsl_sigma_kernel(LargeArgs a, float* __restrict__ sigma_g) {
  __shared__ double red[256];
  const int n = a.n, i = blockIdx.x, tid = threadIdx.x;
  double acc[21];
#pragma unroll
  for (int k = 0; k < 21; k++) acc[k] = 0.0;
  for (int r = 6 * i + tid; r < n; r += 256) {
    double v[6];
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = a.W[(long)(n + 1 + 6 * i + c) * n + r];
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
      for (int q = 0; q <= p; q++) acc[k++] += v[p] * v[q];
  }
  const bool failed = *a.info != 0;
  int k = 0;
  for (int p = 0; p < 6; p++)
    for (int q = 0; q <= p; q++, k++) {
      red[tid] = acc[k];
      __syncthreads();
      for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
      }
      if (tid == 0) {
        const float out = failed ? 0.0f : (float)red[0];
        sigma_g[(long)i * 36 + p * 6 + q] = out;
        sigma_g[(long)i * 36 + q * 6 + p] = out;
      }
      __syncthreads();
    }
}

__global__ void bsl_retract_kernel(const float* __restrict__ dx, float* __restrict__ wTb, float* __restrict__ cTw,
                                   const float* __restrict__ cTb, int kf0, int P, const int32_t* __restrict__ info) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P || *info != 0) return;
  double T[7], dT[7], Tn[7], Ti[7], cb[7], cw[7], xi[6];
  float* wp = wTb + (long)(kf0 + i) * 7;
  for (int k = 0; k < 7; k++) {
    T[k] = (double)wp[k];
    cb[k] = (double)cTb[k];
  }
  for (int k = 0; k < 6; k++) xi[k] = (double)dx[i * 6 + k];
  se3::exp_wv(xi, dT);
  se3::mul(T, dT, Tn);
  const double qn = 1.0 / sqrt(Tn[3] * Tn[3] + Tn[4] * Tn[4] + Tn[5] * Tn[5] + Tn[6] * Tn[6]);
  for (int k = 3; k < 7; k++) Tn[k] *= qn;
  se3::inv(Tn, Ti);
  se3::mul(cb, Ti, cw);
  float* cp = cTw + (long)(kf0 + i) * 7;
  for (int k = 0; k < 7; k++) {
    wp[k] = (float)Tn[k];
    cp[k] = (float)cw[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" size_t ns_ba_solve_large_workspace_bytes(int n6, int want_inv) {
  if (n6 <= 0) return 0;
  const size_t rows = want_inv ? 2 * (size_t)n6 + 1 : (size_t)n6 + 1;
  const size_t nblk = ((size_t)n6 + LNB - 1) / LNB;
  return (rows * (size_t)n6 + nblk * LNB * LNB) * sizeof(double);
}

extern "C" int ns_ba_solve_large(const float* H, const float* v, float* world_T_body, float* cam_T_world,
                                 const float* cam_T_body, const float* prior_pose, float prior_sigma, float ep, float lm,
                                 int kf0, int kf1, int mode, float* dx, double* Hfull_out, float* Linv_out,
                                 float* sigma_g_out, int32_t* info, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  NS_REQUIRE(H && v && dx && info && workspace, "ns_ba_solve_large: null pointer");
  const int P = kf1 - kf0, n = 6 * P;
  NS_REQUIRE(P >= 0, "ns_ba_solve_large: kf1 < kf0");
  NS_REQUIRE(mode == 1 || (world_T_body && cam_T_world && cam_T_body), "ns_ba_solve_large: mode 0 needs the pose buffers");
  if (P == 0) return NS_OK;
  const int want_inv = (Linv_out || sigma_g_out) ? 1 : 0;
  const size_t need = ns_ba_solve_large_workspace_bytes(n, want_inv);
  NS_REQUIRE(workspace_bytes >= need, "ns_ba_solve_large: workspace of %zu B, 6P=%d needs %zu", workspace_bytes, n, need);
  const size_t bs_lds = sizeof(double) * ((size_t)n + LNB * LPAD);
  if (bs_lds > 160 * 1024 - 256) {
    ns_set_error("ns_ba_solve_large: 6P=%d: the back substitution keeps x in LDS (6P <= %d)", n,
                 (int)((160 * 1024 - 256) / sizeof(double) - LNB * LPAD));
    return NS_ENOSUP;
  }
  hipStream_t st = (hipStream_t)stream;
  LargeArgs a;
  a.W = (double*)workspace;
  a.n = n;
  a.nrows = want_inv ? 2 * n + 1 : n + 1;
  a.info = info;
  a.Dfac = a.W + (size_t)a.nrows * n;
  const long total = (long)a.nrows * n;
  hipLaunchKernelGGL(bsl_load_kernel, dim3((unsigned)min((long)2048, (total + 255) / 256)), dim3(256), 0, st, H, v, ep, lm,
                     a, want_inv);
  NS_CHECK_LAUNCH("bsl_load_kernel");
  if (prior_pose != nullptr) {
    NS_REQUIRE(world_T_body, "ns_ba_solve_large: the prior needs world_T_body");
    hipLaunchKernelGGL(bsl_prior_kernel, dim3(1), dim3(64), 0, st, a, prior_pose, world_T_body, kf0, prior_sigma);
    NS_CHECK_LAUNCH("bsl_prior_kernel");
  }
  if (Hfull_out != nullptr) {
    hipLaunchKernelGGL(bsl_hfull_kernel, dim3((unsigned)min((long)2048, ((long)n * n + 255) / 256)), dim3(256), 0, st, a,
                       Hfull_out);
    NS_CHECK_LAUNCH("bsl_hfull_kernel");
  }
  for (int k0 = 0; k0 < n; k0 += LNB) {
    const int nb = n - k0 < LNB ? n - k0 : LNB, s = k0 + nb;
    const int below = a.nrows - s;                   // >= 1: the border row
    hipLaunchKernelGGL(bsl_panel_kernel, dim3(below > 0 ? ns_cdiv(below, 256) : 1), dim3(256), 0, st, a, k0);
    NS_CHECK_LAUNCH("bsl_panel_kernel");
    if (s < n) {
      hipLaunchKernelGGL(bsl_trailing_kernel, dim3(ns_cdiv(n - s, 64), ns_cdiv(a.nrows - s, 64)), dim3(256), 0, st, a, k0);
      NS_CHECK_LAUNCH("bsl_trailing_kernel");
    }
  }
  static thread_local size_t configured = 0;
  if (bs_lds > configured && bs_lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)bsl_backsolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bs_lds) !=
        hipSuccess) {
      ns_set_error("ns_ba_solve_large: hipFuncSetAttribute(%zu) failed", bs_lds);
      return NS_ELAUNCH;
    }
    configured = bs_lds;
  }
  hipLaunchKernelGGL(bsl_backsolve_kernel, dim3(1), dim3(1024), bs_lds, st, a, dx);
  NS_CHECK_LAUNCH("bsl_backsolve_kernel");
  if (mode == 0) {
    hipLaunchKernelGGL(bsl_retract_kernel, dim3(ns_cdiv(P, 64)), dim3(64), 0, st, dx, world_T_body, cam_T_world,
                       cam_T_body, kf0, P, info);
    NS_CHECK_LAUNCH("bsl_retract_kernel");
  }
  if (Linv_out != nullptr) {
    hipLaunchKernelGGL(bsl_linv_kernel, dim3(ns_cdiv(n, 32), ns_cdiv(n, 32)), dim3(256), 0, st, a, Linv_out);
    NS_CHECK_LAUNCH("bsl_linv_kernel");
  }
  if (sigma_g_out != nullptr) {
    hipLaunchKernelGGL(bsl_sigma_kernel, dim3(P), dim3(256), 0, st, a, sigma_g_out);
    NS_CHECK_LAUNCH("bsl_sigma_kernel");
  }
  return NS_OK;
}
