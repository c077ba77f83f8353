// ba_solve.hip -- device-resident replacement of the GTSAM round trip inside
// RaftVisualFrontend.ba() (slam/visual_frontends/visual_frontend.py:1123-1230) for gfx950.
//
// The reference copies H block by block to the host (O(N^2) synchronising .cpu().numpy() calls,
// :1127-1134), lets GTSAM [EXTERNAL] solve the dense system and retract the poses on the CPU,
// and copies poses and deltas back (:1149-1160).  Here one workgroup does all of it in fp64 out
// of LDS: packed lower-triangular Cholesky blocked by the natural 6x6 pose blocks, the two
// triangular solves, the pose retraction, and (optionally) L^-1 and the 6x6 pose marginals that
// the covariance block needs (:1165-1189).  No host involvement, one launch.
//
// GTSAM semantics assumed (un-vendored, unpinned -- see DESIGN.md): the Hessian factors use the
// upper triangle of H; Pose3 retract is the full SE3 exponential applied on the right with
// tangent order [omega, v]; the frame-0 prior contributes I/sigma^2 and -Log(prior^-1 x0)/sigma^2.
#include "common.h"
#include "se3.h"

#define SOLVE_THREADS 256

__device__ __forceinline__ int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // c <= r

struct SolveArgs {
  const float* H;      // [n,n]
  const float* v;      // [n]
  float* wTb;          // world_T_body [*,7]   (mode 0)
  float* cTw;          // cam_T_world  [*,7]   (mode 0)
  const float* cTb;    // cam_T_body [7]
  const float* prior;  // [7] or null
  float prior_sigma;
  float ep, lm;        // (H + ep + lm*diag(H)) as SparseBlock::solve (:1318-1340); 0,0 in the live path
  int kf0, P;
  int mode;            // 0: live path (right retraction of world_T_body), 1: solve only
  int want_inv;        // 1: carry n identity border rows through the factorisation (they become L^-1)
  float* dx;           // [P,6]
  double* Hfull;       // [n,n] or null: the symmetric system actually solved (with prior)
  float* Linv;         // [n,n] or null: inverse Cholesky factor, lower triangular, f32
  double* Linv_ws;     // unused (kept for ABI stability)
  float* sigma_g;      // [P,6,6] or null: diagonal blocks of (L L^T)^-1
  int32_t* info;
};

// 6x6 lower-triangular block helpers; D[r(r+1)/2+c] = L[j0+r][j0+c]
__device__ __forceinline__ void load_diag(const double* Lp, int j0, double* D) {
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) D[r * (r + 1) / 2 + c] = Lp[tri(j0 + r, j0 + c)];
}

// The system is stored as a packed lower triangle with ONE EXTRA ROW holding the right-hand side:
// factoring the bordered matrix [[A, b],[b^T, .]] leaves y = L^-1 b in that row, so the forward
// substitution costs no extra barriers.
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int P = a.P, n = 6 * P;
  const int ntri = n * (n + 1) / 2;
  double* Lp = lds;          // rows 0..n-1: factor; row n (offset ntri): rhs -> y
  double* yrow = lds + ntri;  // n
  double* x = yrow + n;       // n
  double* rd = x + n;         // n: reciprocals of the factor's diagonal
  double* Xb = rd + n;        // [n][n] identity border rows -> Xb[c][k] = (L^-1)[k][c]  (only when a.want_inv)
  const int nrows = a.want_inv ? 2 * n + 1 : n + 1;  // matrix rows + rhs row (+ n identity rows)
  auto rowp = [&](int i, int c0) -> double* {  // address of entry (i, c0); border rows are dense
    return (i < n) ? Lp + tri(i, c0) : (i == n ? yrow + c0 : Xb + (long)(i - n - 1) * n + c0);
  };
  __shared__ int fail;
  if (tid == 0) fail = 0;

  // ---- load the upper triangle of H (what the HessianFactors keep) ----
  // eight independent global loads per thread in flight, then the LDS stores (a load -> store loop ran at one global
  // round trip per 256 elements: 6 us for n = 60)
  for (int base = 0; base < n * n; base += 8 * SOLVE_THREADS) {
    float hv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int idx = base + tid + u * SOLVE_THREADS;
      const int r = idx / n, c = idx - r * n;
      hv[u] = (idx < n * n && c <= r) ? a.H[(long)c * n + r] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int idx = base + tid + u * SOLVE_THREADS;
      const int r = idx / n, c = idx - r * n;
      if (idx < n * n && c <= r) {
        double h = (double)hv[u];
        if (c == r) h += (double)a.ep + (double)a.lm * h;
        Lp[tri(r, c)] = h;
      }
    }
  }
  for (int r = tid; r < n; r += SOLVE_THREADS) yrow[r] = (double)a.v[r];
  if (a.want_inv)
    for (int idx = tid; idx < n * n; idx += SOLVE_THREADS) Xb[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
  __syncthreads();
  if (a.prior != nullptr && tid == 0) {
    double pr[7], x0[7], pinv[7], rel[7], e[6];
    for (int k = 0; k < 7; k++) {
      pr[k] = (double)a.prior[k];
      x0[k] = (double)a.wTb[(long)a.kf0 * 7 + k];
    }
    se3::inv(pr, pinv);
    se3::mul(pinv, x0, rel);
    se3::log_wv(rel, e);
    const double info = 1.0 / ((double)a.prior_sigma * (double)a.prior_sigma);
    for (int k = 0; k < 6; k++) {
      Lp[tri(k, k)] += info;
      yrow[k] += -e[k] * info;
    }
  }
  __syncthreads();
  if (a.Hfull != nullptr) {
    for (int idx = tid; idx < n * n; idx += SOLVE_THREADS) {
      const int r = idx / n, c = idx - r * n;
      a.Hfull[idx] = (c <= r) ? Lp[tri(r, c)] : Lp[tri(c, r)];
    }
    __syncthreads();
  }

  // ---- blocked right-looking Cholesky of the bordered system, block = one pose (6), with LOOK-AHEAD ----
  // The 6x6 diagonal factor is a serial fp64 chain on one lane (0.7 us per block: a quarter of the kernel when it sat
  // between two barriers).  Wave 0 is the diagonal specialist: during the trailing update of block jb it first applies
  // that update to the NEXT diagonal block (21 entries, all the trailing entries the next block's rows have), factors it,
  // and the other three waves meanwhile do the rest of the trailing update.
  auto factor_diag = [&](int j0) {  // lane 0 of wave 0
    double D[21];
    load_diag(Lp, j0, D);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double s = D[j * (j + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= D[j * (j + 1) / 2 + k] * D[j * (j + 1) / 2 + k];
      if (!(s > 0.0)) {
        fail = j0 + j + 1;
        s = 1.0;
      }
      // 1/sqrt(s) from the hardware estimate + two Newton steps (each squares the error: 2^-26 -> < 2^-100), then
      // d = s * di: ~12 dependent ops on the critical path of every block column instead of the ~40 of an IEEE
      // sqrt followed by an IEEE divide
      double di = __builtin_amdgcn_rsq(s);
      di = di * (1.5 - 0.5 * s * di * di);
      di = di * (1.5 - 0.5 * s * di * di);
      const double d = s * di;
      D[j * (j + 1) / 2 + j] = d;
      rd[j0 + j] = di;
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        double t = D[i * (i + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= D[i * (i + 1) / 2 + k] * D[j * (j + 1) / 2 + k];
        D[i * (i + 1) / 2 + j] = t * di;
      }
    }
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) Lp[tri(j0 + r, j0 + c)] = D[r * (r + 1) / 2 + c];
  };
  if (tid == 0) factor_diag(0);
  __syncthreads();
  for (int jb = 0; jb < P; jb++) {
    const int j0 = 6 * jb;
    // panel: rows below the diagonal block (row n = rhs),  X * Ljj^T = A_ij
    for (int i = j0 + 6 + tid; i < nrows; i += SOLVE_THREADS) {
      double D[21], X[6], R[6];
      load_diag(Lp, j0, D);
      double* row = rowp(i, j0);
#pragma unroll
      for (int c = 0; c < 6; c++) {
        X[c] = row[c];
        R[c] = rd[j0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double t = X[c];
#pragma unroll
        for (int k = 0; k < c; k++) t -= X[k] * D[c * (c + 1) / 2 + k];
        X[c] = t * R[c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) row[c] = X[c];
    }
    __syncthreads();
    const bool has_next = jb + 1 < P;
    if (tid < 64) {
      if (has_next) {
        // the next diagonal block: lane l < 21 <-> entry (i, k), k <= i, of rows j0+6 .. j0+11
        if (tid < 21) {
          int li = 0;
          while ((li + 1) * (li + 2) / 2 <= tid) li++;
          const int lk = tid - li * (li + 1) / 2;
          const double* ri = Lp + tri(j0 + 6 + li, j0);
          const double* rk = Lp + tri(j0 + 6 + lk, j0);
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < 6; c++) s += ri[c] * rk[c];
          Lp[tri(j0 + 6 + li, j0 + 6 + lk)] -= s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: the 21 updates are in LDS before lane 0 reads them
        if (tid == 0) factor_diag(j0 + 6);
      }
    } else {
      // trailing update by the other three waves on a 16 x 12 thread grid; rows of the next block are wave 0's
      const int t2 = tid - 64, tx = t2 & 15, ty = t2 >> 4;
      for (int i = j0 + 12 + ty; i < nrows; i += 12) {
        double Ri[6];
        const double* ri = rowp(i, j0);
#pragma unroll
        for (int c = 0; c < 6; c++) Ri[c] = ri[c];
        const int kmax = (i < n) ? i : n - 1;  // border rows have no diagonal entry
        double* out = rowp(i, 0);
        for (int k = j0 + 6 + tx; k <= kmax; k += 16) {
          const double* rk = Lp + tri(k, j0);
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < 6; c++) s += Ri[c] * rk[c];
          out[k] -= s;
        }
      }
    }
    __syncthreads();
  }

  // ---- backward substitution L^T x = y ----
  for (int jb = P - 1; jb >= 0; jb--) {
    const int j0 = 6 * jb;
    if (tid == 0) {
      double D[21], Y[6], X[6];
      load_diag(Lp, j0, D);
#pragma unroll
      for (int c = 0; c < 6; c++) Y[c] = yrow[j0 + c];
#pragma unroll
      for (int c = 5; c >= 0; c--) {
        double t = Y[c];
#pragma unroll
        for (int k = c + 1; k < 6; k++) t -= D[k * (k + 1) / 2 + c] * X[k];
        X[c] = t * rd[j0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) x[j0 + c] = X[c];
    }
    __syncthreads();
    for (int i = tid; i < j0; i += SOLVE_THREADS) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) s += Lp[tri(j0 + c, i)] * x[j0 + c];
      yrow[i] -= s;
    }
    __syncthreads();
  }
  const bool failed = fail != 0;
  if (tid == 0) *a.info = fail;

  // ---- delta and retraction ----
  for (int k = tid; k < n; k += SOLVE_THREADS) a.dx[k] = failed ? 0.0f : (float)x[k];
  if (a.mode == 0 && !failed) {
    for (int i = tid; i < P; i += SOLVE_THREADS) {
      double T[7], dT[7], Tn[7], Ti[7], cb[7], cw[7];
      float* wp = a.wTb + (long)(a.kf0 + i) * 7;
      for (int k = 0; k < 7; k++) {
        T[k] = (double)wp[k];
        cb[k] = (double)a.cTb[k];
      }
      se3::exp_wv(&x[6 * i], dT);
      se3::mul(T, dT, Tn);
      const double qn = 1.0 / sqrt(Tn[3] * Tn[3] + Tn[4] * Tn[4] + Tn[5] * Tn[5] + Tn[6] * Tn[6]);
      for (int k = 3; k < 7; k++) Tn[k] *= qn;
      se3::inv(Tn, Ti);
      se3::mul(cb, Ti, cw);  // cam_T_world = cam_T_body * world_T_body^-1   (visual_frontend.py:1158)
      float* cp = a.cTw + (long)(a.kf0 + i) * 7;
      for (int k = 0; k < 7; k++) {
        wp[k] = (float)Tn[k];
        cp[k] = (float)cw[k];
      }
    }
  }

  // ---- L^-1 (already sitting in the identity border rows) and the pose marginals ----
  if (a.want_inv) {
    if (a.Linv != nullptr)
      for (int idx = tid; idx < n * n; idx += SOLVE_THREADS) {
        const int r = idx / n, c = idx - r * n;
        a.Linv[idx] = (failed || c > r) ? 0.0f : (float)Xb[(long)c * n + r];
      }
    if (a.sigma_g != nullptr) {
      for (int t = tid; t < 36 * P; t += SOLVE_THREADS) {
        const int i = t / 36, aa = (t % 36) / 6, bb = t % 6;
        const double* ca = Xb + (long)(6 * i + aa) * n;
        const double* cb = Xb + (long)(6 * i + bb) * n;
        double s0 = 0.0, s1 = 0.0;
        int r = 6 * i;
        for (; r + 1 < n; r += 2) {
          s0 += ca[r] * cb[r];
          s1 += ca[r + 1] * cb[r + 1];
        }
        if (r < n) s0 += ca[r] * cb[r];
        a.sigma_g[t] = failed ? 0.0f : (float)(s0 + s1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// depth covariances (visual_frontend.py:1191-1230): one lane per (depth slot, pixel)
//   z_cov = Q + sum_j ( Q * sum_{rows n of the slot, pose a in window} E_n[:,px] . Linv[6a+:, j] )^2
// (the reference right-multiplies by L^-1, :1215; reproduced as is).  Column tiles of 48 keep the
// accumulators in registers; Linv entries are wave-uniform scalar operands.
// ---------------------------------------------------------------------------------------------
#define COV_TILE 32
__global__ __launch_bounds__(256) void ba_depth_cov_kernel(const float* __restrict__ Linv,
                                                           const float* __restrict__ Q, const float* __restrict__ E,
                                                           const int32_t* __restrict__ row_pose,
                                                           const int32_t* __restrict__ slot_rows_ptr,
                                                           const int32_t* __restrict__ slot_rows, int HW, int P,
                                                           float* __restrict__ z_cov) {
  extern __shared__ __attribute__((aligned(16))) float Ls[];  // Linv, row stride ns (multiple of 32)
  const int k = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  const int n = 6 * P;
  const int ns = (n + COV_TILE - 1) / COV_TILE * COV_TILE;
  for (int idx = threadIdx.x; idx < n * ns; idx += 256) {
    const int r = idx / ns, c = idx - r * ns;
    Ls[idx] = (c < n) ? Linv[(long)r * n + c] : 0.0f;
  }
  __syncthreads();
  if (p >= HW) return;
  const float q = Q[(long)k * HW + p];
  float total = 0.0f;
  const int r0 = slot_rows_ptr[k], r1 = slot_rows_ptr[k + 1];
  for (int j0 = 0; j0 < ns; j0 += COV_TILE) {
    float acc[COV_TILE];
#pragma unroll
    for (int j = 0; j < COV_TILE; j++) acc[j] = 0.0f;
    for (int r = r0; r < r1; r++) {
      const int row = slot_rows[r];
      const int pose = row_pose[row];
      if (pose < 0 || pose >= P) continue;
      if (6 * pose + 5 < j0) continue;  // Linv is lower triangular: rows above the tile contribute nothing
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const float e = E[((long)row * 6 + c) * HW + p] * q;
        const float4* __restrict__ Lr = reinterpret_cast<const float4*>(Ls + (6 * pose + c) * ns + j0);
#pragma unroll
        for (int j = 0; j < COV_TILE / 4; j++) {
          const float4 l = Lr[j];  // wave-uniform address: LDS broadcast
          acc[4 * j + 0] += e * l.x;
          acc[4 * j + 1] += e * l.y;
          acc[4 * j + 2] += e * l.z;
          acc[4 * j + 3] += e * l.w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < COV_TILE; j++) total += acc[j] * acc[j];
  }
  z_cov[(long)k * HW + p] = q + total;
}

// The same sum for windows whose L^-1 does not fit LDS whole (6P x ceil32(6P) floats > 160 KB, P >= 33): L^-1 is staged
// one 32-column tile and COVC_ROWS rows at a time; the accumulators of a column tile stay in registers across the row
// chunks.  Chunk boundaries are multiples of 6, so the six rows of a pose never straddle two chunks.
#define COVC_ROWS 1020
__global__ __launch_bounds__(256) void ba_depth_cov_chunked_kernel(const float* __restrict__ Linv,
                                                                   const float* __restrict__ Q,
                                                                   const float* __restrict__ E,
                                                                   const int32_t* __restrict__ row_pose,
                                                                   const int32_t* __restrict__ slot_rows_ptr,
                                                                   const int32_t* __restrict__ slot_rows, int HW, int P,
                                                                   float* __restrict__ z_cov) {
  extern __shared__ __attribute__((aligned(16))) float Ls[];  // [COVC_ROWS][COV_TILE]
  const int k = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  const bool live = p < HW;
  const int n = 6 * P;
  const float q = live ? Q[(long)k * HW + p] : 0.0f;
  float total = 0.0f;
  const int r0 = slot_rows_ptr[k], r1 = slot_rows_ptr[k + 1];
  for (int j0 = 0; j0 < n; j0 += COV_TILE) {
    float acc[COV_TILE];
#pragma unroll
    for (int j = 0; j < COV_TILE; j++) acc[j] = 0.0f;
    for (int rc0 = (j0 / 6) * 6; rc0 < n; rc0 += COVC_ROWS) {   // L^-1 is lower triangular: rows < j0 hold zeros here
      const int rows = min(COVC_ROWS, n - rc0);
      __syncthreads();
      for (int idx = threadIdx.x; idx < rows * COV_TILE; idx += 256) {
        const int r = idx / COV_TILE, c = idx % COV_TILE;
        Ls[idx] = (j0 + c < n) ? Linv[(long)(rc0 + r) * n + j0 + c] : 0.0f;
      }
      __syncthreads();
      if (!live) continue;
      for (int r = r0; r < r1; r++) {
        const int row = slot_rows[r];
        const int pose = row_pose[row];
        if (pose < 0 || pose >= P) continue;
        const int lr = 6 * pose - rc0;
        if (lr < 0 || lr >= rows) continue;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const float e = E[((long)row * 6 + c) * HW + p] * q;
          const float4* __restrict__ Lr = reinterpret_cast<const float4*>(Ls + (lr + c) * COV_TILE);
#pragma unroll
          for (int j = 0; j < COV_TILE / 4; j++) {
            const float4 l = Lr[j];
            acc[4 * j + 0] += e * l.x;
            acc[4 * j + 1] += e * l.y;
            acc[4 * j + 2] += e * l.z;
            acc[4 * j + 3] += e * l.w;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < COV_TILE; j++) total += acc[j] * acc[j];
  }
  if (live) z_cov[(long)k * HW + p] = q + total;
}

// standalone retraction (used after the large-system rocSOLVER path): same arithmetic as above
__global__ void ba_retract_kernel(const float* __restrict__ dx, float* __restrict__ wTb, float* __restrict__ cTw,
                                  const float* __restrict__ cTb, int kf0, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
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

extern "C" int ns_ba_retract(const float* dx, float* world_T_body, float* cam_T_world, const float* cam_T_body,
                             int kf0, int kf1, void* stream) {
  NS_REQUIRE(dx && world_T_body && cam_T_world && cam_T_body, "ns_ba_retract: null pointer");
  if (kf1 <= kf0) return NS_OK;
  hipLaunchKernelGGL(ba_retract_kernel, dim3(ns_cdiv(kf1 - kf0, 64)), dim3(64), 0, (hipStream_t)stream, dx,
                     world_T_body, cam_T_world, cam_T_body, kf0, kf1 - kf0);
  NS_CHECK_LAUNCH("ba_retract_kernel");
  return NS_OK;
}

extern "C" int ns_ba_solve(const float* H, const float* v, float* world_T_body, float* cam_T_world,
                           const float* cam_T_body, const float* prior_pose, float prior_sigma, float ep, float lm,
                           int kf0, int kf1, int mode, float* dx, double* Hfull_out, float* Linv_out,
                           double* Linv_ws, float* sigma_g_out, int32_t* info, void* stream) {
  NS_REQUIRE(H && v && dx && info, "ns_ba_solve: null pointer");
  const int P = kf1 - kf0, n = 6 * P;
  NS_REQUIRE(P >= 0, "ns_ba_solve: kf1 < kf0");
  NS_REQUIRE(mode == 1 || (world_T_body && cam_T_world && cam_T_body), "ns_ba_solve: mode 0 needs the pose buffers");
  if (P == 0) return NS_OK;
  const size_t tri_b = sizeof(double) * ((size_t)n * (n + 1) / 2);
  size_t lds = tri_b + sizeof(double) * 3 * (size_t)n;
  const size_t lds_max = 160 * 1024 - 256;
  if (lds > lds_max) {
    ns_set_error("ns_ba_solve: 6P=%d needs %zu B of LDS (> 160 KiB): use the large-system path", n, lds);
    return NS_ENOSUP;
  }
  const int want_inv = (Linv_out || sigma_g_out) ? 1 : 0;
  if (want_inv) {
    lds += sizeof(double) * (size_t)n * n;
    if (lds > lds_max) {
      ns_set_error("ns_ba_solve: 6P=%d with covariances needs %zu B of LDS (> 160 KiB): use the large-system path", n,
                   lds);
      return NS_ENOSUP;
    }
  }
  static thread_local size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute((const void*)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      ns_set_error("ns_ba_solve: hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
      return NS_ELAUNCH;
    }
    configured = lds;
  }
  SolveArgs a;
  a.H = H;
  a.v = v;
  a.wTb = world_T_body;
  a.cTw = cam_T_world;
  a.cTb = cam_T_body;
  a.prior = prior_pose;
  a.prior_sigma = prior_sigma;
  a.ep = ep;
  a.lm = lm;
  a.kf0 = kf0;
  a.P = P;
  a.mode = mode;
  a.want_inv = want_inv;
  a.dx = dx;
  a.Hfull = Hfull_out;
  a.Linv = Linv_out;
  a.Linv_ws = Linv_ws;
  a.sigma_g = sigma_g_out;
  a.info = info;
  hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(SOLVE_THREADS), lds, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ba_solve_kernel");
  return NS_OK;
}

extern "C" int ns_ba_depth_cov(const float* Linv, const float* Q, const float* E, const ns_ba_plan* plan,
                               const int32_t* index, const size_t* off, int HW, float* z_cov, void* stream) {
  NS_REQUIRE(Linv && Q && E && plan && index && off && z_cov, "ns_ba_depth_cov: null pointer");
  if (plan->K == 0) return NS_OK;
  const int n = 6 * plan->P, ns = (n + COV_TILE - 1) / COV_TILE * COV_TILE;
  const size_t lds = sizeof(float) * (size_t)n * ns;
  if (lds > 160 * 1024 - 256) {
    // window too large to keep L^-1 in LDS whole (P >= 33): the chunked kernel (the reference has no such limit)
    const size_t ldc = sizeof(float) * (size_t)COVC_ROWS * COV_TILE;
    static thread_local bool configured_c = false;
    if (!configured_c) {
      if (hipFuncSetAttribute((const void*)ba_depth_cov_chunked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)ldc) != hipSuccess) {
        ns_set_error("ns_ba_depth_cov: hipFuncSetAttribute failed");
        return NS_ELAUNCH;
      }
      configured_c = true;
    }
    hipLaunchKernelGGL(ba_depth_cov_chunked_kernel, dim3(plan->K, ns_cdiv(HW, 256)), dim3(256), ldc, (hipStream_t)stream,
                       Linv, Q, E, index + off[2], index + off[6], index + off[7], HW, plan->P, z_cov);
    NS_CHECK_LAUNCH("ba_depth_cov_chunked_kernel");
    return NS_OK;
  }
  static thread_local size_t configured = 0;
  if (lds > configured && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)ba_depth_cov_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      ns_set_error("ns_ba_depth_cov: hipFuncSetAttribute failed");
      return NS_ELAUNCH;
    }
    configured = lds;
  }
  hipLaunchKernelGGL(ba_depth_cov_kernel, dim3(plan->K, ns_cdiv(HW, 256)), dim3(256), lds, (hipStream_t)stream, Linv, Q, E,
                     index + off[2], index + off[6], index + off[7], HW, plan->P, z_cov);
  NS_CHECK_LAUNCH("ba_depth_cov_kernel");
  return NS_OK;
}
