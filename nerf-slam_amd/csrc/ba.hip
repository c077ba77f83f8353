// ba.hip -- Schur-reduced dense bundle adjustment on gfx950 (MI355X).
//
// Replaces, with no host round trip anywhere:
//   K1  projective_transform_kernel          src/droid_kernels.cu:192-536
//   K6  accum_cuda + accum_kernel            :971-991, 1065-1115
//   K9  EEt6x6_kernel, K10 Ev6x1_kernel      :1118-1173, 1176-1210  (+ schur_block :1349-1438)
//   SparseBlock (Eigen, host, fp64)          :1240-1316
//   K11 EvT6x1_kernel, K8 disp_retr_kernel   :1213-1238, 1050-1063  (solve_depth_cuda :1772-1825)
//   K7  pose_retr_kernel                     :1015-1048
// and the GTSAM solve/retract of RaftVisualFrontend.ba() (visual_frontend.py:1123-1158).
//
// Structure of one linearisation (4 launches + 1 memset, 0 host syncs):
//   linearize   grid (edge, pixel chunk).  Prologue: per-edge constants G_ij and the two 6x6 maps A_i, A_j
//               with J_i = J_raw A_i, J_j = J_raw A_j (the reference applies them per pixel, :376-403;
//               they are linear) into LDS.  Per pixel residual/weights/J_raw, writes Ejz/Eiz/C/b,
//               accumulates only G = sum w J_raw^T J_raw (21) and g = sum w r J_raw (6) per lane
//               instead of the reference's 78+12 (Hii = A_i^T G A_i, Hij = A_i^T G A_j, ...),
//               block-reduces them with wave shuffles and adds the transformed 6x6 blocks into a
//               dense fp64 system with fp64 atomics (the reference: 90 serial block reductions,
//               then a D2H copy and Eigen triplets)
//   accum       per (depth slot, pixel): CSR sum of C/b/Eiz over the slot's edges -> Q, w, Ei
//   schur       one workgroup per row pair of a slot (upper half only), subtracts E Q E^T / E Q w
//               from the same fp64 system
//   finalize    fp64 -> fp32 H, v  (transposed like SparseBlock::get_dense, :1305-1316)
#include "common.h"
#include "se3.h"

#define ET_STRIDE 80  // floats per edge in the edge table
#define ET_T 0
#define ET_Q 3
#define ET_STEREO 7
#define ET_AI 8
#define ET_AJ 44

// ---------------------------------------------------------------------------------------------
// edge_prep: one lane per edge
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void reorder_wt(float* J) {  // [t,w] -> [w,t]   (:387-403)
  const float a = J[0], b = J[1], c = J[2];
  J[0] = J[3];
  J[1] = J[4];
  J[2] = J[5];
  J[3] = a;
  J[4] = b;
  J[5] = c;
}

// Per-edge constants, computed by the first lanes of every workgroup of that edge into LDS:
//   Ts[0..2] t_ij, Ts[3..6] q_ij, Ts[7] stereo flag, Ts[8..43] A_i, Ts[44..79] A_j   (row-major 6x6)
__device__ __forceinline__ void edge_constants(const float* __restrict__ poses, const float* __restrict__ extr,
                                               int ix, int jx, float* Ts) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    float tij[3], qij[4];
    float stereo = 0.0f;
    if (ix == jx) {  // stereo pair (:249-259)
      tij[0] = -0.1f;
      tij[1] = 0.0f;
      tij[2] = 0.0f;
      qij[0] = qij[1] = qij[2] = 0.0f;
      qij[3] = 1.0f;
      stereo = 1.0f;
    } else {
      se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3,
                   tij, qij);
    }
    Ts[ET_T + 0] = tij[0];
    Ts[ET_T + 1] = tij[1];
    Ts[ET_T + 2] = tij[2];
    Ts[ET_Q + 0] = qij[0];
    Ts[ET_Q + 1] = qij[1];
    Ts[ET_Q + 2] = qij[2];
    Ts[ET_Q + 3] = qij[3];
    Ts[ET_STEREO] = stereo;
  }
  __syncthreads();
  if (tid < 6) {
    const int k = tid;
    const float tij[3] = {Ts[ET_T], Ts[ET_T + 1], Ts[ET_T + 2]};
    const float qij[4] = {Ts[ET_Q], Ts[ET_Q + 1], Ts[ET_Q + 2], Ts[ET_Q + 3]};
    const float ext_t[3] = {extr[0], extr[1], extr[2]};
    const float ext_q[4] = {extr[3], extr[4], extr[5], extr[6]};
    float X[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int n = 0; n < 6; n++) X[n] = (n == k) ? 1.0f : 0.0f;
    float Ji[6], Jj[6], tmp[6];
    // Ji = -adj(G_ij, Jj)                        (:376-377)
    se3::adj_se3(tij, qij, X, Ji, false);
#pragma unroll
    for (int n = 0; n < 6; n++) Ji[n] = -Ji[n];
    // camera-to-body adjoint, applied in place by the reference (:380-381)
    se3::adj_se3(ext_t, ext_q, X, tmp, true);
#pragma unroll
    for (int n = 0; n < 6; n++) Jj[n] = -tmp[n];  // (:384)
    se3::adj_se3(ext_t, ext_q, Ji, tmp, true);
#pragma unroll
    for (int n = 0; n < 6; n++) Ji[n] = -tmp[n];  // (:385)
    reorder_wt(Jj);
    reorder_wt(Ji);
#pragma unroll
    for (int c = 0; c < 6; c++) {
      Ts[ET_AI + k * 6 + c] = Ji[c];
      Ts[ET_AJ + k * 6 + c] = Jj[c];
    }
  }
  __syncthreads();
}

// Entry `t` of the per-edge pose blocks from G (21 upper-triangular sums) and g (6):
//   t < 144: block t/36 of {Hii, Hij, Hji, Hjj} = X^T G Y, entry ((t%36)/6, t%6);  t in [144,156): vi / vj = X^T g
__device__ __forceinline__ double edge_block_entry(const float* Ai, const float* Aj, const double* Gs, int t) {
  double val = 0.0;
  if (t < 144) {
    const int blk = t / 36, r = (t % 36) / 6, c = t % 6;
    const float* X = (blk < 2) ? Ai : Aj;
    const float* Y = (blk % 2 == 0) ? Ai : Aj;
    for (int k = 0; k < 6; k++) {
      double s = 0.0;
      for (int m = 0; m < 6; m++) {
        const int lo = k < m ? k : m, hi = k < m ? m : k;
        s += Gs[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)] * (double)Y[m * 6 + c];
      }
      val += (double)X[k * 6 + r] * s;
    }
  } else {
    const int side = (t - 144) / 6, r = (t - 144) % 6;
    const float* X = side == 0 ? Ai : Aj;
    for (int k = 0; k < 6; k++) val += (double)X[k * 6 + r] * Gs[21 + k];
  }
  return val;
}

// ---------------------------------------------------------------------------------------------
// linearize (K1)
// ---------------------------------------------------------------------------------------------
struct LinArgs {
  const float* target;   // [M,2,HW]
  const float* weight;   // [M,2,HW]
  const float* disps;    // [*,HW]
  const float* intr;     // [4]
  const int64_t* ii;
  const int64_t* jj;
  const float* poses;    // [*,7]
  const float* extr;     // [7]
  float* Eiz;            // [M,6,HW]
  float* Ejz;            // [M,6,HW]
  float* Cii;            // [M,HW]
  float* bz;             // [M,HW]
  float* partial;        // [M,nch,32]: per-(edge,chunk) sums G(21), g(6)   (assemble mode)
  float* Hs;             // [4,M,6,6] (per-edge mode, nch == 1)
  float* vs;             // [2,M,6]
  int M, HW, wd, nch, kf0, P;
};

template <bool PER_EDGE>
__global__ __launch_bounds__(256, 4) void ba_linearize_kernel(LinArgs a) {
  const int e = blockIdx.x;
  const int ch = blockIdx.y;
  const int tid = threadIdx.x;
  const int HW = a.HW;
  const int ix = (int)a.ii[e], jx = (int)a.jj[e];
  __shared__ float T[ET_STRIDE];
  edge_constants(a.poses, a.extr, ix, jx, T);

  const float fx = a.intr[0], fy = a.intr[1], cx = a.intr[2], cy = a.intr[3];
  const float tij[3] = {T[ET_T], T[ET_T + 1], T[ET_T + 2]};
  const float qij[4] = {T[ET_Q], T[ET_Q + 1], T[ET_Q + 2], T[ET_Q + 3]};
  const bool stereo = T[ET_STEREO] != 0.0f;

  float G[21];
  float g[6];
#pragma unroll
  for (int l = 0; l < 21; l++) G[l] = 0.0f;
#pragma unroll
  for (int l = 0; l < 6; l++) g[l] = 0.0f;

  const int chunk = (HW + a.nch - 1) / a.nch;
  const int p0 = ch * chunk;
  const int p1 = min(HW, p0 + chunk);

  const float* __restrict__ disp = a.disps + (long)ix * HW;
  const float* __restrict__ tu = a.target + ((long)e * 2 + 0) * HW;
  const float* __restrict__ tv = a.target + ((long)e * 2 + 1) * HW;
  const float* __restrict__ wu_ = a.weight + ((long)e * 2 + 0) * HW;
  const float* __restrict__ wv_ = a.weight + ((long)e * 2 + 1) * HW;
  float* __restrict__ oC = a.Cii + (long)e * HW;
  float* __restrict__ ob = a.bz + (long)e * HW;
  float* __restrict__ oEi = a.Eiz + (long)e * 6 * HW;
  float* __restrict__ oEj = a.Ejz + (long)e * 6 * HW;

  for (int p = p0 + tid; p < p1; p += 256) {
    // keep the 72 entries of A_i/A_j in LDS (broadcast reads) instead of letting the compiler hoist
    // them into 72 VGPRs per lane, which halves the occupancy of this latency-bound kernel
    asm volatile("" ::: "memory");
    const int i = p / a.wd, j = p - i * a.wd;
    const float u = (float)j, v = (float)i;
    float Xi[4], Xj[4];
    Xi[0] = (u - cx) / fx;
    Xi[1] = (v - cy) / fy;
    Xi[2] = 1.0f;
    Xi[3] = disp[p];
    se3::act_se3(tij, qij, Xi, Xj);
    const float x = Xj[0], y = Xj[1], h = Xj[3];
    const bool ok = !(Xj[2] < NS_MIN_DEPTH);
    const float d = ok ? 1.0f / Xj[2] : 0.0f;
    const float d2 = d * d;
    // `.001 * weight` is a double product in the reference (:344-345)
    float wu = ok ? (float)(0.001 * (double)wu_[p]) : 0.0f;
    float wv = ok ? (float)(0.001 * (double)wv_[p]) : 0.0f;
    const float ru = tu[p] - (fx * d * x + cx);
    const float rv = tv[p] - (fy * d * y + cy);

    const float Jzu = fx * (tij[0] * d - tij[2] * (x * d2));
    const float Jzv = fy * (tij[1] * d - tij[2] * (y * d2));
    oC[p] = wu * Jzu * Jzu + wv * Jzv * Jzv;
    ob[p] = wu * ru * Jzu + wv * rv * Jzv;

    if (stereo) {  // pose weights are zeroed for stereo pairs (:367,432)
      wu = 0.0f;
      wv = 0.0f;
    }
    // raw Jacobians wrt the target pose, [t,w] order (:369-374, 434-439)
    const float Ju[6] = {fx * (h * d), 0.0f, fx * (-x * h * d2), fx * (-x * y * d2), fx * (1.0f + x * x * d2),
                         fx * (-y * d)};
    const float Jv[6] = {0.0f, fy * (h * d), fy * (-y * h * d2), fy * (-1.0f - y * y * d2), fy * (x * y * d2),
                         fy * (x * d)};
    float wJu[6], wJv[6], q[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      wJu[k] = wu * Ju[k];
      wJv[k] = wv * Jv[k];
      q[k] = wJu[k] * Jzu + wJv[k] * Jzv;
      g[k] += wJu[k] * ru + wJv[k] * rv;
    }
    int l = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
      for (int m = k; m < 6; m++) {
        G[l] += wJu[k] * Ju[m] + wJv[k] * Jv[m];
        l++;
      }
    // E rows: q A_i, q A_j
#pragma unroll
    for (int c = 0; c < 6; c++) {
      float ei = 0.0f, ej = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        ei += q[k] * T[ET_AI + k * 6 + c];
        ej += q[k] * T[ET_AJ + k * 6 + c];
      }
      oEi[(long)c * HW + p] = ei;
      oEj[(long)c * HW + p] = ej;
    }
  }

  // ---- block reduction of the 27 partial sums ----
  __shared__ float red[4][27];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int l = 0; l < 21; l++) {
    const float s = wave_sum(G[l]);
    if (lane == 0) red[wave][l] = s;
  }
#pragma unroll
  for (int l = 0; l < 6; l++) {
    const float s = wave_sum(g[l]);
    if (lane == 0) red[wave][21 + l] = s;
  }
  __syncthreads();
  if (!PER_EDGE) {
    // deterministic: the (edge, chunk) partial is written once, summed by the assembly blocks of
    // the Schur launch in chunk order
    if (tid < 27)
      a.partial[((long)e * a.nch + ch) * 32 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    return;
  }
  __shared__ double Gs[27];
  if (tid < 27) Gs[tid] = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
  __syncthreads();
  if (tid < 156) {
    const double val = edge_block_entry(T + ET_AI, T + ET_AJ, Gs, tid);
    if (tid < 144) {
      const int blk = tid / 36, r = (tid % 36) / 6, c = tid % 6;
      a.Hs[(((long)blk * a.M + e) * 6 + r) * 6 + c] = (float)val;
    } else {
      const int side = (tid - 144) / 6, r = (tid - 144) % 6;
      a.vs[((long)side * a.M + e) * 6 + r] = (float)val;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// accum (K6 x3 + the depth block, droid_kernels.cu:1750-1757)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_accum_kernel(const float* __restrict__ Cii, const float* __restrict__ bz,
                                                       const float* __restrict__ Eiz,
                                                       const float* __restrict__ disps,
                                                       const float* __restrict__ disps_sens,
                                                       const float* __restrict__ eta, const int32_t* __restrict__ kx,
                                                       const int32_t* __restrict__ src_ptr,
                                                       const int32_t* __restrict__ src_edge, int HW, int kf0, int P,
                                                       float* __restrict__ Q, float* __restrict__ w,
                                                       float* __restrict__ E) {
  const int k = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  const int fid = kx[k];
  const int t = fid - kf0;
  const bool in_window = (t >= 0 && t < P);
  float C = 0.0f, b = 0.0f, Ei[6] = {0, 0, 0, 0, 0, 0};
  const int s0 = src_ptr[k], s1 = src_ptr[k + 1];
  for (int s = s0; s < s1; s++) {
    const int e = src_edge[s];
    C += Cii[(long)e * HW + p];
    b += bz[(long)e * HW + p];
    if (in_window) {
#pragma unroll
      for (int c = 0; c < 6; c++) Ei[c] += Eiz[((long)e * 6 + c) * HW + p];
    }
  }
  const float alpha = 0.05f;  // (:1750)
  const float ds = disps_sens[(long)fid * HW + p];
  const float m = ds > 0.0f ? 1.0f : 0.0f;
  const float Cf = C + m * alpha + (1.0f - m) * eta[(long)k * HW + p];
  const float wf = b - m * alpha * (disps[(long)fid * HW + p] - ds);
  Q[(long)k * HW + p] = 1.0f / Cf;
  w[(long)k * HW + p] = wf;
  if (in_window) {
#pragma unroll
    for (int c = 0; c < 6; c++) E[((long)t * 6 + c) * HW + p] = Ei[c];
  }
}

// ---------------------------------------------------------------------------------------------
// schur (K9 + K10) + pose-block assembly, one launch:
//   blocks [0, n_pairs*sch)   : pair (n <= m) of E rows sharing a depth slot, pixel chunk b % sch
//   blocks [n_pairs*sch, +M)  : edge e: sum its linearize partials over the chunks (fixed order),
//                               transform with A_i, A_j and add the four 6x6 blocks / two 6-vectors
// Both kinds accumulate into the dense fp64 system with fp64 atomics.
// ---------------------------------------------------------------------------------------------
struct SchurArgs {
  const float* E;
  const float* Q;
  const float* w;
  const int32_t* pairs;
  const int32_t* row_pose;
  const float* partial;
  const float* poses;
  const float* extr;
  const int64_t* ii;
  const int64_t* jj;
  double* Hd;
  double* vd;
  int HW, P, kf0, n_pairs, sch, nch, M;
};

__global__ __launch_bounds__(256) void ba_schur_kernel(SchurArgs a) {
  const int tid = threadIdx.x;
  const int n6 = 6 * a.P;
  if ((int)blockIdx.x >= a.n_pairs * a.sch) {
    // ---------------- edge assembly ----------------
    const int e = blockIdx.x - a.n_pairs * a.sch;
    const int ix = (int)a.ii[e], jx = (int)a.jj[e];
    __shared__ float T[ET_STRIDE];
    __shared__ double Gs[27];
    edge_constants(a.poses, a.extr, ix, jx, T);
    if (tid < 27) {
      double s = 0.0;
      for (int c = 0; c < a.nch; c++) s += (double)a.partial[((long)e * a.nch + c) * 32 + tid];
      Gs[tid] = s;
    }
    __syncthreads();
    if (tid < 156) {
      const double val = edge_block_entry(T + ET_AI, T + ET_AJ, Gs, tid);
      if (tid < 144) {
        const int blk = tid / 36, r = (tid % 36) / 6, c = tid % 6;
        const int rp = ((blk < 2) ? ix : jx) - a.kf0;
        const int cp = ((blk % 2 == 0) ? ix : jx) - a.kf0;
        if (rp >= 0 && rp < a.P && cp >= 0 && cp < a.P) atomicAdd(&a.Hd[(long)(6 * rp + r) * n6 + 6 * cp + c], val);
      } else {
        const int side = (tid - 144) / 6, r = (tid - 144) % 6;
        const int rp = (side == 0 ? ix : jx) - a.kf0;
        if (rp >= 0 && rp < a.P) atomicAdd(&a.vd[6 * rp + r], val);
      }
    }
    return;
  }
  // ---------------- Schur pair ----------------
  const int pid = blockIdx.x / a.sch, chunk = blockIdx.x % a.sch;
  const int n = a.pairs[3 * pid + 0];
  const int m = a.pairs[3 * pid + 1];
  const int k = a.pairs[3 * pid + 2];
  const int HW = a.HW;
  const bool diag = (n == m);
  float S[36];
  float bb[6];
#pragma unroll
  for (int q = 0; q < 36; q++) S[q] = 0.0f;
#pragma unroll
  for (int q = 0; q < 6; q++) bb[q] = 0.0f;
  const float* __restrict__ En = a.E + (long)n * 6 * HW;
  const float* __restrict__ Em = a.E + (long)m * 6 * HW;
  const float* __restrict__ Qk = a.Q + (long)k * HW;
  const float* __restrict__ wk = a.w + (long)k * HW;
  const int csz = (HW + a.sch - 1) / a.sch;
  const int p0 = chunk * csz, p1 = min(HW, p0 + csz);
  for (int p = p0 + tid; p < p1; p += 256) {
    const float q = Qk[p];
    float ei[6], ej[6], en[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
      en[c] = En[(long)c * HW + p];
      ei[c] = en[c] * q;
      ej[c] = Em[(long)c * HW + p];
    }
#pragma unroll
    for (int c = 0; c < 6; c++)
#pragma unroll
      for (int d = 0; d < 6; d++) S[c * 6 + d] += ei[c] * ej[d];
    if (diag) {
      const float qw = q * wk[p];
#pragma unroll
      for (int c = 0; c < 6; c++) bb[c] += qw * en[c];
    }
  }
  __shared__ float red[4][42];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int q = 0; q < 36; q++) {
    const float s = wave_sum(S[q]);
    if (lane == 0) red[wave][q] = s;
  }
  if (diag) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const float s = wave_sum(bb[q]);
      if (lane == 0) red[wave][36 + q] = s;
    }
  }
  __syncthreads();
  const int pn = a.row_pose[n], pm = a.row_pose[m];
  if (tid < 36) {
    const double val = (double)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
    const int c = tid / 6, d = tid % 6;
    atomicAdd(&a.Hd[(long)(6 * pn + c) * n6 + 6 * pm + d], -val);
    if (!diag) atomicAdd(&a.Hd[(long)(6 * pm + d) * n6 + 6 * pn + c], -val);
  } else if (diag && tid < 42) {
    const double val = (double)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
    atomicAdd(&a.vd[6 * pn + (tid - 36)], -val);
  }
}

// fp64 -> fp32, and re-zero the accumulators for the next linearisation (every element is read by
// exactly one thread, so the system buffer needs a memset only once, when it is allocated)
__global__ void ba_finalize_kernel(double* __restrict__ Hd, double* __restrict__ vd, int n6,
                                   float* __restrict__ H, float* __restrict__ v) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n6 * n6) {
    const int r = idx / n6, c = idx - r * n6;
    H[idx] = (float)Hd[(long)c * n6 + r];  // get_dense() hands the column-major data over as row-major (:1305-1316)
    Hd[(long)c * n6 + r] = 0.0;
  }
  if (idx < n6) {
    v[idx] = (float)vd[idx];
    vd[idx] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// solve_depth (K11 + K6 + K8 fused): per (depth slot, pixel)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_solve_depth_kernel(const float* __restrict__ dx, float* __restrict__ disps,
                                                             const float* __restrict__ Q,
                                                             const float* __restrict__ E,
                                                             const float* __restrict__ w,
                                                             const int32_t* __restrict__ kx,
                                                             const int32_t* __restrict__ row_pose,
                                                             const int32_t* __restrict__ slot_rows_ptr,
                                                             const int32_t* __restrict__ slot_rows, int HW, int P,
                                                             float clamp_min) {
  const int k = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  float acc = 0.0f;
  const int r0 = slot_rows_ptr[k], r1 = slot_rows_ptr[k + 1];
  for (int r = r0; r < r1; r++) {
    const int n = slot_rows[r];
    const int pose = row_pose[n];
    if (pose <= 0 || pose >= P) continue;  // EvT6x1_kernel skips these rows (:1225)
    float dw = 0.0f;
#pragma unroll
    for (int c = 0; c < 6; c++) dw += E[((long)n * 6 + c) * HW + p] * dx[pose * 6 + c];
    acc += dw;
  }
  const float dz = Q[(long)k * HW + p] * (w[(long)k * HW + p] - acc);
  const long o = (long)kx[k] * HW + p;
  float d = disps[o] + dz;
  if (clamp_min >= 0.0f) d = fmaxf(d, clamp_min);  // visual_frontend.py:1162
  disps[o] = d;
}

// ---------------------------------------------------------------------------------------------
// pose_retr_kernel (K7, :1015-1048): poses[k] <- Exp([tau,phi]) * poses[k]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void exp_so3_f(const float* phi, float* q) {
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta_p4 = theta_sq * theta_sq;
  const float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

__device__ __forceinline__ void cross_inplace(const float* a, float* b) {
  const float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  b[0] = x;
  b[1] = y;
  b[2] = z;
}

__global__ void pose_retr_kernel(float* __restrict__ poses, const float* __restrict__ dx, int kf0, int kf1) {
  const int k = kf0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kf1) return;
  float* p = poses + (long)k * 7;
  const float* xi = dx + (long)(k - kf0) * 6;
  float dq[4], dt[3];
  exp_so3_f(xi + 3, dq);
  float tau[3] = {xi[0], xi[1], xi[2]};
  const float phi[3] = {xi[3], xi[4], xi[5]};
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta_sq);
  dt[0] = tau[0];
  dt[1] = tau[1];
  dt[2] = tau[2];
  if (theta > 1e-4f) {
    const float a = (1.0f - cosf(theta)) / theta_sq;
    cross_inplace(phi, tau);
    dt[0] += a * tau[0];
    dt[1] += a * tau[1];
    dt[2] += a * tau[2];
    const float b = (theta - sinf(theta)) / (theta * theta_sq);
    cross_inplace(phi, tau);
    dt[0] += b * tau[0];
    dt[1] += b * tau[1];
    dt[2] += b * tau[2];
  }
  const float t[3] = {p[0], p[1], p[2]};
  const float q[4] = {p[3], p[4], p[5], p[6]};
  float q1[4], t1[3];
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  se3::act_so3(dq, t, t1);
  p[0] = t1[0] + dt[0];
  p[1] = t1[1] + dt[1];
  p[2] = t1[2] + dt[2];
  p[3] = q1[0];
  p[4] = q1[1];
  p[5] = q1[2];
  p[6] = q1[3];
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static int max_nch(int HW);

struct WsLayout {
  size_t Hd, vd, partial, Eiz, Cii, bz, total;
};

static WsLayout ws_layout(int M, int P, int HW) {
  WsLayout L;
  size_t off = 0;
  L.Hd = off;
  off += align256(sizeof(double) * (size_t)36 * P * P + 8);
  L.vd = off;
  off += align256(sizeof(double) * (size_t)6 * P + 8);
  L.partial = off;
  off += align256(sizeof(float) * (size_t)32 * (M > 0 ? M : 1) * max_nch(HW));
  L.Eiz = off;
  off += align256(sizeof(float) * (size_t)M * 6 * HW + 4);
  L.Cii = off;
  off += align256(sizeof(float) * (size_t)M * HW + 4);
  L.bz = off;
  off += align256(sizeof(float) * (size_t)M * HW + 4);
  L.total = off;
  return L;
}

extern "C" size_t ns_ba_workspace_bytes(const ns_ba_plan* plan, int HW) {
  if (!plan) return 0;
  return ws_layout(plan->M, plan->P, HW).total;
}

static int max_nch(int HW) { return (HW + 511) / 512; }

static int choose_nch(int M, int HW) {
  // The kernel is latency bound at tracking sizes (a few pixels per lane, dependent loads): use
  // two pixels per lane until ~4096 workgroups are in flight, then grow the chunks instead.
  int nch = max_nch(HW);
  while (nch > 1 && (long)M * nch > 4096) nch = (nch + 1) / 2;
  return nch;
}

static int choose_sch(int n_pairs, int HW) {
  int sch = 2048 / (n_pairs > 0 ? n_pairs : 1);
  const int maxch = (HW + 255) / 256;
  if (sch > maxch) sch = maxch;
  if (sch < 1) sch = 1;
  return sch;
}

// K1 as its own op (per-edge outputs, exactly the reference kernel's contract).
extern "C" int ns_projective_transform(const float* targets, const float* weights, const float* poses,
                                       const float* disps, const float* intrinsics, const float* extrinsics,
                                       const int64_t* ii, const int64_t* jj, int M, int ht, int wd, float* Hs,
                                       float* vs, float* Eiz, float* Ejz, float* Cii, float* bz, float* etab_ws,
                                       void* stream) {
  NS_REQUIRE(targets && weights && poses && disps && intrinsics && extrinsics && ii && jj,
             "ns_projective_transform: null input");
  NS_REQUIRE(Hs && vs && Eiz && Ejz && Cii && bz, "ns_projective_transform: null output");
  NS_REQUIRE(M >= 0 && ht > 0 && wd > 0, "ns_projective_transform: bad shape");
  if (M == 0) return NS_OK;
  hipStream_t st = (hipStream_t)stream;
  (void)etab_ws;  // kept in the signature for ABI stability; the constants now live in LDS
  LinArgs a;
  a.target = targets;
  a.weight = weights;
  a.disps = disps;
  a.intr = intrinsics;
  a.ii = ii;
  a.jj = jj;
  a.poses = poses;
  a.extr = extrinsics;
  a.Eiz = Eiz;
  a.Ejz = Ejz;
  a.Cii = Cii;
  a.bz = bz;
  a.partial = nullptr;
  a.Hs = Hs;
  a.vs = vs;
  a.M = M;
  a.HW = ht * wd;
  a.wd = wd;
  a.nch = 1;
  a.kf0 = 0;
  a.P = 0;
  hipLaunchKernelGGL(ba_linearize_kernel<true>, dim3(M, 1), dim3(256), 0, st, a);
  NS_CHECK_LAUNCH("ba_linearize_kernel<per-edge>");
  return NS_OK;
}

extern "C" int ns_reduced_camera_matrix(const float* poses, const float* disps, const float* intrinsics,
                                        const float* extrinsics, const float* disps_sens, const float* targets,
                                        const float* weights, const float* eta, const int64_t* ii, const int64_t* jj,
                                        const ns_ba_plan* plan, const int32_t* index, const size_t* off, int ht,
                                        int wd, float* H, float* v, float* Q, float* E, float* w, void* workspace,
                                        int ws_zeroed, void* stream) {
  NS_REQUIRE(plan && index && off, "ns_reduced_camera_matrix: null plan");
  NS_REQUIRE(poses && disps && intrinsics && extrinsics && disps_sens && eta, "ns_reduced_camera_matrix: null input");
  NS_REQUIRE(H && v && Q && E && w && workspace, "ns_reduced_camera_matrix: null output/workspace");
  NS_REQUIRE(plan->M == 0 || (targets && weights && ii && jj), "ns_reduced_camera_matrix: null edge data");
  NS_REQUIRE(ht > 0 && wd > 0 && plan->P >= 0, "ns_reduced_camera_matrix: bad shape");
  NS_REQUIRE(((uintptr_t)workspace & 255) == 0, "ns_reduced_camera_matrix: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int M = plan->M, P = plan->P, K = plan->K, HW = ht * wd, n6 = 6 * P;
  const WsLayout L = ws_layout(M, P, HW);
  char* ws = (char*)workspace;
  double* Hd = (double*)(ws + L.Hd);
  double* vd = (double*)(ws + L.vd);
  float* Eiz = (float*)(ws + L.Eiz);
  float* Cii = (float*)(ws + L.Cii);
  float* bz = (float*)(ws + L.bz);
  float* partial = (float*)(ws + L.partial);
  // Hd and vd are adjacent in the layout: one memset, needed only the first time a workspace is used
  // (ba_finalize_kernel re-zeroes what it reads)
  if (!ws_zeroed && hipMemsetAsync(Hd, 0, L.partial - L.Hd, st) != hipSuccess) {
    ns_set_error("ns_reduced_camera_matrix: hipMemsetAsync failed");
    return NS_ELAUNCH;
  }
  int nch = 1;
  if (M > 0) {
    LinArgs a;
    a.target = targets;
    a.weight = weights;
    a.disps = disps;
    a.intr = intrinsics;
    a.ii = ii;
    a.jj = jj;
    a.poses = poses;
    a.extr = extrinsics;
    a.Eiz = Eiz;
    a.Ejz = E + (long)P * 6 * HW;
    a.Cii = Cii;
    a.bz = bz;
    a.partial = partial;
    a.Hs = nullptr;
    a.vs = nullptr;
    a.M = M;
    a.HW = HW;
    a.wd = wd;
    a.nch = nch = choose_nch(M, HW);
    a.kf0 = plan->kf0;
    a.P = P;
    hipLaunchKernelGGL(ba_linearize_kernel<false>, dim3(M, a.nch), dim3(256), 0, st, a);
    NS_CHECK_LAUNCH("ba_linearize_kernel");
  }
  const int32_t* kx = index + off[0];
  const int32_t* row_pose = index + off[2];
  const int32_t* src_ptr = index + off[3];
  const int32_t* src_edge = index + off[4];
  const int32_t* pairs = index + off[5];
  if (K > 0) {
    hipLaunchKernelGGL(ba_accum_kernel, dim3(K, ns_cdiv(HW, 256)), dim3(256), 0, st, Cii, bz, Eiz, disps, disps_sens,
                       eta, kx, src_ptr, src_edge, HW, plan->kf0, P, Q, w, E);
    NS_CHECK_LAUNCH("ba_accum_kernel");
  }
  if (plan->n_pairs + M > 0) {
    SchurArgs sa;
    sa.E = E;
    sa.Q = Q;
    sa.w = w;
    sa.pairs = pairs;
    sa.row_pose = row_pose;
    sa.partial = partial;
    sa.poses = poses;
    sa.extr = extrinsics;
    sa.ii = ii;
    sa.jj = jj;
    sa.Hd = Hd;
    sa.vd = vd;
    sa.HW = HW;
    sa.P = P;
    sa.kf0 = plan->kf0;
    sa.n_pairs = plan->n_pairs;
    sa.sch = choose_sch(plan->n_pairs, HW);
    sa.nch = nch;
    sa.M = M;
    hipLaunchKernelGGL(ba_schur_kernel, dim3(plan->n_pairs * sa.sch + M), dim3(256), 0, st, sa);
    NS_CHECK_LAUNCH("ba_schur_kernel");
  }
  if (n6 > 0) {
    hipLaunchKernelGGL(ba_finalize_kernel, dim3(ns_cdiv((long)n6 * n6, 256)), dim3(256), 0, st, Hd, vd, n6, H, v);
    NS_CHECK_LAUNCH("ba_finalize_kernel");
  }
  return NS_OK;
}

extern "C" int ns_solve_depth(const float* dx, float* disps, const float* Q, const float* E, const float* w,
                              const ns_ba_plan* plan, const int32_t* index, const size_t* off, int ht, int wd,
                              float clamp_min, void* stream) {
  NS_REQUIRE(plan && index && off, "ns_solve_depth: null plan");
  NS_REQUIRE(dx && disps && Q && E && w, "ns_solve_depth: null pointer");
  if (plan->K == 0) return NS_OK;
  const int HW = ht * wd;
  hipLaunchKernelGGL(ba_solve_depth_kernel, dim3(plan->K, ns_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, dx,
                     disps, Q, E, w, index + off[0], index + off[2], index + off[6], index + off[7], HW, plan->P,
                     clamp_min);
  NS_CHECK_LAUNCH("ba_solve_depth_kernel");
  return NS_OK;
}

extern "C" int ns_pose_retr(float* poses, const float* dx, int kf0, int kf1, void* stream) {
  NS_REQUIRE(poses && dx, "ns_pose_retr: null pointer");
  if (kf1 <= kf0) return NS_OK;
  hipLaunchKernelGGL(pose_retr_kernel, dim3(ns_cdiv(kf1 - kf0, 64)), dim3(64), 0, (hipStream_t)stream, poses, dx, kf0,
                     kf1);
  NS_CHECK_LAUNCH("pose_retr_kernel");
  return NS_OK;
}
