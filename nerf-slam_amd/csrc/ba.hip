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
// Structure of one linearisation (5 launches, 0 host syncs; round 6 -- SURVEY 7 steps 4-5):
//   edge_table  per edge: G_ij and the two 6x6 maps A_i, A_j with J_i = J_raw A_i, J_j = J_raw A_j (the reference applies them
//               per pixel, :376-403; they are linear)
//   linearize_slot  grid (depth slot, pixel chunk): ALL edges of one source frame in one workgroup, each lane owning a
//               few pixels.  Per edge the constants G_ij and the two 6x6 maps A_i, A_j with J_i = J_raw A_i,
//               J_j = J_raw A_j (the reference applies them per pixel, :376-403; they are linear) are staged in LDS, 16
//               edges at a time.  Per (edge, pixel): residual / weights / J_raw, the Ejz row is written, and
//               C, b, Eiz are summed over the slot's edges IN REGISTERS (the reference's three accum_cuda round trips,
//               :1065-1115, :1750-1757; rounds 1-5 here: a separate kernel re-reading 1.8 GB of per-edge C / b / Eiz at
//               config #5) -> Q = 1 / (C + prior or damping), w, and the slot's own E row.  Only G = sum w J_raw^T J_raw (21)
//               and g = sum w r J_raw (6) are accumulated per edge instead of the reference's 78+12 (Hii = A_i^T G A_i,
//               Hij = A_i^T G A_j, ...), reduced per wave with DPP adds and written as (edge, chunk, wave) partials
//   schur_gram  the Schur complement of one depth slot is ONE Gram matrix: its window rows X (6 values each, [6 rows, HW])
//               give S = X diag(Q) X^T and s = X (Q o w).  A workgroup takes (slot, block of <= 8 x 8 tiles of 16 values, pixel
//               split), streams the rows ONCE (float4 per lane, two steps ahead in registers) and forms the tile products
//               on the matrix cores with v_mfma_f32_16x16x4_f32 -- exact f32, a k-ordered fmaf chain -- 36 accumulator
//               tiles per wave; the pixel splits write partials.  Rounds 1-5: one
//               workgroup per ROW PAIR (the reference's EEt6x6 structure, :1118-1173), 37 k pairs re-reading 13 planes each
//               = 18x the operand set, 2.28 ms at config #5.
//   schur_reduce  per job: the splits' partials summed in fixed order (f64) and subtracted from the dense fp64 system (fp64
//               atomics); per edge: the linearize partials summed (fixed order), transformed with A_i, A_j and the four 6x6
//               blocks / two 6-vectors added (the reference: 90 serial block reductions, then a D2H copy and Eigen triplets)
//   finalize    fp64 -> fp32 H, v  (transposed like SparseBlock::get_dense, :1305-1316)
#include "common.h"
#include "se3.h"

#include <type_traits>

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

// Per-edge constants in LDS:
//   Ts[0..2] t_ij, Ts[3..6] q_ij, Ts[7] stereo flag, Ts[8..43] A_i, Ts[44..79] A_j   (row-major 6x6)
// edge_rel: one lane per edge; edge_maps: six lanes per edge (k = column of the identity pushed through the adjoints).
__device__ __forceinline__ void edge_rel(const float* __restrict__ poses, int ix, int jx, float* Ts) {
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
    se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij,
                 qij);
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

__device__ __forceinline__ void edge_maps(const float* __restrict__ extr, int k, float* Ts) {
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

// one edge per workgroup: the first lanes compute its constants
__device__ __forceinline__ void edge_constants(const float* __restrict__ poses, const float* __restrict__ extr,
                                               int ix, int jx, float* Ts) {
  const int tid = threadIdx.x;
  if (tid == 0) edge_rel(poses, ix, jx, Ts);
  __syncthreads();
  if (tid < 6) edge_maps(extr, tid, Ts);
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
// edge table: the per-edge constants of one linearisation, [M][ET_STRIDE] floats, one launch of M x 8 lanes (lane k < 6 of an
// edge computes column k of both maps; every lane repeats the cheap relative pose instead of waiting for a neighbour).
// Rounds 1-5 computed them in the first lanes of EVERY workgroup of an edge, behind two barriers; with a workgroup per
// (depth slot, pixel chunk) that staging was ~40 % of a workgroup's time at config #5.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_edge_table_kernel(const float* __restrict__ poses, const float* __restrict__ extr,
                                                           const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                           int M, float* __restrict__ table) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int e = idx >> 3, k = idx & 7;
  if (e >= M || k >= 6) return;
  float Ts[ET_STRIDE];
  edge_rel(poses, (int)ii[e], (int)jj[e], Ts);
  edge_maps(extr, k, Ts);
  float* __restrict__ o = table + (long)e * ET_STRIDE;
  if (k == 0) {
#pragma unroll
    for (int n = 0; n < 8; n++) o[n] = Ts[n];
  }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    o[ET_AI + k * 6 + c] = Ts[ET_AI + k * 6 + c];
    o[ET_AJ + k * 6 + c] = Ts[ET_AJ + k * 6 + c];
  }
}

// ---------------------------------------------------------------------------------------------
// linearize + accumulate in one launch (K1 + K6 x3 + the depth block, droid_kernels.cu:192-536, 971-991, 1750-1757):
// grid (depth slot, pixel chunk); a lane owns PPL pixels of the chunk for ALL edges of the slot's source frame.
// ---------------------------------------------------------------------------------------------

typedef float ls_f2 __attribute__((ext_vector_type(2)));

// packed f32 with ONE half of the second operand broadcast to both lanes (VOP3P op_sel): a * b[H] (+ c).  The compiler forms
// such operands by copying the scalar into a fresh register pair (two v_mov per use: 85 moves per (edge, pixel) in the
// lineariser); the instruction can select the half itself.
template <int H>
__device__ __forceinline__ ls_f2 pk_fma_b(ls_f2 a, ls_f2 b, ls_f2 c) {
  ls_f2 d;
  if (H == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int H>
__device__ __forceinline__ ls_f2 pk_mul_b(ls_f2 a, ls_f2 b) {
  ls_f2 d;
  if (H == 0)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

struct LinSlotArgs {
  const float* target;      // [M,2,HW]
  const float* weight;      // [M,2,HW]
  const float* disps;       // [*,HW]
  const float* disps_sens;  // [*,HW]
  const float* eta;         // [K,HW]
  const float* intr;        // [4]
  const float* poses;       // [*,7]
  const float* extr;        // [7]
  const float* table;       // [M,ET_STRIDE] edge constants (ba_edge_table_kernel)
  const int32_t* kx;        // [K] source frame of the slot
  const int32_t* src_ptr;   // [K+1] CSR over the slot's edges
  const int32_t* src_edge;  // [M]
  float* E;                 // [P+M,6,HW]: rows [0,P) the slots' own rows, rows P+e the edges' Ejz
  float* Q;                 // [K,HW]
  float* w;                 // [K,HW]
  float* partial;           // [M, nch, 32]: per (edge, chunk) sums G(21), g(6)
  int M, HW, wd, nch, kf0, P;
};

template <int PPL>
__global__ __launch_bounds__(256, PPL == 1 ? 3 : 2) void ba_linearize_slot_kernel(LinSlotArgs a) {
  // A workgroup owns 64 * PPL pixels of one depth slot; its four WAVES take the slot's edges in turn (edge q of a batch goes to
  // wave q % 4), every lane owning the same PPL pixels for all of them.  So the per-edge sums G, g are complete after one wave
  // reduction, four edges are in flight per workgroup (a tracking window's ~10 edges per slot: three serial edge bodies instead
  // of ten), and the slot's C, b, Eiz meet once at the end, through LDS, in wave order.
  constexpr int NPX = 64 * PPL;
  const int k = blockIdx.x;
  const int ch = blockIdx.y;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int HW = a.HW;
  const int fid = a.kx[k];
  const int t = fid - a.kf0;
  const bool in_window = (t >= 0 && t < a.P);
  __shared__ __attribute__((aligned(16))) float T[4][2][ET_STRIDE + 16];  // per wave: the constants of its current and next edge
  __shared__ float red[4][8][NPX];

  const float fx = a.intr[0], fy = a.intr[1], cx = a.intr[2], cy = a.intr[3];
  const float* __restrict__ disp = a.disps + (long)fid * HW;

  unsigned px[PPL], pxb[PPL];   // pixel and its BYTE offset (unsigned: a uniform plane pointer + a zero-extended 32-bit lane
  bool ok[PPL];                 //  offset is ONE address register per access)
  float X0[PPL], X1[PPL], dsp[PPL];
  float C[PPL], b[PPL], Ei[PPL][6];
#pragma unroll
  for (int i = 0; i < PPL; i++) {
    const int p = ch * NPX + i * 64 + lane;
    ok[i] = p < HW;
    px[i] = (unsigned)(ok[i] ? p : HW - 1);
    pxb[i] = px[i] * 4u;
    const int row = (int)px[i] / a.wd, col = (int)px[i] - row * a.wd;
    X0[i] = ((float)col - cx) / fx;
    X1[i] = ((float)row - cy) / fy;
    dsp[i] = disp[px[i]];
    C[i] = 0.0f;
    b[i] = 0.0f;
#pragma unroll
    for (int c = 0; c < 6; c++) Ei[i][c] = 0.0f;
  }
  const int s0 = a.src_ptr[k], s1 = a.src_ptr[k + 1];
  {
    const int nb = s1 - s0;
    // The four input planes AND the constants of this wave's next edge are requested before the current one is computed (PPL x 4
    // more requests per lane in flight: with two waves per SIMD the loads of ONE edge do not cover the ~300 instructions per
    // pixel that follow them).  No workgroup barrier inside the edge loop: a wave's constants live in its own LDS rows.
    float ntu[PPL], ntv[PPL], nwu[PPL], nwv[PPL], ntab[2];
    int ne;
    auto request = [&](int eid) __attribute__((always_inline)) {
      ne = eid;
      const long e = eid;
      const float* __restrict__ tu = a.target + (e * 2 + 0) * HW;
      const float* __restrict__ tv = a.target + (e * 2 + 1) * HW;
      const float* __restrict__ wu_ = a.weight + (e * 2 + 0) * HW;
      const float* __restrict__ wv_ = a.weight + (e * 2 + 1) * HW;
      const float* __restrict__ tb = a.table + e * ET_STRIDE;
      ntab[0] = tb[lane];
      ntab[1] = tb[lane < ET_STRIDE - 64 ? 64 + lane : 0];
#pragma unroll
      for (int i = 0; i < PPL; i++) {
        ntu[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tu) + pxb[i]);
        ntv[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tv) + pxb[i]);
        nwu[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(wu_) + pxb[i]);
        nwv[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(wv_) + pxb[i]);
      }
    };
    // edge ids run TWO edges ahead of the arithmetic: the planes of edge q+4 are requested with an id that was loaded while edge
    // q-4 was computed (an id loaded inside `request` put a dependent round trip -- id, then planes -- in front of every edge
    // body: the waves sat parked 49 % of their cycles)
    int id_next = wave < nb ? a.src_edge[s0 + wave] : 0;
    int id_next2 = a.src_edge[s0 + min(wave + 4, nb > 0 ? nb - 1 : 0)];
    if (wave < nb) request(id_next);
    int flip = 0;
    for (int q = wave; q < nb; q += 4, flip ^= 1) {
      // this wave's LDS row: [0,8) t_ij, q_ij, stereo flag; then the 36 entries of the two maps INTERLEAVED, (A_i[m][c], A_j[m][c])
      // at 8 + 2 (6 m + c): one packed multiply-add per entry forms both E rows
      float* __restrict__ Tw = T[wave][flip];
      {
        const int v0 = lane, v1 = 64 + lane;
        Tw[v0 < 8 ? v0 : (v0 < ET_AJ ? 8 + 2 * (v0 - ET_AI) : 9 + 2 * (v0 - ET_AJ))] = ntab[0];
        if (v1 < ET_STRIDE) Tw[9 + 2 * (v1 - ET_AJ)] = ntab[1];
      }
      const int e = ne;
      float ltu[PPL], ltv[PPL], lwu[PPL], lwv[PPL];
#pragma unroll
      for (int i = 0; i < PPL; i++) {
        ltu[i] = ntu[i];
        ltv[i] = ntv[i];
        lwu[i] = nwu[i];
        lwv[i] = nwv[i];
      }
      request(q + 4 < nb ? id_next2 : e);  // (always issued: no branch between a load and its use; the last edge is read twice)
      id_next2 = a.src_edge[s0 + min(q + 8, nb - 1)];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS row is written (wave-private: no barrier)
      const float* __restrict__ Tq = Tw;
      const float tij[3] = {Tq[ET_T], Tq[ET_T + 1], Tq[ET_T + 2]};
      const float qij[4] = {Tq[ET_Q], Tq[ET_Q + 1], Tq[ET_Q + 2], Tq[ET_Q + 3]};
      const bool stereo = Tq[ET_STEREO] != 0.0f;
      float* __restrict__ oEj = a.E + ((long)a.P + e) * 6 * HW;
      // The per-pixel arithmetic in PACKED f32 (v_pk_mul_f32 / v_pk_fma_f32: two lanes of a 64-bit register pair per instruction).
      // SQ counters of the scalar form: 410 vector instructions per (edge, pixel), the SIMDs issuing 98 % of the time -- the
      // kernel was instruction-bound at 2.7 TB/s.  Pairs run along the Jacobian index (J[2k], J[2k+1]); the 21 upper-triangle
      // sums of G are kept as 12 aligned pairs (an odd row starts one entry early: three redundant lower-triangle entries
      // cost no instruction), the two E rows as pairs (E_i[c], E_j[c]).
      ls_f2 G2[12];
      ls_f2 g2[3];
#pragma unroll
      for (int l = 0; l < 12; l++) G2[l] = ls_f2{0.0f, 0.0f};
#pragma unroll
      for (int l = 0; l < 3; l++) g2[l] = ls_f2{0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < PPL; i++) {
        // keep the 72 entries of A_i/A_j in LDS (broadcast reads) instead of letting the compiler hoist
        // them into 72 VGPRs per lane
        asm volatile("" ::: "memory");
        if (ok[i]) {  // ONE divergent region per pixel (only the last chunk has lanes outside the map)
          float Xi[4], Xj[4];
          Xi[0] = X0[i];
          Xi[1] = X1[i];
          Xi[2] = 1.0f;
          Xi[3] = dsp[i];
          se3::act_se3(tij, qij, Xi, Xj);
          const float x = Xj[0], y = Xj[1], h = Xj[3];
          const bool zok = !(Xj[2] < NS_MIN_DEPTH);
          // 1 / Z: reciprocal + one Newton step (correctly rounded but for rare last-bit cases; the IEEE division sequence is ten
          // instructions)
          float rz = __builtin_amdgcn_rcpf(Xj[2]);
          rz = fmaf(rz, fmaf(-Xj[2], rz, 1.0f), rz);
          const float d = zok ? rz : 0.0f;
          const float d2 = d * d;
          // `.001 * weight` is a double product in the reference (:344-345): float(0.001 w) from the two-term split of the double
          // constant -- one multiply and one fused multiply-add instead of two conversions and a double multiply; the results
          // agree except where the double product lies within 2^-48 of a rounding boundary
          constexpr float C_HI = 0.001f;
          constexpr float C_LO = (float)(0.001 - (double)C_HI);
          float wu = zok ? fmaf(lwu[i], C_HI, lwu[i] * C_LO) : 0.0f;
          float wv = zok ? fmaf(lwv[i], C_HI, lwv[i] * C_LO) : 0.0f;
          const float ru = ltu[i] - (fx * d * x + cx);
          const float rv = ltv[i] - (fy * d * y + cy);
          const float Jzu = fx * (tij[0] * d - tij[2] * (x * d2));
          const float Jzv = fy * (tij[1] * d - tij[2] * (y * d2));
          const float ce = wu * Jzu * Jzu + wv * Jzv * Jzv;
          const float be = wu * ru * Jzu + wv * rv * Jzv;
          C[i] += ce;
          b[i] += be;
          if (stereo) {  // pose weights are zeroed for stereo pairs (:367,432)
            wu = 0.0f;
            wv = 0.0f;
          }
          // raw Jacobians wrt the target pose, [t,w] order (:369-374, 434-439), as pairs (J[2k], J[2k+1])
          const ls_f2 Ju2[3] = {ls_f2{fx * (h * d), 0.0f}, ls_f2{fx * (-x * h * d2), fx * (-x * y * d2)},
                                ls_f2{fx * (1.0f + x * x * d2), fx * (-y * d)}};
          const ls_f2 Jv2[3] = {ls_f2{0.0f, fy * (h * d)}, ls_f2{fy * (-y * h * d2), fy * (-1.0f - y * y * d2)},
                                ls_f2{fy * (x * y * d2), fy * (x * d)}};
          const ls_f2 w2 = ls_f2{wu, wv}, Jz2 = ls_f2{Jzu, Jzv}, r2 = ls_f2{ru, rv};
          ls_f2 wJu2[3], wJv2[3], qq2[3];
#pragma unroll
          for (int k2 = 0; k2 < 3; k2++) {
            wJu2[k2] = pk_mul_b<0>(Ju2[k2], w2);
            wJv2[k2] = pk_mul_b<1>(Jv2[k2], w2);
            qq2[k2] = pk_fma_b<1>(wJv2[k2], Jz2, pk_mul_b<0>(wJu2[k2], Jz2));
            g2[k2] = pk_fma_b<1>(wJv2[k2], r2, pk_fma_b<0>(wJu2[k2], r2, g2[k2]));
          }
          // G: row m, column pairs from the (even) pair that holds column m
          {
            int l = 0;
#pragma unroll
            for (int m = 0; m < 6; m++) {
#pragma unroll
              for (int k2 = m >> 1; k2 < 3; k2++) {
                if (m & 1)
                  G2[l] = pk_fma_b<1>(Jv2[k2], wJv2[m >> 1], pk_fma_b<1>(Ju2[k2], wJu2[m >> 1], G2[l]));
                else
                  G2[l] = pk_fma_b<0>(Jv2[k2], wJv2[m >> 1], pk_fma_b<0>(Ju2[k2], wJu2[m >> 1], G2[l]));
                l++;
              }
            }
          }
          // E rows: q A_i (summed over the slot's edges), q A_j (written), one packed multiply-add per map entry
          const ls_f2* __restrict__ A2 = reinterpret_cast<const ls_f2*>(Tq + 8);
#pragma unroll
          for (int c = 0; c < 6; c++) {
            ls_f2 e2 = ls_f2{0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 6; m++) {
              if (m & 1)
                e2 = pk_fma_b<1>(A2[m * 6 + c], qq2[m >> 1], e2);
              else
                e2 = pk_fma_b<0>(A2[m * 6 + c], qq2[m >> 1], e2);
            }
            Ei[i][c] += e2[0];
            // (uniform plane pointer + this lane's 32-bit BYTE offset: one address register, no 64-bit add per access)
            *reinterpret_cast<float*>(reinterpret_cast<char*>(oEj + (long)c * HW) + pxb[i]) = e2[1];
          }
        }
      }
      // the wave's sums of G, g ARE the (edge, chunk) sums (fixed DPP order), gathered into lanes 0..26 and stored once; the
      // assembly blocks of the Schur-reduce launch add the chunks in order
      float mine = 0.0f;
      {
        int l = 0, lp = 0;
#pragma unroll
        for (int m = 0; m < 6; m++)
#pragma unroll
          for (int k2 = m >> 1; k2 < 3; k2++) {
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
              const int n = 2 * k2 + hh;
              if (n >= m) {   // (compile time: the entry before an odd row's diagonal is the redundant one)
                const float sum = wave_sum(G2[lp][hh]);
                mine = (lane == l) ? sum : mine;
                l++;
              }
            }
            lp++;
          }
      }
#pragma unroll
      for (int l = 0; l < 6; l++) {
        const float sum = wave_sum(g2[l >> 1][l & 1]);
        mine = (lane == 21 + l) ? sum : mine;
      }
      if (lane < 27) a.partial[((long)e * a.nch + ch) * 32 + lane] = mine;
    }
  }
  // the four waves' sums over their edges meet in LDS; depth block of the slot (:1750-1754) and its own E row (:1757)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PPL; i++) {
    red[wave][0][i * 64 + lane] = C[i];
    red[wave][1][i * 64 + lane] = b[i];
#pragma unroll
    for (int c = 0; c < 6; c++) red[wave][2 + c][i * 64 + lane] = Ei[i][c];
  }
  __syncthreads();
  if (tid < NPX) {
    const int p = ch * NPX + tid;
    if (p < HW) {
      float sum[8];
#pragma unroll
      for (int v = 0; v < 8; v++) sum[v] = ((red[0][v][tid] + red[1][v][tid]) + red[2][v][tid]) + red[3][v][tid];
      const float alpha = 0.05f;
      const float ds = (a.disps_sens + (long)fid * HW)[p];
      const float m = ds > 0.0f ? 1.0f : 0.0f;
      const float Cf = sum[0] + m * alpha + (1.0f - m) * (a.eta + (long)k * HW)[p];
      const float wf = sum[1] - m * alpha * (disp[p] - ds);
      (a.Q + (long)k * HW)[p] = 1.0f / Cf;
      (a.w + (long)k * HW)[p] = wf;
      if (in_window) {
#pragma unroll
        for (int c = 0; c < 6; c++) (a.E + ((long)t * 6 + c) * HW)[p] = sum[2 + c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Schur complement (K9 + K10, droid_kernels.cu:1118-1210, 1349-1438) as one Gram matrix per depth slot, on the matrix
// cores in exact f32, + the pose-block assembly of the linearisation, one launch:
//   blocks [0, n_jobs*S)      : job (slot, A tiles, B tiles) x pixel split s
//   blocks [n_jobs*S, +M)     : edge e: sum its linearize partials (fixed order), transform with A_i, A_j, add to the system
// ---------------------------------------------------------------------------------------------
typedef float gr_f4 __attribute__((ext_vector_type(4)));
typedef float gr_f4u __attribute__((ext_vector_type(4), aligned(4)));  // a row of E is only 4-byte aligned when HW % 4 != 0

#define GR_NACC 36                      // accumulator tiles of a job: 8x8 upper triangle, or 4 x 8
#define GR_PART (GR_NACC * 256 + 128)   // floats of one (job, split) partial: the tiles + 8 x 16 entries of X (Q o w)
#define GR_ROUND 12                     // tiles per LDS round of the cross-wave sum
#define GR_RG 4                         // tiles per workgroup of the reduce launch
#define GR_NRG (GR_NACC / GR_RG)

struct GramArgs {
  const float* E;
  const float* Q;
  const float* w;
  const float* zrow;  // HW zeros: what the padding rows of a slot's last tile read
  const int32_t* jobs;       // [n_jobs][NS_GRAM_JOB_INTS]
  const int32_t* job_plane;  // [n_jobs][2][128]
  const int32_t* job_hrow;   // [n_jobs][2][128]
  float* part;   // [n_jobs * S][GR_PART]
  double* Hd;
  double* vd;
  // edge assembly
  const float* partial;
  const float* table;   // [M,ET_STRIDE] edge constants
  const int64_t* ii;
  const int64_t* jj;
  int HW, P, kf0, n_jobs, S, npart, M;
};

// index of tile pair (ta <= tb) in the upper triangle of an 8 x 8 block
__device__ __forceinline__ constexpr int gr_pair(int ta, int tb) { return ta * 8 - ta * (ta - 1) / 2 + (tb - ta); }

template <bool DIAG>
__device__ __forceinline__ void gram_job(const GramArgs& a, const int job, const int split, float* lds, float* lds_v) {
  constexpr int NA = DIAG ? 8 : 4;  // A tiles a job may have
  constexpr int NB = DIAG ? 0 : 8;  // separately loaded B tiles (a diagonal job's B operand is Q o its A tiles)
  constexpr int NBX = DIAG ? 1 : 8;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, kg = lane >> 4;
  const int HW = a.HW;
  const int32_t* __restrict__ hdr = a.jobs + (long)job * NS_GRAM_JOB_INTS;
  const int slot = hdr[0];
  const int na = __builtin_amdgcn_readfirstlane(hdr[2]);
  const int nb = __builtin_amdgcn_readfirstlane(hdr[4]);

  // this lane's row of every tile: plane of E of value tile*16 + r16 (from the plan), the zero row for padding
  const int32_t* __restrict__ jp = a.job_plane + (long)job * 256;
  const float* pa[NA];
  const float* pb[NBX];
#pragma unroll
  for (int t = 0; t < NA; t++) {
    const int pl = jp[t * 16 + r16];
    pa[t] = pl >= 0 ? a.E + (long)pl * HW : a.zrow;
  }
#pragma unroll
  for (int t = 0; t < NBX; t++) {
    const int pl = DIAG ? -1 : jp[128 + t * 16 + r16];
    pb[t] = pl >= 0 ? a.E + (long)pl * HW : a.zrow;
  }
  const float* __restrict__ Qk = a.Q + (long)slot * HW;
  const float* __restrict__ wk = a.w + (long)slot * HW;

  // pixel range of the split, in rounds of 64 pixels (16 per wave)
  const int NR = (HW + 63) >> 6;
  const int r0 = (int)((long)NR * split / a.S), r1 = (int)((long)NR * (split + 1) / a.S);
  const int n = r1 - r0;

  gr_f4 acc[GR_NACC];
#pragma unroll
  for (int i = 0; i < GR_NACC; i++) acc[i] = gr_f4{0.0f, 0.0f, 0.0f, 0.0f};
  float vs[8];
#pragma unroll
  for (int t = 0; t < 8; t++) vs[t] = 0.0f;

  // three register stages: the loads of rounds i+1 and i+2 are in flight while round i is on the matrix core (one wave per
  // SIMD, 512 registers: latency is covered inside the wave).  A stage's loads are ALWAYS issued -- past the range they repeat the
  // last round -- so that no branch sits between a load and its use and the compiler keeps counting vmcnt.
  gr_f4 sx[3][NA], sb[3][NBX], sq[3], sw[3];
  auto load = [&](auto stage_c, int i) __attribute__((always_inline)) {
    constexpr int st = decltype(stage_c)::value;
    const int rr = r0 + min(i, n - 1);
    const int p = rr * 64 + wave * 16 + kg * 4;
    const int pl = min(p, HW - 4);
    gr_f4 q = *reinterpret_cast<const gr_f4u*>(Qk + pl);
    // element j of the lane's float4 is pixel pl + j: it is this lane's iff pl + j >= p (pl < p only at the end of a row
    // whose length is no multiple of 4, or past it; then the earlier pixels belong to the neighbour lane / nobody).  A round
    // past the split's range (the loop runs in threes) contributes nothing either: Q = 0 there.
    const int first = i < n ? p : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = (pl + j >= first) ? q[j] : 0.0f;
    sq[st] = q;
    sw[st] = *reinterpret_cast<const gr_f4u*>(wk + pl);
#pragma unroll
    for (int t = 0; t < NA; t++) sx[st][t] = *reinterpret_cast<const gr_f4u*>(pa[t] + pl);
#pragma unroll
    for (int t = 0; t < NB; t++) sb[st][t] = *reinterpret_cast<const gr_f4u*>(pb[t] + pl);
  };
  auto compute = [&](auto stage_c) __attribute__((always_inline)) {
    constexpr int st = decltype(stage_c)::value;
    const gr_f4 q = sq[st];
    gr_f4 y[8];
#pragma unroll
    for (int t = 0; t < 8; t++) y[t] = (DIAG ? sx[st][t < NA ? t : 0] : sb[st][t < NBX ? t : 0]) * q;
    // ONE uniform branch per B tile, the MFMAs of a column back to back behind it (a guard per MFMA -- `break`s in the unrolled
    // loops -- put an s_cbranch between every two matrix instructions and the conditions into spilled SGPRs: 1.40 ms at config #5).
    // An off-diagonal job's absent A tiles read the zero row: up to 3 wasted MFMAs per column instead of a branch each.
#pragma unroll
    for (int tb = 0; tb < 8; tb++) {
      if (tb < nb) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if constexpr (DIAG) {
#pragma unroll
            for (int ta = 0; ta <= tb; ta++)
              acc[gr_pair(ta, tb)] = __builtin_amdgcn_mfma_f32_16x16x4f32(sx[st][ta][j], y[tb][j], acc[gr_pair(ta, tb)], 0, 0, 0);
          } else {
#pragma unroll
            for (int ta = 0; ta < 4; ta++)
              acc[ta * 8 + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sx[st][ta][j], y[tb][j], acc[ta * 8 + tb], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (DIAG) {  // s = X (Q o w) of the block's own rows
      const gr_f4 wv = sw[st];
#pragma unroll
      for (int t = 0; t < 8; t++)
        vs[t] += (y[t][0] * wv[0] + y[t][1] * wv[1]) + (y[t][2] * wv[2] + y[t][3] * wv[3]);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  if (n > 0) {
    // straight-line body, three rounds per trip (rounds past the range run on Q = 0): with a branch around any of the stages
    // the compiler's load counter gave up and the loop header waited for vmcnt(0) -- no prefetch at all
    load(S0{}, 0);
    load(S1{}, 1);
    for (int i = 0; i < n; i += 3) {
      load(S2{}, i + 2);
      compute(S0{});
      load(S0{}, i + 3);
      compute(S1{});
      load(S1{}, i + 4);
      compute(S2{});
    }
  }

  // ---- the four waves' tiles summed through LDS (fixed order), GR_ROUND tiles at a time -> this split's partial ----
  float* __restrict__ part = a.part + ((long)job * a.S + split) * GR_PART;
  auto tile_live = [&](int tile) -> bool {  // (uniform) does the job compute accumulator tile `tile`
    if (DIAG) {
      int ta = 0, rem = tile;  // invert gr_pair
      while (rem >= 8 - ta) {
        rem -= 8 - ta;
        ta++;
      }
      return ta + rem < nb;
    }
    return tile < 32 && (tile >> 3) < na && (tile & 7) < nb;
  };
#pragma unroll
  for (int rd = 0; rd < GR_NACC / GR_ROUND; rd++) {
#pragma unroll
    for (int i = 0; i < GR_ROUND; i++)
      if (tile_live(rd * GR_ROUND + i))
        *reinterpret_cast<gr_f4*>(lds + ((wave * GR_ROUND + i) * 64 + lane) * 4) = acc[rd * GR_ROUND + i];
    if (DIAG && rd == 0) {
#pragma unroll
      for (int t = 0; t < 8; t++) lds_v[(wave * 8 + t) * 64 + lane] = vs[t];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GR_ROUND; i++)
      if (tile_live(rd * GR_ROUND + i)) {
        const float* __restrict__ l0 = lds + i * 256 + tid;
        part[(rd * GR_ROUND + i) * 256 + tid] =
            ((l0[0] + l0[GR_ROUND * 256]) + l0[2 * GR_ROUND * 256]) + l0[3 * GR_ROUND * 256];
      }
    if (DIAG && rd == 0 && tid < 128) {
      const int t = tid >> 4, r = tid & 15;
      float sum = 0.0f;
#pragma unroll
      for (int wv = 0; wv < 4; wv++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) sum += lds_v[(wv * 8 + t) * 64 + g4 * 16 + r];
      part[GR_NACC * 256 + tid] = sum;
    }
    __syncthreads();
  }
}

// A job's block of the system: the S pixel-split partials summed in fixed order (f64), mapped from (tile, element) to the rows /
// columns of H through the slot's window rows, subtracted with fp64 atomics.  One workgroup per job, in the launch that follows
// the Gram kernel: the kernel boundary is the only device-wide synchronisation (a last-arriver reduction inside the Gram kernel
// needs an agent-scope release per workgroup -- an L2 write-back on this 8-XCD part -- and cost 60 us at C640).
template <bool DIAG>
__device__ __forceinline__ void gram_reduce(const GramArgs& a, const int job, const int group, int* hA, int* hB) {
  const int tid = threadIdx.x;
  const int na = a.jobs[(long)job * NS_GRAM_JOB_INTS + 2];
  const int nb = a.jobs[(long)job * NS_GRAM_JOB_INTS + 4];
  hA[tid & 127] = a.job_hrow[(long)job * 256 + (tid & 127)];   // (both halves of the workgroup write the same values: no branch)
  hB[tid & 127] = a.job_hrow[(long)job * 256 + 128 + (tid & 127)];
  __syncthreads();
  const long n6 = 6L * a.P;
  const float* __restrict__ pj = a.part + (long)job * a.S * GR_PART;
  const int el = tid >> 2, rg = tid & 3;              // element `tid` of a tile: accumulator register rg of lane el
  const int ti = 4 * (el >> 4) + rg, tj = el & 15;    // row (A value) and column (B value) inside the tile
  // this block's GR_RG tiles: all their partial loads are requested before the first sum (a job's tiles one after the other, in
  // one workgroup, were a chain of ~2-us round trips: 39 us at C640)
  int ta_[GR_RG], tb_[GR_RG];
  bool live[GR_RG];
  double sum[GR_RG];
#pragma unroll
  for (int u = 0; u < GR_RG; u++) {
    const int tile = group * GR_RG + u;
    int ta, tb;
    if (DIAG) {
      ta = 0;
      int rem = tile;
      while (rem >= 8 - ta) {
        rem -= 8 - ta;
        ta++;
      }
      tb = ta + rem;
    } else {
      ta = tile >> 3;
      tb = tile & 7;
    }
    ta_[u] = ta;
    tb_[u] = tb;
    live[u] = tile < (DIAG ? GR_NACC : 32) && tb < nb && (DIAG || ta < na);   // (uniform)
    sum[u] = 0.0;
  }
  // (eight splits' loads requested together, summed in split order: at C640 a job has 25 splits)
  for (int s0 = 0; s0 < a.S; s0 += 8) {
    float v[8][GR_RG];
#pragma unroll
    for (int ds = 0; ds < 8; ds++)
#pragma unroll
      for (int u = 0; u < GR_RG; u++)
        v[ds][u] = (live[u] && s0 + ds < a.S) ? pj[(long)(s0 + ds) * GR_PART + (group * GR_RG + u) * 256 + tid] : 0.0f;
#pragma unroll
    for (int ds = 0; ds < 8; ds++)
#pragma unroll
      for (int u = 0; u < GR_RG; u++) sum[u] += (double)v[ds][u];
  }
#pragma unroll
  for (int u = 0; u < GR_RG; u++) {
    if (!live[u]) continue;
    const int ha = hA[ta_[u] * 16 + ti], hb = hB[tb_[u] * 16 + tj];
    const bool same_tile = DIAG && ta_[u] == tb_[u];
    // upper triangle of the slot's Gram matrix, into the upper triangle of the system (ba_finalize_kernel mirrors it: H is exactly
    // symmetric and the fp64 atomics -- executed at the memory side on this part, what bounds this launch -- are halved): an
    // off-diagonal value pair {a, b} stands for both G[a][b] and G[b][a]; where both values map to the SAME row of H (two edges
    // of one slot to one pose) that is twice the value on H's diagonal
    if (ha >= 0 && hb >= 0 && (!same_tile || ti <= tj)) {
      const bool diag_val = same_tile && ti == tj;
      const int lo = ha < hb ? ha : hb, hi = ha < hb ? hb : ha;
      atomicAdd(&a.Hd[(long)lo * n6 + hi], (!diag_val && ha == hb) ? -2.0 * sum[u] : -sum[u]);
    }
  }
  if (group != 0) return;
  if (DIAG && tid < 128 && hA[tid] >= 0) {
    double sum = 0.0;
    for (int s = 0; s < a.S; s++) sum += (double)pj[(long)s * GR_PART + GR_NACC * 256 + tid];
    atomicAdd(&a.vd[hA[tid]], -sum);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ba_schur_gram_kernel(GramArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * GR_ROUND * 256];
  __shared__ float lds_v[4 * 8 * 64];
  const int job = blockIdx.x / a.S, split = blockIdx.x % a.S;
  if (a.jobs[(long)job * NS_GRAM_JOB_INTS + 1] == a.jobs[(long)job * NS_GRAM_JOB_INTS + 3])
    gram_job<true>(a, job, split, lds, lds_v);
  else
    gram_job<false>(a, job, split, lds, lds_v);
}

//   blocks [0, n_jobs * GR_NRG) : GR_RG tiles of a Gram job's partials -> the system
//   blocks [.., +M)             : edge e: sum its linearize partials (fixed order), transform with A_i, A_j, add to the system
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(GramArgs a) {
  const int tid = threadIdx.x;
  __shared__ int hA[128], hB[128];
  __shared__ float T[ET_STRIDE];
  __shared__ double Gs[27], Gp[8][32];
  if ((int)blockIdx.x < a.n_jobs * GR_NRG) {
    const int job = blockIdx.x / GR_NRG, group = blockIdx.x % GR_NRG;
    if (a.jobs[(long)job * NS_GRAM_JOB_INTS + 1] == a.jobs[(long)job * NS_GRAM_JOB_INTS + 3])
      gram_reduce<true>(a, job, group, hA, hB);
    else
      gram_reduce<false>(a, job, group, hA, hB);
    return;
  }
  // ---------------- edge assembly ----------------
  const long n6 = 6L * a.P;
  const int e = blockIdx.x - a.n_jobs * GR_NRG;
  const int ix = (int)a.ii[e], jx = (int)a.jj[e];
  if (tid < ET_STRIDE) T[tid] = a.table[(long)e * ET_STRIDE + tid];
  // the (edge, chunk) partials of the lineariser: eight lanes per value take every eighth chunk, then the eight are added in order
  // (a fixed order; one lane per value walking all chunks was a chain of up to 75 dependent round trips at C640)
  {
    const int v = tid & 31, c0 = tid >> 5;
    double s = 0.0;
    if (v < 27)
      for (int c = c0; c < a.npart; c += 8) s += (double)a.partial[((long)e * a.npart + c) * 32 + v];
    Gp[c0][v] = s;
  }
  __syncthreads();
  if (tid < 27) Gs[tid] = ((Gp[0][tid] + Gp[1][tid]) + (Gp[2][tid] + Gp[3][tid])) + ((Gp[4][tid] + Gp[5][tid]) + (Gp[6][tid] + Gp[7][tid]));
  __syncthreads();
  if (tid < 156) {
    const double val = edge_block_entry(T + ET_AI, T + ET_AJ, Gs, tid);
    if (tid < 144) {
      const int blk = tid / 36, r = (tid % 36) / 6, c = tid % 6;
      const int rp = ((blk < 2) ? ix : jx) - a.kf0;
      const int cp = ((blk % 2 == 0) ? ix : jx) - a.kf0;
      // upper triangle of the system only: of H_ij / H_ji = H_ij^T the one above the diagonal, of a diagonal block its upper half
      if (rp >= 0 && rp < a.P && cp >= 0 && cp < a.P && (rp < cp || (rp == cp && r <= c)))
        atomicAdd(&a.Hd[(long)(6 * rp + r) * n6 + 6 * cp + c], val);
    } else {
      const int side = (tid - 144) / 6, r = (tid - 144) % 6;
      const int rp = (side == 0 ? ix : jx) - a.kf0;
      if (rp >= 0 && rp < a.P) atomicAdd(&a.vd[6 * rp + r], val);
    }
  }
}

#ifdef NS_TEST_VARIANTS   // rounds 1-5: separate accumulate kernel and one workgroup per row pair (NS_BA_UNFUSED=1: the A/B baseline)
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

#endif  // NS_TEST_VARIANTS

// fp64 -> fp32, and re-zero the accumulators for the next linearisation (every element is read by
// exactly one thread, so the system buffer needs a memset only once, when it is allocated).
// upper = 1: only the upper triangle of Hd was accumulated; H gets it mirrored.
__global__ void ba_finalize_kernel(double* __restrict__ Hd, double* __restrict__ vd, int n6, float* __restrict__ H,
                                   float* __restrict__ v, int upper) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n6 * n6) {
    const int r = idx / n6, c = idx - r * n6;
    if (upper) {
      if (r <= c) {
        const float val = (float)Hd[(long)r * n6 + c];
        Hd[(long)r * n6 + c] = 0.0;
        H[idx] = val;
        if (r < c) H[(long)c * n6 + r] = val;
      }
    } else {
      H[idx] = (float)Hd[(long)c * n6 + r];  // get_dense() hands the column-major data over as row-major (:1305-1316)
      Hd[(long)c * n6 + r] = 0.0;
    }
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

// pixels per lane of the fused lineariser (a workgroup = 64 x PPL pixels of a slot): as many as still leave ~2048 workgroups
// (the per-edge wave reductions of G, g are paid once per (edge, chunk): 190 DPP adds against ~300 instructions per pixel)
static int choose_ppl(int K, int HW) {
  if (const char* e = ns_variant_env("NS_BA_PPL")) return e[0] == '4' ? 4 : e[0] == '2' ? 2 : 1;   // (variants build: A/B)
  if ((long)K * ns_cdiv(HW, 256) >= 2048) return 4;
  if ((long)K * ns_cdiv(HW, 128) >= 2048) return 2;
  return 1;
}
// pixel splits of a Gram job: ~2048 workgroups (one per CU at a time: 512 registers per lane; 8 rounds of the chip lose 4 % to the
// last, partly filled one, 4.4 lost 12 %), each an integral number of the kernel's three-round trips where that is possible
// (C640: 75 rounds of 64 pixels -> 9 splits of 8 / 9 rounds; config #5: 225 rounds -> 7 splits of 32 / 33 rounds)
static int choose_splits(int n_jobs, int HW) {
  const int NR = ns_cdiv(HW, 64);
  int S = ns_cdiv(2048, n_jobs > 0 ? n_jobs : 1);
  const int cap = NR / 6 > 1 ? NR / 6 : 1;   // at least two trips per split: at C640 25 one-trip splits made the Gram kernel 3 us
  if (S > cap) S = cap;                       // faster and the reduce launch, which reads every split's partial, 15 us slower
  if (S < 1) S = 1;
  const int rounds = ns_cdiv(ns_cdiv(NR, S), 3) * 3;   // rounds per split, a multiple of three
  S = ns_cdiv(NR, rounds);
  return S < 1 ? 1 : S;
}

struct WsLayout {
  size_t Hd, vd, zrow, zero_end, table, partial, part, Eiz, Cii, bz, total;
};

static WsLayout ws_layout(const ns_ba_plan* plan, int HW) {
  const int M = plan->M, P = plan->P;
  WsLayout L;
  size_t off = 0;
  L.Hd = off;
  off += align256(sizeof(double) * (size_t)36 * P * P + 8);
  L.vd = off;
  off += align256(sizeof(double) * (size_t)6 * P + 8);
  L.zrow = off;
  off += align256(sizeof(float) * (size_t)(HW + 4));
  L.zero_end = off;  // [Hd, zero_end) is zero between calls
  L.table = off;
  off += align256(sizeof(float) * (size_t)ET_STRIDE * (M > 0 ? M : 1));
  const int nch = ns_cdiv(HW, 64 * choose_ppl(plan->K, HW));
  L.partial = off;
  off += align256(sizeof(float) * (size_t)32 * (M > 0 ? M : 1) * nch);
  L.part = off;
  off += align256(sizeof(float) * (size_t)GR_PART * (plan->n_jobs > 0 ? plan->n_jobs : 1) * choose_splits(plan->n_jobs, HW));
  L.Eiz = L.Cii = L.bz = off;
#ifdef NS_TEST_VARIANTS
  L.Eiz = off;
  off += align256(sizeof(float) * (size_t)M * 6 * HW + 4);
  L.Cii = off;
  off += align256(sizeof(float) * (size_t)M * HW + 4);
  L.bz = off;
  off += align256(sizeof(float) * (size_t)M * HW + 4);
#endif
  L.total = off;
  return L;
}

extern "C" size_t ns_ba_workspace_bytes(const ns_ba_plan* plan, int HW) {
  if (!plan) return 0;
  return ws_layout(plan, HW).total;
}

#ifdef NS_TEST_VARIANTS
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
#endif

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

// `ev`: nullptr, or six events recorded around the five launches of the product path (ns_reduced_camera_matrix_timed)
static int rcm_impl(const float* poses, const float* disps, const float* intrinsics, const float* extrinsics,
                    const float* disps_sens, const float* targets, const float* weights, const float* eta, const int64_t* ii,
                    const int64_t* jj, const ns_ba_plan* plan, const int32_t* index, const size_t* off, int ht, int wd, float* H,
                    float* v, float* Q, float* E, float* w, void* workspace, int ws_zeroed, void* stream, hipEvent_t* ev) {
  NS_REQUIRE(plan && index && off, "ns_reduced_camera_matrix: null plan");
  NS_REQUIRE(poses && disps && intrinsics && extrinsics && disps_sens && eta, "ns_reduced_camera_matrix: null input");
  NS_REQUIRE(H && v && Q && E && w && workspace, "ns_reduced_camera_matrix: null output/workspace");
  NS_REQUIRE(plan->M == 0 || (targets && weights && ii && jj), "ns_reduced_camera_matrix: null edge data");
  NS_REQUIRE(ht > 0 && wd > 0 && plan->P >= 0, "ns_reduced_camera_matrix: bad shape");
  NS_REQUIRE(((uintptr_t)workspace & 255) == 0, "ns_reduced_camera_matrix: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int M = plan->M, P = plan->P, K = plan->K, HW = ht * wd, n6 = 6 * P;
  NS_REQUIRE(HW >= 4, "ns_reduced_camera_matrix: maps of fewer than 4 pixels");
  const WsLayout L = ws_layout(plan, HW);
  char* ws = (char*)workspace;
  double* Hd = (double*)(ws + L.Hd);
  double* vd = (double*)(ws + L.vd);
  float* partial = (float*)(ws + L.partial);
  // Hd, vd and the zero row are adjacent in the layout: one memset, needed only the first time a workspace is used
  // (ba_finalize_kernel re-zeroes what it reads)
  if (!ws_zeroed && hipMemsetAsync(Hd, 0, L.zero_end - L.Hd, st) != hipSuccess) {
    ns_set_error("ns_reduced_camera_matrix: hipMemsetAsync failed");
    return NS_ELAUNCH;
  }
  const int32_t* kx = index + off[0];
  const int32_t* src_ptr = index + off[3];
  const int32_t* src_edge = index + off[4];
#ifdef NS_TEST_VARIANTS
  const int32_t* row_pose = index + off[2];
  static const bool unfused = [] { const char* e = ns_variant_env("NS_BA_UNFUSED"); return e != nullptr && e[0] == '1'; }();
  if (ns_variant_env("NS_BA_UNFUSED") != nullptr ? ns_variant_env("NS_BA_UNFUSED")[0] == '1' : unfused) {
    // rounds 1-5: linearise per edge -> accumulate per slot -> one workgroup per row pair
    float* Eiz = (float*)(ws + L.Eiz);
    float* Cii = (float*)(ws + L.Cii);
    float* bz = (float*)(ws + L.bz);
    const int32_t* pairs = index + off[5];
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
      hipLaunchKernelGGL(ba_finalize_kernel, dim3(ns_cdiv((long)n6 * n6, 256)), dim3(256), 0, st, Hd, vd, n6, H, v, 0);
      NS_CHECK_LAUNCH("ba_finalize_kernel");
    }
    return NS_OK;
  }
#endif
  const int ppl = choose_ppl(K, HW);
  const int nch = ns_cdiv(HW, 64 * ppl);
  float* table = (float*)(ws + L.table);
  if (ev) (void)hipEventRecord(ev[0], st);
  if (M > 0) {
    hipLaunchKernelGGL(ba_edge_table_kernel, dim3(ns_cdiv((long)M * 8, 256)), dim3(256), 0, st, poses, extrinsics, ii, jj, M, table);
    NS_CHECK_LAUNCH("ba_edge_table_kernel");
  }
  if (ev) (void)hipEventRecord(ev[1], st);
  if (K > 0) {
    LinSlotArgs a;
    a.target = targets;
    a.weight = weights;
    a.disps = disps;
    a.disps_sens = disps_sens;
    a.eta = eta;
    a.intr = intrinsics;
    a.poses = poses;
    a.extr = extrinsics;
    a.table = table;
    a.kx = kx;
    a.src_ptr = src_ptr;
    a.src_edge = src_edge;
    a.E = E;
    a.Q = Q;
    a.w = w;
    a.partial = partial;
    a.M = M;
    a.HW = HW;
    a.wd = wd;
    a.nch = nch;
    a.kf0 = plan->kf0;
    a.P = P;
    const dim3 grid(K, nch);
    if (ppl == 4)
      hipLaunchKernelGGL(ba_linearize_slot_kernel<4>, grid, dim3(256), 0, st, a);
    else if (ppl == 2)
      hipLaunchKernelGGL(ba_linearize_slot_kernel<2>, grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(ba_linearize_slot_kernel<1>, grid, dim3(256), 0, st, a);
    NS_CHECK_LAUNCH("ba_linearize_slot_kernel");
  }
  if (ev) (void)hipEventRecord(ev[2], st);
  if (plan->n_jobs + M > 0) {
    GramArgs g;
    g.E = E;
    g.Q = Q;
    g.w = w;
    g.zrow = (const float*)(ws + L.zrow);
    g.jobs = index + off[10];
    g.job_plane = index + off[11];
    g.job_hrow = index + off[12];
    g.part = (float*)(ws + L.part);
    g.Hd = Hd;
    g.vd = vd;
    g.partial = partial;
    g.table = table;
    g.ii = ii;
    g.jj = jj;
    g.HW = HW;
    g.P = P;
    g.kf0 = plan->kf0;
    g.n_jobs = plan->n_jobs;
    g.S = choose_splits(plan->n_jobs, HW);
    g.npart = nch;
    g.M = M;
    if (plan->n_jobs > 0) {
      hipLaunchKernelGGL(ba_schur_gram_kernel, dim3(plan->n_jobs * g.S), dim3(256), 0, st, g);
      NS_CHECK_LAUNCH("ba_schur_gram_kernel");
    }
    if (ev) (void)hipEventRecord(ev[3], st);
    hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(plan->n_jobs * GR_NRG + M), dim3(256), 0, st, g);
    NS_CHECK_LAUNCH("ba_schur_reduce_kernel");
  }
  if (ev) (void)hipEventRecord(ev[4], st);
  if (n6 > 0) {
    hipLaunchKernelGGL(ba_finalize_kernel, dim3(ns_cdiv((long)n6 * n6, 256)), dim3(256), 0, st, Hd, vd, n6, H, v, 1);
    NS_CHECK_LAUNCH("ba_finalize_kernel");
  }
  if (ev) (void)hipEventRecord(ev[5], st);
  return NS_OK;
}

extern "C" int ns_reduced_camera_matrix(const float* poses, const float* disps, const float* intrinsics,
                                        const float* extrinsics, const float* disps_sens, const float* targets,
                                        const float* weights, const float* eta, const int64_t* ii, const int64_t* jj,
                                        const ns_ba_plan* plan, const int32_t* index, const size_t* off, int ht,
                                        int wd, float* H, float* v, float* Q, float* E, float* w, void* workspace,
                                        int ws_zeroed, void* stream) {
  return rcm_impl(poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj, plan, index, off, ht, wd, H, v,
                  Q, E, w, workspace, ws_zeroed, stream, nullptr);
}

// The same call `reps` times with HIP events between its five launches: us_out[0..4] = mean microseconds of the edge table, the
// fused lineariser, the Gram kernel, the reduce / assembly launch, the finalisation (bench.py: the BA roofline entries are
// timed on the stream the kernels run on, live, not taken from a profile).  Synchronises the stream.
extern "C" int ns_reduced_camera_matrix_timed(const float* poses, const float* disps, const float* intrinsics,
                                              const float* extrinsics, const float* disps_sens, const float* targets,
                                              const float* weights, const float* eta, const int64_t* ii, const int64_t* jj,
                                              const ns_ba_plan* plan, const int32_t* index, const size_t* off, int ht, int wd,
                                              float* H, float* v, float* Q, float* E, float* w, void* workspace, int ws_zeroed,
                                              void* stream, int reps, float* us_out) {
  NS_REQUIRE(reps >= 1 && us_out, "ns_reduced_camera_matrix_timed: bad arguments");
  hipEvent_t ev[6];
  for (int i = 0; i < 6; i++)
    if (hipEventCreate(&ev[i]) != hipSuccess) {
      ns_set_error("ns_reduced_camera_matrix_timed: hipEventCreate failed");
      return NS_ELAUNCH;
    }
  double acc[5] = {0, 0, 0, 0, 0};
  int rc = NS_OK;
  for (int r = 0; r < reps && rc == NS_OK; r++) {
    rc = rcm_impl(poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj, plan, index, off, ht, wd, H, v, Q,
                  E, w, workspace, r == 0 ? ws_zeroed : 1, stream, ev);
    if (rc != NS_OK) break;
    if (hipEventSynchronize(ev[5]) != hipSuccess) {
      ns_set_error("ns_reduced_camera_matrix_timed: hipEventSynchronize failed");
      rc = NS_ELAUNCH;
      break;
    }
    for (int i = 0; i < 5; i++) {
      float ms = 0.0f;
      (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      acc[i] += (double)ms * 1e3;
    }
  }
  for (int i = 0; i < 6; i++) (void)hipEventDestroy(ev[i]);
  for (int i = 0; i < 5; i++) us_out[i] = (float)(acc[i] / reps);
  return rc;
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
