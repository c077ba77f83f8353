// ngp_mlp_wgrad.hip -- weight gradients of the NeRF's two MLPs with every turn-around done on the matrix cores (round 4).
// A file of its own because it is compiled with -mllvm -amdgpu-mfma-vgpr-form: the chain / turning MFMAs of this kernel write
// their 16 results to VGPRs, where the conversions read them; with the default (AGPR) form of the builtin every one of the
// ~35 temporary tiles per 32 samples came back through 16 v_accvgpr_read (560 of the loop's 1720 instructions).
// Boundary: instant-ngp's FullyFusedMLP backward inside Testbed.frame() (/root/reference/fusion/nerf_fusion.py:299).
#include "ngp_mlp_common.h"
#include <algorithm>

// ---------------------------------------------------------------------------------------------
// Weight gradients WITHOUT LDS staging and without barriers (round 4, default; NS_NGP_WGRAD=staged selects the kernel above).
//
// dW = dY X^T contracts over samples: both MFMA operands need the sample index along k, i.e. INSIDE a lane's fragment, while the
// forward / backward chains hold a sample per lane and the units inside the fragment.  The kernel above turned every activation and
// gradient around through a 46-KB LDS tile (240 two-byte LDS stores per lane and 11 barriers per 64 samples, two waves per workgroup:
// 105-167 us for 6 us worth of MFMA work, and the longest link of the training step's side stream).  The matrix core can do the
// turning itself: for a chain fragment F (lane = sample s, elements = 16 units) and the selection matrix E (E[k][j] = 1 iff unit k
// lands in column j), the product F x E has the SAME values with lane = unit and elements = samples -- exactly a k-fragment of the
// weight-gradient MFMA, because the accumulator layout of v_mfma_f32_32x32x16 (rows 4h + (r & 3) + 8 (r >> 2)) is the operand
// layout the chains already use (frag_k).  Exact: every output is one f16 value times 1 plus zeros.  Two MFMAs per 32 units.
// The features are read UNIT-MAJOR as stored (lane = unit, 8-byte loads of 4 consecutive samples): that is already the turned
// form, and the chain's form of them is one more such product.  Everything a 32-sample tile needs stays in the registers of ONE
// wave: 90 MFMAs per tile (36 chain + 30 turning + 24 weight-gradient), all twelve 32 x 32 accumulator tiles of the five weight
// matrices resident (192 registers; one wave per SIMD), the weight fragments read from a 44-KB LDS copy of the packed table.
// No barrier inside the sample loop; the four waves of a workgroup meet once, at the end, to add their accumulators in LDS in
// wave order (deterministic) into one 40-KB slab per workgroup.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tr_pair(const f16x8& a0, const f16x8& a1, const f16x8& E0, const f16x8& E1, f16x8 (&out)[2]) {
  f32x16 acc = (f32x16)0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, E0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, E1, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    out[0][r] = (_Float16)acc[r];
    out[1][r] = (_Float16)acc[8 + r];
  }
}
__device__ __forceinline__ void tr_one(const f16x8& a0, const f16x8& E0, f16x8 (&out)[2]) {
  f32x16 acc = (f32x16)0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, E0, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    out[0][r] = (_Float16)acc[r];
    out[1][r] = (_Float16)acc[8 + r];
  }
}
// dW tile += dY^T (rows = output units) x X^T (columns = input units) over the tile's 32 samples (two k chunks)
__device__ __forceinline__ void wg_acc(f32x16& acc, const f16x8 (&dyT)[2], const f16x8 (&xT)[2]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dyT[0], xT[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dyT[1], xT[1], acc, 0, 0, 0);
}
template <int NCHUNK>
__device__ __forceinline__ f32x16 layer_tile_l(const f16x8* Wf, int first, int it, int lane, const f16x8* bin) {
  f32x16 acc = (f32x16)0.0f;
#pragma unroll
  for (int cc = 0; cc < NCHUNK; cc++)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[(first + it * NCHUNK + cc) * 64 + lane], bin[cc], acc, 0, 0, 0);
  return acc;
}
// ReLU on PACKED halves, and its mask as one bit per unit in a layout of this kernel's own: pair p (registers 2p, 2p + 1 of the
// accumulator tile) of row tile `it` -> bits (8 it + p) and (16 + 8 it + p).  Four instructions per pair for the mask (add,
// shift, and, shift-or), four for the gate (shift, and, multiply by 0xffff, and).
__device__ __forceinline__ void relu_frag(const f32x16& acc, int it, f16x8* out, uint32_t& mask) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    f16x2 v = {(_Float16)acc[2 * p], (_Float16)acc[2 * p + 1]};
    v = __builtin_elementwise_max(v, (f16x2)(_Float16)0);
    out[2 * it + (p >> 2)][2 * (p & 3)] = v[0];
    out[2 * it + (p >> 2)][2 * (p & 3) + 1] = v[1];
    // (relu output: halves 0x0000 .. 0x7c00, or 0x8000 if the packed max hands back the -0 a tiny negative sum rounds to; with
    //  the sign cleared, adding 0x7fff sets bit 15 exactly when the half is non-zero, no carry out)
    const uint32_t nz = (((__builtin_bit_cast(uint32_t, v) & 0x7fff7fffu) + 0x7fff7fffu) >> 15) & 0x00010001u;
    mask |= nz << (8 * it + p);
  }
}
__device__ __forceinline__ void gate_frag(const f32x16& acc, int it, uint32_t mask, f16x8* out) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const f16x2 v = {(_Float16)acc[2 * p], (_Float16)acc[2 * p + 1]};
    const uint32_t keep = ((mask >> (8 * it + p)) & 0x00010001u) * 0xffffu;
    const f16x2 g = __builtin_bit_cast(f16x2, __builtin_bit_cast(uint32_t, v) & keep);
    out[2 * it + (p >> 2)][2 * (p & 3)] = g[0];
    out[2 * it + (p >> 2)][2 * (p & 3) + 1] = g[1];
  }
}
// one wave's accumulator tile into the workgroup's slab (LDS): stored by the first wave, added by the others
__device__ __forceinline__ void wgrad_slab(float* S, bool add, int woff, int to, int ti, int nout, int nin, int lane, const f32x16& acc) {
  const int col = lane & 31, half = lane >> 5;
  const int j = ti * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = to * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (i < nout && j < nin) {
      float* p = S + woff + i * nin + j;
      *p = add ? *p + acc[r] : acc[r];
    }
  }
}

#define WT_SLAB_FLOATS (((FW_NFRAG + BW_NFRAG) * 64 * 16) / 4)   // the fragment table's LDS doubles as the slab (11264 >= 10240 floats)
__global__ __launch_bounds__(256, 1) void ngp_mlp_wgrad_tr_kernel(MlpWgradArgs a) {
  __shared__ __attribute__((aligned(16))) f16x8 Wf0[(FW_NFRAG + BW_NFRAG) * 64];   // 44 KB
#pragma unroll 4
  for (int e = threadIdx.x; e < (FW_NFRAG + BW_NFRAG) * 64; e += 256) Wf0[e] = a.frags[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const long N = a.N, cnt = ngp_count(a.N, a.n_dev), exact = ngp_exact(a.N, a.n_dev);
  // selection matrices: E0 sends unit k(h, q) of an even chunk to column k, E1 unit k of an odd chunk to column 16 + k
  f16x8 E0, E1;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int kq = frag_k(0, h, q);
    E0[q] = (j == kq) ? (_Float16)1.0f : (_Float16)0.0f;
    E1[q] = (j == 16 + kq) ? (_Float16)1.0f : (_Float16)0.0f;
  }
  f32x16 w5[2], w4[2][2], w3[2], w2[2], w1[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    w5[t] = w3[t] = w2[t] = w1[t] = (f32x16)0.0f;
    w4[t][0] = w4[t][1] = (f32x16)0.0f;
  }
  typedef _Float16 f16x4l __attribute__((ext_vector_type(4)));
  const long ntile = (cnt + 31) / 32;
  // the tile's global inputs are fetched ONE TILE AHEAD: with one wave per SIMD nothing else covers a load's ~2 us
  struct TileIn {
    f16x4l xlo[2], xhi[2];   // features, unit-major: lane = unit j, samples s0 + 16 c + 4 h + {0..3}, {8..11}
    float dx, dy, dz;        // direction of sample s0 + j
    f16x4l go;               // loss gradient (r, g, b, d) of sample s0 + j
  };
  auto fetch = [&](long tile, TileIn& in) {
    const long s0 = tile * 32;
    const bool live = tile < ntile;
    const _Float16* row = a.featT + (long)j * N + (live ? s0 : 0) + 4 * h;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      in.xlo[c] = *reinterpret_cast<const f16x4l*>(row + 16 * c);
      in.xhi[c] = *reinterpret_cast<const f16x4l*>(row + 16 * c + 8);
    }
    const long np = s0 + j;
    const long ns = live && np < cnt ? np : 0;
    in.dx = a.dirs[ns * 3];
    in.dy = a.dirs[ns * 3 + 1];
    in.dz = a.dirs[ns * 3 + 2];
    in.go = (f16x4l)(_Float16)0;
    if (live && np < exact) in.go = *reinterpret_cast<const f16x4l*>(a.dLdout + np * 4);
  };
  TileIn cur, nxt;
  fetch((long)blockIdx.x * 4 + wave, cur);
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntile; tile += (long)gridDim.x * 4) {
    const long s0 = tile * 32;
    fetch(tile + (long)gridDim.x * 4, nxt);
    // (an offset the compiler cannot see through: the 36 fragment reads of a tile are loop-invariant LDS loads, and hoisted out of
    //  the tile loop they would occupy 144 registers for the whole kernel)
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const f16x8* Wf = Wf0 + opq;
    const f16x8* Wb = Wf + FW_NFRAG * 64;
    f16x8 xT[2];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xT[c][q] = cur.xlo[c][q];
        xT[c][4 + q] = cur.xhi[c][q];
      }
    if (s0 + 32 > cnt) {          // (wave-uniform: the last tile) slots the marcher did not fill hold stale features
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (s0 + 16 * c + frag_k(0, h, q) >= cnt) xT[c][q] = (_Float16)0;
    }
    const f16x4l go = cur.go;
    // ---- forward chain (the instruction sequence of ngp_mlp_fwd_kernel), every activation also turned for the weight gradients ----
    f16x8 x[2];
    tr_pair(xT[0], xT[1], E0, E1, x);                 // lane = sample, elements = units: the chain's form of the features
    uint32_t m1 = 0, m3 = 0, m4 = 0;
    f16x8 h1T[2][2], cinT[2], h3T[2][2], h4T[2][2];
    f16x8 cin[2];
    {
      f16x8 h1[4];
#pragma unroll
      for (int it = 0; it < 2; it++) relu_frag(layer_tile_l<2>(Wf, FW_L1, it, lane, x), it, h1, m1);
      tr_pair(h1[0], h1[1], E0, E1, h1T[0]);
      tr_pair(h1[2], h1[3], E0, E1, h1T[1]);
      const f32x16 acc = layer_tile_l<4>(Wf, FW_L2, 0, lane, h1);
#pragma unroll
      for (int r = 0; r < 8; r++) cin[0][r] = (_Float16)acc[r];
      cin[1] = sh_chunk(cur.dx, cur.dy, cur.dz, h);
      tr_pair(cin[0], cin[1], E0, E1, cinT);
    }
    f16x8 d4[4];
    {
      f16x8 h3[4], h4[4];
#pragma unroll
      for (int it = 0; it < 2; it++) relu_frag(layer_tile_l<2>(Wf, FW_L3, it, lane, cin), it, h3, m3);
      tr_pair(h3[0], h3[1], E0, E1, h3T[0]);
      tr_pair(h3[2], h3[3], E0, E1, h3T[1]);
#pragma unroll
      for (int it = 0; it < 2; it++) relu_frag(layer_tile_l<4>(Wf, FW_L4, it, lane, h3), it, h4, m4);
      tr_pair(h4[0], h4[1], E0, E1, h4T[0]);
      tr_pair(h4[2], h4[3], E0, E1, h4T[1]);
    }
    // ---- backward chain; every gradient turned and contracted with the turned activation of its layer's input ----
    f16x8 d5 = (f16x8)(_Float16)0;
    if (h == 0) {
      d5[0] = go[0];
      d5[1] = go[1];
      d5[2] = go[2];
    }
    {
      f16x8 d5T[2];
      tr_one(d5, E0, d5T);                            // W5: dY = d5 (16 rows), X = h4
      wg_acc(w5[0], d5T, h4T[0]);
      wg_acc(w5[1], d5T, h4T[1]);
    }
#pragma unroll
    for (int it = 0; it < 2; it++) gate_frag(layer_tile_l<1>(Wb, BW_L5, it, lane, &d5), it, m4, d4);
    {
      f16x8 d4T[2][2];
      tr_pair(d4[0], d4[1], E0, E1, d4T[0]);
      tr_pair(d4[2], d4[3], E0, E1, d4T[1]);          // W4: dY = d4, X = h3
#pragma unroll
      for (int to = 0; to < 2; to++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) wg_acc(w4[to][ti], d4T[to], h3T[ti]);
    }
    f16x8 d3[4];
#pragma unroll
    for (int it = 0; it < 2; it++) gate_frag(layer_tile_l<4>(Wb, BW_L4, it, lane, d4), it, m3, d3);
    {
      f16x8 d3T[2][2];
      tr_pair(d3[0], d3[1], E0, E1, d3T[0]);
      tr_pair(d3[2], d3[3], E0, E1, d3T[1]);          // W3: dY = d3, X = cin (32 rows)
      wg_acc(w3[0], d3T[0], cinT);
      wg_acc(w3[1], d3T[1], cinT);
    }
    f16x8 dd;
    {
      const f32x16 acc = layer_tile_l<4>(Wb, BW_L3, 0, lane, d3);
#pragma unroll
      for (int r = 0; r < 8; r++) dd[r] = (_Float16)acc[r];
      if (h == 0) dd[0] = (_Float16)((float)dd[0] + (float)go[3]);
      f16x8 ddT[2];
      tr_one(dd, E0, ddT);                            // W2: dY = dd (16 rows), X = h1
      wg_acc(w2[0], ddT, h1T[0]);
      wg_acc(w2[1], ddT, h1T[1]);
    }
    {
      f16x8 d1[4], d1T[2][2];
#pragma unroll
      for (int it = 0; it < 2; it++) gate_frag(layer_tile_l<1>(Wb, BW_L2, it, lane, &dd), it, m1, d1);
      tr_pair(d1[0], d1[1], E0, E1, d1T[0]);
      tr_pair(d1[2], d1[3], E0, E1, d1T[1]);          // W1: dY = d1, X = the features
      wg_acc(w1[0], d1T[0], xT);
      wg_acc(w1[1], d1T[1], xT);
    }
    cur = nxt;
  }
  // ---- the four waves' accumulators -> one slab per workgroup, added in wave order ----
  float* S = reinterpret_cast<float*>(Wf0);
  for (int w = 0; w < 4; w++) {
    __syncthreads();          // (first round: every wave has read its last weight fragment)
    if (wave == w) {
      const bool add = w > 0;
      wgrad_slab(S, add, W5_OFF, 0, 0, 16, 64, lane, w5[0]);
      wgrad_slab(S, add, W5_OFF, 0, 1, 16, 64, lane, w5[1]);
#pragma unroll
      for (int to = 0; to < 2; to++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) wgrad_slab(S, add, W4_OFF, to, ti, 64, 64, lane, w4[to][ti]);
      wgrad_slab(S, add, W3_OFF, 0, 0, 64, 32, lane, w3[0]);
      wgrad_slab(S, add, W3_OFF, 1, 0, 64, 32, lane, w3[1]);
      wgrad_slab(S, add, W2_OFF, 0, 0, 16, 64, lane, w2[0]);
      wgrad_slab(S, add, W2_OFF, 0, 1, 16, 64, lane, w2[1]);
      wgrad_slab(S, add, W1_OFF, 0, 0, 64, 32, lane, w1[0]);
      wgrad_slab(S, add, W1_OFF, 1, 0, 64, 32, lane, w1[1]);
    }
  }
  __syncthreads();
  float4* P = reinterpret_cast<float4*>(a.partial + (long)blockIdx.x * W_TOTAL);
  const float4* S4 = reinterpret_cast<const float4*>(S);
  for (int e = threadIdx.x; e < W_TOTAL / 4; e += 256) P[e] = S4[e];
}


// Tried and dropped (round 4): the same kernel in TWO ROLES (workgroup A: dW4, dW5; workgroup B: dW3, dW2, dW1; 96 accumulator
// registers each, both chains recomputed by both) so that a wave leaves room for another kernel's waves on its SIMD: 301 registers
// without spills (256 only with 40 of them in scratch), 60.8 us stand-alone against 44.4, and no gain in the training step
// (0.289-0.293 against 0.282-0.286 ms) or in the pipeline (123.7 against 122.6-124.8 frames/s).  One role it stays.

// wgs = slabs of a.partial that end up written (the caller reduces exactly that many)
int ngp_mlp_wgrad_tr_launch(const MlpWgradArgs& a, int wgs, hipStream_t stream) {
  hipLaunchKernelGGL(ngp_mlp_wgrad_tr_kernel, dim3(wgs), dim3(256), 0, stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_tr_kernel");
  return NS_OK;
}
