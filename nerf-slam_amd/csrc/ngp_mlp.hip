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
#include <algorithm>

#include "ngp_mlp_common.h"
#include "ngp_adam.h"
// store the 16 (tile 0, tile 1) pairs of one 32-unit tile unit-major: row acc_unit(it, h, r), samples np, np + 1
__device__ __forceinline__ void store_tile(_Float16* __restrict__ dst, long N, uint32_t boff, int it, const f16x8* t0,
                                           const f16x8* t1) {  // t0/t1: chunks [2 it], [2 it + 1] of tile 0 / 1
#pragma unroll
  for (int r = 0; r < 16; r++) *um_at(dst + (long)urow(it, r) * N, boff) = pack2(t0[r >> 3][r & 7], t1[r >> 3][r & 7]);
}

struct MlpFwdArgs {
  const _Float16* W;      // packed weights
  const _Float16* featT;  // [32,N] unit-major encoding
  const float* dirs;      // [N,3]
  _Float16* out;          // [N,4] (r,g,b raw, log-density)
  // unit-major activations for the backward pass (all null in inference)
  _Float16* h1T;          // [64,N]
  _Float16* cinT;         // [32,N]
  _Float16* h3T;          // [64,N]
  _Float16* h4T;          // [64,N]
  long N;
  const int* n_dev;       // optional: device sample count (<= N); N stays the row stride of the unit-major tensors
  uint32_t* masks;        // optional [3 layers (h1, h3, h4)][2 lane halves][N]: ReLU bit masks for the backward pass (see below)
  const f16x8* frags;     // optional: ngp_mlp_pack_frags_kernel's table (then W is not read: 6 16-byte loads per thread instead
                          // of 48 element gathers with their index arithmetic, per workgroup)
};

// ReLU masks.  The backward pass needs, per hidden unit, only whether the forward activation was positive; re-reading the f16
// activations for that cost 384 B per sample and put three rounds of 16 dependent loads on the critical path of the
// activation-gradient kernel.  The forward pass therefore also writes ONE BIT per unit: lane (j, h) holds, for its two samples,
// the 32 units {acc_unit(it, h, r)} of each 64-wide layer -- exactly the units the backward lane (j, h) will hold in its
// accumulators (same instruction, same layout).  Its word for sample n (round 6 layout: the two halves of a PACKED pair of
// accumulator registers 2p, 2p + 1 of tile `it` sit 16 bits apart, so that both kernels work on packed halves):
//     bit (8 it + p) = [h(unit of register 2p) > 0],   bit (16 + 8 it + p) = [h(unit of register 2p + 1) > 0]
// stored at masks[layer][h][n]: the two words of a lane are adjacent (8 B per lane, 256 contiguous bytes per half-wave).  24 B per
// sample written by the forward pass, 24 B read by the backward pass at its very start.
//
// relu_tile: ReLU + f16 rounding + mask bits of one 32-unit accumulator tile, 7 vector instructions per PAIR of values (v_max_i32
// x 2, v_cvt_pk_f16_f32, add / shift / and = "is the half non-zero", v_lshl_or_b32).  The plain form -- fmaxf (a
// canonicalising second v_max_f32 each), a compare + select + shift-or per value for the bit -- was 11 per pair, and with 1342
// vector instructions against 48 MFMAs per 64 samples the forward kernel is bound by exactly these (round 6).  Same values, same
// bits: the integer max of a negative value's or -0's bits with 0 is +0, so "non-zero half" is "activation > 0".
__device__ __forceinline__ void relu_tile(const f32x16& acc, int it, f16x8* out, uint32_t& mask) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    // (no inline asm on an accumulator: the compiler places the MFMA -> VALU wait states only for instructions it can see)
    // ReLU on the float's bits as a signed integer (negative values and -0 have the sign bit set -> +0): ONE v_max_i32; fmaxf
    // and fmed3 are each emitted with a canonicalising second v_max_f32
    const int ia = __builtin_bit_cast(int, (float)acc[2 * p]), ib = __builtin_bit_cast(int, (float)acc[2 * p + 1]);
    const float a = __builtin_bit_cast(float, ia > 0 ? ia : 0), b = __builtin_bit_cast(float, ib > 0 ? ib : 0);
    const f16x2 v = {(_Float16)a, (_Float16)b};
    // both halves are >= +0: adding 0x7fff carries into bit 15 exactly when a half is non-zero
    const uint32_t nz = ((__builtin_bit_cast(uint32_t, v) + 0x7fff7fffu) >> 15) & 0x00010001u;
    mask |= nz << (8 * it + p);
    out[p >> 2][2 * (p & 3)] = v[0];
    out[p >> 2][2 * (p & 3) + 1] = v[1];
  }
}
__device__ __forceinline__ uint2* mask_at(uint32_t* masks, int layer, int h, long N, long np) {
  return reinterpret_cast<uint2*>(masks + ((long)(layer * 2 + h) * N + np));
}

// SAVE = the activation buffers are written (round 2's training form, inference never).  As a template parameter, not a run-time
// test: the trainer's form (masks only) then is compiled without the store path's 64-bit row addresses -- 204 -> 148 registers,
// three waves per SIMD instead of two, 31 -> 26 us (four waves: 128 registers + 44 B of scratch, 24 us, not taken).
template <bool SAVE>
__global__ __launch_bounds__(256, SAVE ? 2 : 4) void ngp_mlp_fwd_kernel(MlpFwdArgs a) {
  __shared__ f16x8 Wf[FW_NFRAG * 64];
  if (a.frags != nullptr) {      // (workgroup-uniform)
#pragma unroll
    for (int e = threadIdx.x; e < FW_NFRAG * 64; e += 256) Wf[e] = a.frags[e];
  } else {
    fill_frags<false>(Wf, a.W, W1_OFF, 64, 32, FW_L1);
    fill_frags<false>(Wf, a.W, W2_OFF, 16, 64, FW_L2);
    fill_frags<false>(Wf, a.W, W3_OFF, 64, 32, FW_L3);
    fill_frags<false>(Wf, a.W, W4_OFF, 64, 64, FW_L4);
    fill_frags<false>(Wf, a.W, W5_OFF, 16, 64, FW_L5);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const long N = a.N, cnt = ngp_count(a.N, a.n_dev);
  constexpr bool save = SAVE;
  for (int iter = 0; iter < MLP_ITERS; iter++) {
    const long n0 = (((long)blockIdx.x * MLP_ITERS + iter) * 4 + wave) * 64;
    if (n0 >= cnt) return;  // wave-uniform
    const bool ok = n0 + 2 * j < cnt;  // cnt is even: the pair (np, np + 1) is valid or not as a whole
    const long np = ok ? n0 + 2 * j : 0;
    const uint32_t boff = lane_bytes(h, N, np);
    // input: 2 chunks x 8 units, both tiles in one dword
    f16x8 x[2][2];  // [tile][chunk]
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const f16x2 v = unpack2(*um_at(a.featT + (long)ufrag(cc, q) * N, boff));
        x[0][cc][q] = ok ? v[0] : (_Float16)0;
        x[1][cc][q] = ok ? v[1] : (_Float16)0;
      }
    // layer-major over both sample tiles: an activation is stored as soon as it exists and dies after the next layer
    const bool st = save && ok;
    _Float16 res[2][4];
    f16x8 h1[2][4], cin[2][2];
    uint32_t mk_h1[2] = {0u, 0u}, mk_h3[2] = {0u, 0u}, mk_h4[2] = {0u, 0u};
#pragma unroll
    for (int it = 0; it < 2; it++) {  // L1 32 -> 64, ReLU
#pragma unroll
      for (int t = 0; t < 2; t++) relu_tile(layer_tile<2>(Wf, FW_L1, it, lane, x[t]), it, &h1[t][2 * it], mk_h1[t]);
      if (st) store_tile(a.h1T, N, boff, it, &h1[0][2 * it], &h1[1][2 * it]);
    }
    if (ok && a.masks)
      *mask_at(a.masks, 0, h, N, np) = make_uint2(mk_h1[0], mk_h1[1]);
#pragma unroll
    for (int t = 0; t < 2; t++) {  // L2 64 -> 16 (rows 16..31 of the tile are padding) + direction encoding
      const f32x16 acc = layer_tile<4>(Wf, FW_L2, 0, lane, h1[t]);
#pragma unroll
      for (int r = 0; r < 8; r++) cin[t][0][r] = (_Float16)acc[r];
      res[t][3] = cin[t][0][0];  // log-density = unit 0 (held by h == 0)
      const long ns = ok ? np + t : 0;
      cin[t][1] = sh_chunk(a.dirs[ns * 3], a.dirs[ns * 3 + 1], a.dirs[ns * 3 + 2], h);
    }
    if (st) store_tile(a.cinT, N, boff, 0, cin[0], cin[1]);
    f16x8 h3[2][4];
#pragma unroll
    for (int it = 0; it < 2; it++) {  // L3 32 -> 64, ReLU
#pragma unroll
      for (int t = 0; t < 2; t++) relu_tile(layer_tile<2>(Wf, FW_L3, it, lane, cin[t]), it, &h3[t][2 * it], mk_h3[t]);
      if (st) store_tile(a.h3T, N, boff, it, &h3[0][2 * it], &h3[1][2 * it]);
    }
    if (ok && a.masks)
      *mask_at(a.masks, 1, h, N, np) = make_uint2(mk_h3[0], mk_h3[1]);
    f16x8 h4[2][4];
#pragma unroll
    for (int it = 0; it < 2; it++) {  // L4 64 -> 64, ReLU
#pragma unroll
      for (int t = 0; t < 2; t++) relu_tile(layer_tile<4>(Wf, FW_L4, it, lane, h3[t]), it, &h4[t][2 * it], mk_h4[t]);
      if (st) store_tile(a.h4T, N, boff, it, &h4[0][2 * it], &h4[1][2 * it]);
    }
    if (ok && a.masks)
      *mask_at(a.masks, 2, h, N, np) = make_uint2(mk_h4[0], mk_h4[1]);
#pragma unroll
    for (int t = 0; t < 2; t++) {  // L5 64 -> 16
      const f32x16 acc = layer_tile<4>(Wf, FW_L5, 0, lane, h4[t]);
      res[t][0] = (_Float16)acc[0];
      res[t][1] = (_Float16)acc[1];
      res[t][2] = (_Float16)acc[2];
    }
    if (ok && h == 0) {
      const f16x8 o = {res[0][0], res[0][1], res[0][2], res[0][3], res[1][0], res[1][1], res[1][2], res[1][3]};
      *reinterpret_cast<f16x8*>(a.out + np * 4) = o;
    }
  }
}

struct MlpBwdArgs {
  const _Float16* W;      // packed weights (natural [out][in]; the transposed fragments are gathered in-kernel)
  const _Float16* dLdout; // [N,4]
  const _Float16 *h1T, *h3T, *h4T;  // saved activations (ReLU masks)
  _Float16* dLdfeatT;     // [32,N] unit-major
  _Float16 *d5T, *d4T, *d3T, *ddT, *d1T;  // [16,N] [64,N] [64,N] [16,N] [64,N] unit-major output gradients
  long N;
  const int* n_dev;
  const uint32_t* masks;  // optional: the forward pass's ReLU bit masks (then h1T / h3T / h4T are not read)
  const f16x8* frags;     // optional: packed fragment table (see MlpFwdArgs)
};

// dy = relu'(h) * f16(acc) for one 32-unit tile of both sample tiles; returns the B chunks and stores unit-major
__device__ __forceinline__ void mask_tile(const f32x16& acc0, const f32x16& acc1, const _Float16* __restrict__ hT,
                                          _Float16* __restrict__ dT, long N, uint32_t boff, bool ok, int it,
                                          f16x8* o0, f16x8* o1) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const long row = (long)urow(it, r) * N;
    const f16x2 act = unpack2(*um_at(hT + row, boff));
    const _Float16 v0 = (float)act[0] > 0.0f ? (_Float16)acc0[r] : (_Float16)0;
    const _Float16 v1 = (float)act[1] > 0.0f ? (_Float16)acc1[r] : (_Float16)0;
    o0[r >> 3][r & 7] = v0;
    o1[r >> 3][r & 7] = v1;
    if (ok) *um_at(dT + row, boff) = pack2(v0, v1);
  }
}

// the same with the ReLU derivative taken from the forward pass's bit mask (layout: see relu_tile): no loads, and on PACKED pairs
// -- v_cvt_pk_f16_f32, then the pair's two mask bits (16 apart) widened to 0x0000 / 0xffff per half with one 24-bit multiply and
// applied with one AND: 5 vector instructions per pair where the per-value select took 9 (round 6)
__device__ __forceinline__ void mask_tile_bits(const f32x16& acc0, const f32x16& acc1, uint32_t m0, uint32_t m1,
                                               _Float16* __restrict__ dT, long N, uint32_t boff, bool ok, int it, f16x8* o0,
                                               f16x8* o1) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const f16x2 a = {(_Float16)acc0[2 * p], (_Float16)acc0[2 * p + 1]};
    const f16x2 b = {(_Float16)acc1[2 * p], (_Float16)acc1[2 * p + 1]};
    const uint32_t k0 = __umul24((m0 >> (8 * it + p)) & 0x00010001u, 0xffffu);
    const uint32_t k1 = __umul24((m1 >> (8 * it + p)) & 0x00010001u, 0xffffu);
    const f16x2 v0 = __builtin_bit_cast(f16x2, __builtin_bit_cast(uint32_t, a) & k0);
    const f16x2 v1 = __builtin_bit_cast(f16x2, __builtin_bit_cast(uint32_t, b) & k1);
    o0[p >> 2][2 * (p & 3)] = v0[0];
    o0[p >> 2][2 * (p & 3) + 1] = v0[1];
    o1[p >> 2][2 * (p & 3)] = v1[0];
    o1[p >> 2][2 * (p & 3) + 1] = v1[1];
    if (ok && dT != nullptr) {   // (dT: wave-uniform)
      *um_at(dT + (long)urow(it, 2 * p) * N, boff) = pack2(v0[0], v1[0]);
      *um_at(dT + (long)urow(it, 2 * p + 1) * N, boff) = pack2(v0[1], v1[1]);
    }
  }
}

// LEAN (round 6): the trainer's form -- bit masks in, dL/dfeature out, none of the five unit-major gradient tensors -- compiled
// without their store paths: as run-time tests of wave-uniform null pointers they were ~100 scalar branches through the chain.
template <bool BITS, bool LEAN = false>
__global__ __launch_bounds__(256, 4) void ngp_mlp_bwd_kernel(MlpBwdArgs a) {
  __shared__ f16x8 Wf[BW_NFRAG * 64];
  if (a.frags != nullptr) {      // (workgroup-uniform)
#pragma unroll
    for (int e = threadIdx.x; e < BW_NFRAG * 64; e += 256) Wf[e] = a.frags[FW_NFRAG * 64 + e];
  } else {
    fill_frags<true>(Wf, a.W, W5_OFF, 16, 64, BW_L5);
    fill_frags<true>(Wf, a.W, W4_OFF, 64, 64, BW_L4);
    fill_frags<true>(Wf, a.W, W3_OFF, 64, 32, BW_L3);
    fill_frags<true>(Wf, a.W, W2_OFF, 16, 64, BW_L2);
    fill_frags<true>(Wf, a.W, W1_OFF, 64, 32, BW_L1);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const long N = a.N, cnt = ngp_count(a.N, a.n_dev), exact = ngp_exact(a.N, a.n_dev);
  for (int iter = 0; iter < MLP_ITERS; iter++) {
    const long n0 = (((long)blockIdx.x * MLP_ITERS + iter) * 4 + wave) * 64;
    if (n0 >= cnt) return;
    // every lane runs every MFMA (lane i also supplies row i of A): out-of-range pairs read pair 0 and store nothing
    const bool ok = n0 + 2 * j < cnt;
    const long np = ok ? n0 + 2 * j : 0;
    const uint32_t boff = lane_bytes(h, N, np);
    f16x8 go = *reinterpret_cast<const f16x8*>(a.dLdout + np * 4);  // (r,g,b,d) of samples np, np + 1
#pragma unroll
    for (int t = 0; t < 2; t++)
      if (np + t >= exact) go[4 * t] = go[4 * t + 1] = go[4 * t + 2] = go[4 * t + 3] = (_Float16)0;
    uint2 mk1 = make_uint2(0, 0), mk3 = mk1, mk4 = mk1;
    if (BITS) {
      mk1 = *reinterpret_cast<const uint2*>(a.masks + ((long)(0 * 2 + h) * N + np));
      mk3 = *reinterpret_cast<const uint2*>(a.masks + ((long)(1 * 2 + h) * N + np));
      mk4 = *reinterpret_cast<const uint2*>(a.masks + ((long)(2 * 2 + h) * N + np));
    }
    // dY5: units 0..2 = colour gradients (held by h == 0, q = 0..2), rest zero
    f16x8 d5[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      d5[t] = (f16x8)(_Float16)0;
      if (h == 0) {
        d5[t][0] = go[4 * t];
        d5[t][1] = go[4 * t + 1];
        d5[t][2] = go[4 * t + 2];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (!LEAN && ok && a.d5T != nullptr) *um_at(a.d5T + (long)ufrag(0, q) * N, boff) = pack2(d5[0][q], d5[1][q]);
    // layer 5^T (K = 16) -> d(h4), ReLU' of layer 4
    f16x8 d4[2][4], d3[2][4], dd[2], d1[2][4];
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 a0 = layer_tile<1>(Wf, BW_L5, it, lane, &d5[0]);
      const f32x16 a1 = layer_tile<1>(Wf, BW_L5, it, lane, &d5[1]);
      if (BITS) {
        mask_tile_bits(a0, a1, mk4.x, mk4.y, LEAN ? nullptr : a.d4T, N, boff, ok, it, &d4[0][2 * it], &d4[1][2 * it]);
      } else {
        asm volatile("" ::: "memory");  // keep the mask loads of later tiles / layers from being hoisted up here
        mask_tile(a0, a1, a.h4T, a.d4T, N, boff, ok, it, &d4[0][2 * it], &d4[1][2 * it]);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 a0 = layer_tile<4>(Wf, BW_L4, it, lane, d4[0]);
      const f32x16 a1 = layer_tile<4>(Wf, BW_L4, it, lane, d4[1]);
      if (BITS) {
        mask_tile_bits(a0, a1, mk3.x, mk3.y, LEAN ? nullptr : a.d3T, N, boff, ok, it, &d3[0][2 * it], &d3[1][2 * it]);
      } else {
        asm volatile("" ::: "memory");  // keep the mask loads of later tiles / layers from being hoisted up here
        mask_tile(a0, a1, a.h3T, a.d3T, N, boff, ok, it, &d3[0][2 * it], &d3[1][2 * it]);
      }
    }
    // layer 3^T -> d(cin); only the density half (units 0..15 = registers 0..7) flows on; the density gradient joins unit 0
    {
      const f32x16 a0 = layer_tile<4>(Wf, BW_L3, 0, lane, d3[0]);
      const f32x16 a1 = layer_tile<4>(Wf, BW_L3, 0, lane, d3[1]);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        dd[0][r] = (_Float16)a0[r];
        dd[1][r] = (_Float16)a1[r];
      }
      if (h == 0) {
        dd[0][0] = (_Float16)((float)dd[0][0] + (float)go[3]);
        dd[1][0] = (_Float16)((float)dd[1][0] + (float)go[7]);
      }
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (!LEAN && ok && a.ddT != nullptr) *um_at(a.ddT + (long)ufrag(0, q) * N, boff) = pack2(dd[0][q], dd[1][q]);
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 a0 = layer_tile<1>(Wf, BW_L2, it, lane, &dd[0]);
      const f32x16 a1 = layer_tile<1>(Wf, BW_L2, it, lane, &dd[1]);
      if (BITS) {
        mask_tile_bits(a0, a1, mk1.x, mk1.y, LEAN ? nullptr : a.d1T, N, boff, ok, it, &d1[0][2 * it], &d1[1][2 * it]);
      } else {
        asm volatile("" ::: "memory");  // keep the mask loads of later tiles / layers from being hoisted up here
        mask_tile(a0, a1, a.h1T, a.d1T, N, boff, ok, it, &d1[0][2 * it], &d1[1][2 * it]);
      }
    }
    {
      const f32x16 a0 = layer_tile<4>(Wf, BW_L1, 0, lane, d1[0]);
      const f32x16 a1 = layer_tile<4>(Wf, BW_L1, 0, lane, d1[1]);
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (ok) *um_at(a.dLdfeatT + (long)urow(0, r) * N, boff) = pack2((_Float16)a0[r], (_Float16)a1[r]);
    }
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
  const int* n_dev;
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
  const long cnt = ngp_count(a.N, a.n_dev);
  const long per = ((cnt + a.ksplit - 1) / a.ksplit + 63) / 64 * 64;
  const long n0 = (long)blockIdx.x * per, n1 = min(cnt, n0 + per);
  for (long nb = n0; nb < n1; nb += 64) {
    __syncthreads();
    // stage: row r (0..nout-1: dYT, then XT) x 8 pieces of 16 B
    for (int piece = tid; piece < nrows * 8; piece += 256) {
      const int r = piece >> 3, s = piece & 7;
      const _Float16* src = (r < L.nout ? L.dYT + (long)r * a.N : L.XT + (long)(r - L.nout) * a.N) + nb + s * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (nb + s * 8 + 8 <= cnt) {
        v = *reinterpret_cast<const uint4*>(src);
      } else {
        _Float16 t[8];
        for (int q = 0; q < 8; q++) t[q] = (nb + s * 8 + q < cnt) ? src[q] : (_Float16)0;
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

// grid W_TOTAL / 16: thread (idx = tid & 15, group = tid >> 4) sums every 16th split of weight 16 * block + idx with all of
// its loads in flight at once (the round-2 form walked 64 splits per thread one dependent load at a time: 34 us for 10 MB);
// the sixteen groups meet in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void ngp_mlp_wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                                   float* __restrict__ grad) {
  __shared__ float part[16][17];
  const int idx = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + idx;
  float s = 0.0f;
  for (int k0 = grp; k0 < ksplit; k0 += 16 * 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int k = k0 + 16 * u;
      v[u] = k < ksplit ? partial[(long)k * W_TOTAL + i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) s += v[u];
  }
  part[grp][idx] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; q++) t += part[q][idx];
    grad[i] += t;
  }
}

// ---------------------------------------------------------------------------------------------
// FUSED backward pass (round 3): forward recompute + activation gradients + weight gradients in ONE kernel.
//
// Round 2's sequence moved ~1.95 KB per sample through HBM for the two MLPs: the forward pass wrote the four saved activations
// (448 B), the activation-gradient kernel wrote five gradient tensors (448 B) and dL/dfeature, and the weight-gradient kernel read
// all nine of them back (960 B) -- for 48 MFMAs per 32 samples.  Here a workgroup keeps all of it on chip:
//   * the forward chain is RECOMPUTED from the 64-byte feature vector (24 MFMAs per 32 samples, the same instruction sequence
//     as ngp_mlp_fwd_kernel, hence the same bits: the ReLU derivative is the sign of the recomputed activation);
//   * the backward chain follows in the same registers;
//   * dW = dY X^T contracts over SAMPLES, i.e. needs both operands with the sample index along k, while the chains hold them
//     with the sample index across lanes: each layer's (dY, X) pair of the workgroup's 128 samples is transposed through one
//     34-KB LDS tile [unit][sample] and read back as MFMA operands (16-byte reads); the 12 output tiles of the five weight
//     matrices are spread over the 4 waves (3 accumulator tiles = 48 VGPRs each) and stay in registers for the whole launch.
// HBM traffic per sample: 64 B features + 12 B direction + 8 B loss gradient read, 64 B dL/dfeature written: 148 B.
// Per workgroup one 40-KB slab of partial weight gradients at the end (summed by ngp_mlp_wgrad_reduce_kernel in slab order:
// deterministic).  The forward kernel proper only writes the network output then (no activation buffers).
// ---------------------------------------------------------------------------------------------
#define FU_SP 136   // halfs per stage row: 128 samples + 8 pad (272 B: 16-byte aligned rows)
// stage rows: the X operands of the five weight gradients stay resident for the whole block, the dY operand is rewritten per layer
#define FU_XF 0      // features   (32 rows)
#define FU_XH1 32    // h1         (64)
#define FU_XC 96     // cin        (32)
#define FU_XH3 128   // h3         (64)
#define FU_XH4 192   // h4         (64)
#define FU_DY 256    // dY         (64)
#define FU_ROWS 320

struct MlpFusedArgs {
  const _Float16* W;
  const _Float16* featT;   // [32,N] unit-major
  const float* dirs;       // [N,3]
  const _Float16* dLdout;  // [N,4]
  _Float16* dLdfeatT;      // [32,N] unit-major
  float* partial;          // [gridDim.x][W_TOTAL]
  long N;
  const int* n_dev;
};

// one B-layout chunk (element q of lane half h = unit frag_k(cc, h, q)) into the stage tile, column `col`
__device__ __forceinline__ void stage_chunk(_Float16* stage, int row0, const f16x8& v, int cc, int h, int col) {
#pragma unroll
  for (int q = 0; q < 8; q++) stage[(row0 + frag_k(cc, h, q)) * FU_SP + col] = v[q];
}

// one 32 x 32 tile of dW over the 128 staged samples: rows (to) of dY (nout valid rows), rows (ti) of X at xrow0 (nin valid rows)
__device__ __forceinline__ void wgrad_tile(const _Float16* stage, int xrow0, int to, int ti, int nout, int nin, int lane,
                                           f32x16& acc) {
  const int col = lane & 31, half = lane >> 5;
  const int ro = to * 32 + col, ri = ti * 32 + col;
  const bool oko = ro < nout, oki = ri < nin;
  const _Float16* pa = stage + (FU_DY + (oko ? ro : 0)) * FU_SP + 8 * half;
  const _Float16* pb = stage + (xrow0 + (oki ? ri : 0)) * FU_SP + 8 * half;
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {   // 8 x 16 samples
    f16x8 av = *reinterpret_cast<const f16x8*>(pa + 16 * ks);
    f16x8 bv = *reinterpret_cast<const f16x8*>(pb + 16 * ks);
    if (!oko) av = (f16x8)(_Float16)0;
    if (!oki) bv = (f16x8)(_Float16)0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void wgrad_store(float* P, int woff, int to, int ti, int nout, int nin, int lane, const f32x16& acc) {
  const int col = lane & 31, half = lane >> 5;
  const int j = ti * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = to * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (i < nout && j < nin) P[woff + i * nin + j] = acc[r];
  }
}

// relu + f16 of one accumulator tile -> two B chunks, staged, sign bits into bits 16 it .. 16 it + 15 of `mask`
__device__ __forceinline__ void relu_stage(const f32x16& acc, int it, _Float16* stage, int row0, int h, int col, f16x8* out,
                                           uint32_t& mask) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const _Float16 v = (_Float16)fmaxf(acc[r], 0.0f);
    out[2 * it + (r >> 3)][r & 7] = v;
    mask |= ((float)v > 0.0f ? 1u : 0u) << (16 * it + r);
  }
  stage_chunk(stage, row0, out[2 * it], 2 * it, h, col);
  stage_chunk(stage, row0, out[2 * it + 1], 2 * it + 1, h, col);
}

// f16(acc) gated by the mask bits -> two B chunks of the gradient, staged as the dY operand
__device__ __forceinline__ void gate_stage(const f32x16& acc, int it, uint32_t mask, _Float16* stage, int h, int col, f16x8* out) {
#pragma unroll
  for (int r = 0; r < 16; r++) out[2 * it + (r >> 3)][r & 7] = (mask >> (16 * it + r)) & 1u ? (_Float16)acc[r] : (_Float16)0;
  stage_chunk(stage, FU_DY, out[2 * it], 2 * it, h, col);
  stage_chunk(stage, FU_DY, out[2 * it + 1], 2 * it + 1, h, col);
}

__global__ __launch_bounds__(256, 1) void ngp_mlp_bwd_fused_kernel(MlpFusedArgs a) {
  __shared__ f16x8 Wff[FW_NFRAG * 64];   // forward fragments (24 KB)
  __shared__ f16x8 Wfb[BW_NFRAG * 64];   // transposed fragments (20 KB)
  __shared__ __attribute__((aligned(16))) _Float16 stage[FU_ROWS * FU_SP];   // 85 KB: [unit rows][128 samples]
  fill_frags<false>(Wff, a.W, W1_OFF, 64, 32, FW_L1);
  fill_frags<false>(Wff, a.W, W2_OFF, 16, 64, FW_L2);
  fill_frags<false>(Wff, a.W, W3_OFF, 64, 32, FW_L3);
  fill_frags<false>(Wff, a.W, W4_OFF, 64, 64, FW_L4);
  fill_frags<true>(Wfb, a.W, W5_OFF, 16, 64, BW_L5);
  fill_frags<true>(Wfb, a.W, W4_OFF, 64, 64, BW_L4);
  fill_frags<true>(Wfb, a.W, W3_OFF, 64, 32, BW_L3);
  fill_frags<true>(Wfb, a.W, W2_OFF, 16, 64, BW_L2);
  fill_frags<true>(Wfb, a.W, W1_OFF, 64, 32, BW_L1);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const long N = a.N, cnt = ngp_count(a.N, a.n_dev), exact = ngp_exact(a.N, a.n_dev);
  const long nblk = (cnt + 127) / 128;
  const int col = wave * 32 + j;
  // weight-gradient tiles of this wave: waves 0,1: W4 (wave>>1, wave&1), W2 (0, wave), W1 (wave, 0);
  //                                     waves 2,3: W4 (wave>>1, wave&1), W5 (0, wave-2), W3 (wave-2, 0)
  f32x16 acc4 = (f32x16)0.0f, accA = (f32x16)0.0f, accB = (f32x16)0.0f;
  // inputs of a block (features, direction, loss gradient: 84 B per sample) are loaded ONE BLOCK AHEAD: with one workgroup per
  // CU nothing else hides their latency (two dependent HBM round trips per block were a third of the kernel's time)
  typedef _Float16 f16x4l __attribute__((ext_vector_type(4)));
  f16x8 xn[2];
  f16x4l gon;
  float dn[3];
  auto fetch = [&](long blk) {
    const long np = blk * 128 + col;
    const bool ok = blk < nblk && np < cnt;
    const long ns = ok ? np : 0;
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
      for (int q = 0; q < 8; q++) xn[cc][q] = ok ? a.featT[(long)frag_k(cc, h, q) * N + ns] : (_Float16)0;
    gon = (ok && np < exact) ? *reinterpret_cast<const f16x4l*>(a.dLdout + ns * 4) : (f16x4l)(_Float16)0;   // (r, g, b, d)
    dn[0] = a.dirs[ns * 3];
    dn[1] = a.dirs[ns * 3 + 1];
    dn[2] = a.dirs[ns * 3 + 2];
  };
  fetch(blockIdx.x);
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long np = blk * 128 + col;
    const bool ok = np < cnt;
    uint32_t m1 = 0, m3 = 0, m4 = 0;
    f16x8 x[2] = {xn[0], xn[1]};
    const f16x4l go = gon;
    const float dir[3] = {dn[0], dn[1], dn[2]};
    fetch(blk + gridDim.x);
    // ---- forward recompute (one 32-sample tile; same MFMA sequence as ngp_mlp_fwd_kernel); every activation is staged as the
    //      X operand of its layer's weight gradient as soon as it exists, only its sign bits stay in registers ----
    __syncthreads();   // the previous block's last weight-gradient tile has been read
    {
      f16x8 h1[4], cin[2], h3[4], h4[4];
      stage_chunk(stage, FU_XF, x[0], 0, h, col);
      stage_chunk(stage, FU_XF, x[1], 1, h, col);
#pragma unroll
      for (int it = 0; it < 2; it++) relu_stage(layer_tile<2>(Wff, FW_L1, it, lane, x), it, stage, FU_XH1, h, col, h1, m1);
      {
        const f32x16 acc = layer_tile<4>(Wff, FW_L2, 0, lane, h1);
#pragma unroll
        for (int r = 0; r < 8; r++) cin[0][r] = (_Float16)acc[r];
        cin[1] = sh_chunk(dir[0], dir[1], dir[2], h);
        stage_chunk(stage, FU_XC, cin[0], 0, h, col);
        stage_chunk(stage, FU_XC, cin[1], 1, h, col);
      }
#pragma unroll
      for (int it = 0; it < 2; it++) relu_stage(layer_tile<2>(Wff, FW_L3, it, lane, cin), it, stage, FU_XH3, h, col, h3, m3);
#pragma unroll
      for (int it = 0; it < 2; it++) relu_stage(layer_tile<4>(Wff, FW_L4, it, lane, h3), it, stage, FU_XH4, h, col, h4, m4);
    }
    // ---- backward chain; each layer's dY through the stage tile, its weight gradient contracted over the 128 samples ----
    f16x8 d5 = (f16x8)(_Float16)0;
    if (h == 0) {
      d5[0] = go[0];
      d5[1] = go[1];
      d5[2] = go[2];
    }
    stage_chunk(stage, FU_DY, d5, 0, h, col);                        // W5: dY = d5 (16 rows), X = h4
    __syncthreads();
    if (wave >= 2) wgrad_tile(stage, FU_XH4, 0, wave - 2, 16, 64, lane, accA);
    __syncthreads();
    f16x8 d4[4];
#pragma unroll
    for (int it = 0; it < 2; it++) gate_stage(layer_tile<1>(Wfb, BW_L5, it, lane, &d5), it, m4, stage, h, col, d4);
    __syncthreads();                                                 // W4: dY = d4, X = h3
    wgrad_tile(stage, FU_XH3, wave >> 1, wave & 1, 64, 64, lane, acc4);
    __syncthreads();
    f16x8 d3[4];
#pragma unroll
    for (int it = 0; it < 2; it++) gate_stage(layer_tile<4>(Wfb, BW_L4, it, lane, d4), it, m3, stage, h, col, d3);
    __syncthreads();                                                 // W3: dY = d3, X = cin (32 rows)
    if (wave >= 2) wgrad_tile(stage, FU_XC, wave - 2, 0, 64, 32, lane, accB);
    __syncthreads();
    // layer 3^T -> d(cin): only the density half (units 0..15 = registers 0..7) flows on; the density gradient joins unit 0
    f16x8 dd;
    {
      const f32x16 acc = layer_tile<4>(Wfb, BW_L3, 0, lane, d3);
#pragma unroll
      for (int r = 0; r < 8; r++) dd[r] = (_Float16)acc[r];
      if (h == 0) dd[0] = (_Float16)((float)dd[0] + (float)go[3]);
    }
    stage_chunk(stage, FU_DY, dd, 0, h, col);                        // W2: dY = dd (16 rows), X = h1
    __syncthreads();
    if (wave < 2) wgrad_tile(stage, FU_XH1, 0, wave, 16, 64, lane, accA);
    __syncthreads();
    f16x8 d1[4];
#pragma unroll
    for (int it = 0; it < 2; it++) gate_stage(layer_tile<1>(Wfb, BW_L2, it, lane, &dd), it, m1, stage, h, col, d1);
    __syncthreads();                                                 // W1: dY = d1, X = the features
    if (wave < 2) wgrad_tile(stage, FU_XF, wave, 0, 64, 32, lane, accB);
    {
      const f32x16 acc = layer_tile<4>(Wfb, BW_L1, 0, lane, d1);
      if (ok) {
#pragma unroll
        for (int r = 0; r < 16; r++) a.dLdfeatT[(long)acc_unit(0, h, r) * N + np] = (_Float16)acc[r];
      }
    }
  }
  float* P = a.partial + (long)blockIdx.x * W_TOTAL;
  wgrad_store(P, W4_OFF, wave >> 1, wave & 1, 64, 64, lane, acc4);
  if (wave < 2) {
    wgrad_store(P, W2_OFF, 0, wave, 16, 64, lane, accA);
    wgrad_store(P, W1_OFF, wave, 0, 64, 32, lane, accB);
  } else {
    wgrad_store(P, W5_OFF, 0, wave - 2, 16, 64, lane, accA);
    wgrad_store(P, W3_OFF, wave - 2, 0, 64, 32, lane, accB);
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int ns_ngp_mlp_forward(const void* weights, const void* featT, const float* dirs, void* out, void* h1T,
                                  void* cinT, void* h3T, void* h4T, long N, void* stream) {
  return ns_ngp_mlp_forward_n(weights, featT, dirs, out, h1T, cinT, h3T, h4T, N, nullptr, stream);
}

extern "C" int ns_ngp_mlp_forward_n(const void* weights, const void* featT, const float* dirs, void* out, void* h1T,
                                    void* cinT, void* h3T, void* h4T, long N, const int* n_dev, void* stream) {
  return ns_ngp_mlp_forward_m_n(weights, featT, dirs, out, h1T, cinT, h3T, h4T, nullptr, N, n_dev, stream);
}

extern "C" int ns_ngp_mlp_forward_m_n(const void* weights, const void* featT, const float* dirs, void* out, void* h1T,
                                      void* cinT, void* h3T, void* h4T, void* relu_masks, long N, const int* n_dev,
                                      void* stream) {
  NS_REQUIRE(weights && featT && dirs && out, "ns_ngp_mlp_forward: null pointer");
  NS_REQUIRE((h1T == nullptr) == (cinT == nullptr) && (h1T == nullptr) == (h3T == nullptr) &&
                 (h1T == nullptr) == (h4T == nullptr),
             "ns_ngp_mlp_forward: pass all activation buffers (training) or none (inference)");
  NS_REQUIRE(N % 2 == 0, "ns_ngp_mlp_forward: N must be even (a lane owns two adjacent samples)");
  if (N <= 0) return NS_OK;
  MlpFwdArgs a{(const _Float16*)weights, (const _Float16*)featT, dirs, (_Float16*)out, (_Float16*)h1T,
               (_Float16*)cinT, (_Float16*)h3T, (_Float16*)h4T, N, n_dev, (uint32_t*)relu_masks, nullptr};
  if (a.h1T != nullptr) hipLaunchKernelGGL(ngp_mlp_fwd_kernel<true>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, (hipStream_t)stream, a); else hipLaunchKernelGGL(ngp_mlp_fwd_kernel<false>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_fwd_kernel");
  return NS_OK;
}

// The trainer's forward pass (split form): no activation buffers, ReLU bit masks only, weights taken from the packed fragment
// table (ns_ngp_mlp_pack_fragments, once per optimiser step) -- same MFMA sequence, same outputs bit for bit.
extern "C" int ns_ngp_mlp_forward_f_n(const void* frags, const void* featT, const float* dirs, void* out, void* relu_masks, long N,
                                      const int* n_dev, void* stream) {
  NS_REQUIRE(frags && featT && dirs && out, "ns_ngp_mlp_forward_f: null pointer");
  NS_REQUIRE(N % 2 == 0, "ns_ngp_mlp_forward_f: N must be even (a lane owns two adjacent samples)");
  if (N <= 0) return NS_OK;
  MlpFwdArgs a{nullptr, (const _Float16*)featT, dirs, (_Float16*)out, nullptr, nullptr, nullptr, nullptr, N, n_dev,
               (uint32_t*)relu_masks, (const f16x8*)frags};
  if (a.h1T != nullptr) hipLaunchKernelGGL(ngp_mlp_fwd_kernel<true>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, (hipStream_t)stream, a); else hipLaunchKernelGGL(ngp_mlp_fwd_kernel<false>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_fwd_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_mlp_backward(const void* weights, const void* dLdout, const void* featT, const void* h1T,
                                   const void* cinT, const void* h3T, const void* h4T, void* dLdfeatT, void* d5T,
                                   void* d4T, void* d3T, void* ddT, void* d1T, float* partial_ws, int ksplit,
                                   float* grad_weights, long N, void* stream) {
  return ns_ngp_mlp_backward_n(weights, dLdout, featT, h1T, cinT, h3T, h4T, dLdfeatT, d5T, d4T, d3T, ddT, d1T, partial_ws, ksplit,
                               grad_weights, N, nullptr, stream);
}

static int mlp_dgrad_launch(const void* weights, const void* dLdout, const void* h1T, const void* h3T, const void* h4T,
                            void* dLdfeatT, void* d5T, void* d4T, void* d3T, void* ddT, void* d1T, long N, const int* n_dev,
                            hipStream_t st, const void* relu_masks = nullptr, const void* frags = nullptr) {
  MlpBwdArgs b{(const _Float16*)weights, (const _Float16*)dLdout, (const _Float16*)h1T, (const _Float16*)h3T,
               (const _Float16*)h4T,     (_Float16*)dLdfeatT,     (_Float16*)d5T,       (_Float16*)d4T,
               (_Float16*)d3T,           (_Float16*)ddT,          (_Float16*)d1T,       N,
               n_dev,                    (const uint32_t*)relu_masks, (const f16x8*)frags};
  if (relu_masks != nullptr && !d5T && !d4T && !d3T && !ddT && !d1T)
    hipLaunchKernelGGL((ngp_mlp_bwd_kernel<true, true>), dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, st, b);
  else if (relu_masks != nullptr)
    hipLaunchKernelGGL(ngp_mlp_bwd_kernel<true>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL(ngp_mlp_bwd_kernel<false>, dim3(ns_cdiv(N, 256 * MLP_ITERS)), dim3(256), 0, st, b);
  NS_CHECK_LAUNCH("ngp_mlp_bwd_kernel");
  return NS_OK;
}

static int mlp_wgrad_launch(const void* featT, const void* h1T, const void* cinT, const void* h3T, const void* h4T, const void* d5T,
                            const void* d4T, const void* d3T, const void* ddT, const void* d1T, float* partial_ws, int ksplit,
                            float* grad_weights, long N, const int* n_dev, hipStream_t st) {
  WgradArgs w;
  w.layer[0] = WgradLayer{(const _Float16*)d1T, (const _Float16*)featT, 64, 32, W1_OFF};
  w.layer[1] = WgradLayer{(const _Float16*)ddT, (const _Float16*)h1T, 16, 64, W2_OFF};
  w.layer[2] = WgradLayer{(const _Float16*)d3T, (const _Float16*)cinT, 64, 32, W3_OFF};
  w.layer[3] = WgradLayer{(const _Float16*)d4T, (const _Float16*)h3T, 64, 64, W4_OFF};
  w.layer[4] = WgradLayer{(const _Float16*)d5T, (const _Float16*)h4T, 16, 64, W5_OFF};
  w.partial = partial_ws;
  w.N = N;
  w.ksplit = ksplit;
  w.n_dev = n_dev;
  hipLaunchKernelGGL(ngp_mlp_wgrad_kernel, dim3(ksplit, 5), dim3(256), 0, st, w);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_kernel");
  hipLaunchKernelGGL(ngp_mlp_wgrad_reduce_kernel, dim3(W_TOTAL / 16), dim3(256), 0, st, partial_ws, ksplit,
                     grad_weights);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_reduce_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_mlp_backward_n(const void* weights, const void* dLdout, const void* featT, const void* h1T,
                                     const void* cinT, const void* h3T, const void* h4T, void* dLdfeatT, void* d5T,
                                     void* d4T, void* d3T, void* ddT, void* d1T, float* partial_ws, int ksplit,
                                     float* grad_weights, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(weights && dLdout && featT && h1T && cinT && h3T && h4T && dLdfeatT && d5T && d4T && d3T && ddT && d1T &&
                 partial_ws && grad_weights,
             "ns_ngp_mlp_backward: null pointer");
  NS_REQUIRE(ksplit >= 1 && N % 8 == 0, "ns_ngp_mlp_backward: ksplit >= 1 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  const int rc = mlp_dgrad_launch(weights, dLdout, h1T, h3T, h4T, dLdfeatT, d5T, d4T, d3T, ddT, d1T, N, n_dev, (hipStream_t)stream);
  if (rc != NS_OK) return rc;
  return mlp_wgrad_launch(featT, h1T, cinT, h3T, h4T, d5T, d4T, d3T, ddT, d1T, partial_ws, ksplit, grad_weights, N, n_dev,
                          (hipStream_t)stream);
}

// the two halves of ns_ngp_mlp_backward_n as separate entries: the weight gradients only READ what the activation backward
// wrote, so a caller may run them on another stream, next to the hash-grid backward that consumes dLdfeatT
extern "C" int ns_ngp_mlp_dgrad_n(const void* weights, const void* dLdout, const void* h1T, const void* h3T, const void* h4T,
                                  void* dLdfeatT, void* d5T, void* d4T, void* d3T, void* ddT, void* d1T, long N, const int* n_dev,
                                  void* stream) {
  NS_REQUIRE(weights && dLdout && h1T && h3T && h4T && dLdfeatT && d5T && d4T && d3T && ddT && d1T, "ns_ngp_mlp_dgrad: null pointer");
  NS_REQUIRE(N % 8 == 0, "ns_ngp_mlp_dgrad: N must be a multiple of 8");
  if (N <= 0) return NS_OK;
  return mlp_dgrad_launch(weights, dLdout, h1T, h3T, h4T, dLdfeatT, d5T, d4T, d3T, ddT, d1T, N, n_dev, (hipStream_t)stream);
}

// the activation gradients with the ReLU derivatives taken from the forward pass's bit masks (ns_ngp_mlp_forward_m_n): the saved
// activations are not read here at all (the weight gradients still read them)
extern "C" int ns_ngp_mlp_dgrad_m_n(const void* weights, const void* dLdout, const void* relu_masks, void* dLdfeatT, void* d5T,
                                    void* d4T, void* d3T, void* ddT, void* d1T, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(weights && dLdout && relu_masks && dLdfeatT, "ns_ngp_mlp_dgrad_m: null pointer");
  NS_REQUIRE((d5T == nullptr) == (d4T == nullptr) && (d5T == nullptr) == (d3T == nullptr) && (d5T == nullptr) == (ddT == nullptr) &&
                 (d5T == nullptr) == (d1T == nullptr),
             "ns_ngp_mlp_dgrad_m: pass all five gradient buffers (for ns_ngp_mlp_wgrad_n) or none (dLdfeatT only)");
  NS_REQUIRE(N % 8 == 0, "ns_ngp_mlp_dgrad_m: N must be a multiple of 8");
  if (N <= 0) return NS_OK;
  return mlp_dgrad_launch(weights, dLdout, nullptr, nullptr, nullptr, dLdfeatT, d5T, d4T, d3T, ddT, d1T, N, n_dev, (hipStream_t)stream,
                          relu_masks);
}

// the trainer's activation backward (split form): dL/dfeature only, ReLU derivatives from the bit masks, transposed weight
// fragments from the packed table
extern "C" int ns_ngp_mlp_dgrad_f_n(const void* frags, const void* dLdout, const void* relu_masks, void* dLdfeatT, long N,
                                    const int* n_dev, void* stream) {
  NS_REQUIRE(frags && dLdout && relu_masks && dLdfeatT, "ns_ngp_mlp_dgrad_f: null pointer");
  NS_REQUIRE(N % 8 == 0, "ns_ngp_mlp_dgrad_f: N must be a multiple of 8");
  if (N <= 0) return NS_OK;
  return mlp_dgrad_launch(nullptr, dLdout, nullptr, nullptr, nullptr, dLdfeatT, nullptr, nullptr, nullptr, nullptr, nullptr, N, n_dev,
                          (hipStream_t)stream, relu_masks, frags);
}

// ---------------------------------------------------------------------------------------------
// Weight gradients OFF the critical path (round 3, second form).  ngp_mlp_bwd_fused_kernel needs 145 KB of LDS per workgroup:
// one workgroup per CU, one wave per SIMD, every LDS / MFMA / VALU dependency exposed (98 us), and no other LDS-using kernel
// can start next to it.  The training step therefore splits the backward pass:
//   main stream  ngp_mlp_bwd_kernel<BITS> WITHOUT its five gradient stores: dL/dfeature only (96 B per sample), then the table
//                gradient that consumes it;
//   side stream  THIS kernel: forward and backward chains recomputed once more (72 MFMAs per 32 samples: nothing), every layer's
//                weight gradient contracted on chip as in the fused kernel -- but with 2 waves and 64 samples per workgroup (46 KB
//                of LDS: it co-resides with the scatter / accumulate workgroups of the table gradient) and the weight
//                fragments read from a packed table in global memory (44 KB, L1 / L2 resident) instead of LDS.
// 6 accumulator tiles per wave: wave 0: W4 (0,0) (0,1), W2 (0,0) (0,1), W1 (0,0) (1,0); wave 1: W4 (1,0) (1,1), W5 (0,0) (0,1), W3 (0,0) (1,0).
// Registers: 256 (two waves per SIMD, `__launch_bounds__(128, 2)`; 40 B of scratch for loop-invariant values).  It was 384 -- one
// wave owned a SIMD and nothing else could join it -- because of ADDRESSES, not data: a 64-bit per-lane address for each of the 44
// weight fragments and each of the 16 feature rows, hoisted out of the sample loop and kept live.  The fragments are read through
// a buffer resource now (SGPR descriptor + one VGPR lane offset + scalar fragment offset: layer_tile_b), the feature rows through
// uniform row bases + one 32-bit lane offset, the LDS operands of the weight-gradient tiles at one per-lane base + immediates.
// ---------------------------------------------------------------------------------------------
#define FR_SP 72    // halfs per stage row: 64 samples + 8 pad

__global__ __launch_bounds__(256) void ngp_mlp_pack_frags_kernel(const _Float16* __restrict__ W, f16x8* __restrict__ out) {
  // out = [forward fragments FW_NFRAG x 64 | transposed fragments BW_NFRAG x 64], the layout the LDS tables of the kernels have
  fill_frags<false>(out, W, W1_OFF, 64, 32, FW_L1);
  fill_frags<false>(out, W, W2_OFF, 16, 64, FW_L2);
  fill_frags<false>(out, W, W3_OFF, 64, 32, FW_L3);
  fill_frags<false>(out, W, W4_OFF, 64, 64, FW_L4);
  fill_frags<false>(out, W, W5_OFF, 16, 64, FW_L5);
  f16x8* ob = out + FW_NFRAG * 64;
  fill_frags<true>(ob, W, W5_OFF, 16, 64, BW_L5);
  fill_frags<true>(ob, W, W4_OFF, 64, 64, BW_L4);
  fill_frags<true>(ob, W, W3_OFF, 64, 32, BW_L3);
  fill_frags<true>(ob, W, W2_OFF, 16, 64, BW_L2);
  fill_frags<true>(ob, W, W1_OFF, 64, 32, BW_L1);
}


__device__ __forceinline__ void stage_chunk64(_Float16* stage, int row0, const f16x8& v, int cc, int h, int col) {
#pragma unroll
  for (int q = 0; q < 8; q++) stage[(row0 + frag_k(cc, h, q)) * FR_SP + col] = v[q];
}

__device__ __forceinline__ void wgrad_tile64(const _Float16* stage, int xrow0, int to, int ti, int nout, int nin, int lane,
                                             f32x16& acc) {
  const int col = lane & 31, half = lane >> 5;
  const int ro = to * 32 + col, ri = ti * 32 + col;
  const bool oko = ro < nout, oki = ri < nin;
  // (rows beyond nout / nin are read all the same -- they exist in the stage -- and zeroed below: the address then is ONE per-lane
  //  base + a compile-time offset for every tile, instead of a select per tile that the compiler keeps live across the loop)
  const _Float16* pa = stage + (FU_DY + ro) * FR_SP + 8 * half;
  const _Float16* pb = stage + (xrow0 + ri) * FR_SP + 8 * half;
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {   // 4 x 16 samples
    f16x8 av = *reinterpret_cast<const f16x8*>(pa + 16 * ks);
    f16x8 bv = *reinterpret_cast<const f16x8*>(pb + 16 * ks);
    if (!oko) av = (f16x8)(_Float16)0;
    if (!oki) bv = (f16x8)(_Float16)0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(128, 2) void ngp_mlp_wgrad_recompute_kernel(MlpWgradArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 stage[FU_ROWS * FR_SP];   // 46 KB: [unit rows][64 samples]
  const __amdgpu_buffer_rsrc_t Wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16x8*>(a.frags), 0, (FW_NFRAG + BW_NFRAG) * 1024, 0x00027000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const long N = a.N, cnt = ngp_count(a.N, a.n_dev), exact = ngp_exact(a.N, a.n_dev);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const long nblk = (cnt + 63) / 64;
  const int col = wave * 32 + j;
  f32x16 t0 = (f32x16)0.0f, t1 = (f32x16)0.0f, t2 = (f32x16)0.0f, t3 = (f32x16)0.0f, t4 = (f32x16)0.0f, t5 = (f32x16)0.0f;
  typedef _Float16 f16x4l __attribute__((ext_vector_type(4)));
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long np = blk * 64 + col;
    const bool ok = np < cnt;
    const long ns = ok ? np : 0;
    uint32_t m1 = 0, m3 = 0, m4 = 0;
    const uint32_t xoff = lane_bytes(h, N, ns);
    __syncthreads();   // the previous block's last weight-gradient tiles have been read
    {
      f16x8 x[2], h1[4], cin[2], h3[4], h4[4];
#pragma unroll
      for (int cc = 0; cc < 2; cc++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {   // (uniform row base + one 32-bit lane offset: see lane_bytes)
          const _Float16 v = *reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(a.featT + (long)ufrag(cc, q) * N) + xoff);
          x[cc][q] = ok ? v : (_Float16)0;
        }
        stage_chunk64(stage, FU_XF, x[cc], cc, h, col);
      }
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const f32x16 acc = layer_tile_b<2>(Wrs, FW_L1, it, lane16, x);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const _Float16 v = (_Float16)fmaxf(acc[r], 0.0f);
          h1[2 * it + (r >> 3)][r & 7] = v;
          m1 |= ((float)v > 0.0f ? 1u : 0u) << (16 * it + r);
        }
        stage_chunk64(stage, FU_XH1, h1[2 * it], 2 * it, h, col);
        stage_chunk64(stage, FU_XH1, h1[2 * it + 1], 2 * it + 1, h, col);
      }
      {
        const f32x16 acc = layer_tile_b<4>(Wrs, FW_L2, 0, lane16, h1);
#pragma unroll
        for (int r = 0; r < 8; r++) cin[0][r] = (_Float16)acc[r];
        cin[1] = sh_chunk(a.dirs[ns * 3], a.dirs[ns * 3 + 1], a.dirs[ns * 3 + 2], h);
        stage_chunk64(stage, FU_XC, cin[0], 0, h, col);
        stage_chunk64(stage, FU_XC, cin[1], 1, h, col);
      }
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const f32x16 acc = layer_tile_b<2>(Wrs, FW_L3, it, lane16, cin);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const _Float16 v = (_Float16)fmaxf(acc[r], 0.0f);
          h3[2 * it + (r >> 3)][r & 7] = v;
          m3 |= ((float)v > 0.0f ? 1u : 0u) << (16 * it + r);
        }
        stage_chunk64(stage, FU_XH3, h3[2 * it], 2 * it, h, col);
        stage_chunk64(stage, FU_XH3, h3[2 * it + 1], 2 * it + 1, h, col);
      }
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const f32x16 acc = layer_tile_b<4>(Wrs, FW_L4, it, lane16, h3);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const _Float16 v = (_Float16)fmaxf(acc[r], 0.0f);
          h4[2 * it + (r >> 3)][r & 7] = v;
          m4 |= ((float)v > 0.0f ? 1u : 0u) << (16 * it + r);
        }
        stage_chunk64(stage, FU_XH4, h4[2 * it], 2 * it, h, col);
        stage_chunk64(stage, FU_XH4, h4[2 * it + 1], 2 * it + 1, h, col);
      }
    }
    f16x4l go = (f16x4l)(_Float16)0;
    if (ok && np < exact) go = *reinterpret_cast<const f16x4l*>(a.dLdout + np * 4);   // (r, g, b, d)
    f16x8 d5 = (f16x8)(_Float16)0;
    if (h == 0) {
      d5[0] = go[0];
      d5[1] = go[1];
      d5[2] = go[2];
    }
    stage_chunk64(stage, FU_DY, d5, 0, h, col);                        // W5: dY = d5 (16 rows), X = h4
    __syncthreads();
    if (wave == 1) {
      wgrad_tile64(stage, FU_XH4, 0, 0, 16, 64, lane, t2);
      wgrad_tile64(stage, FU_XH4, 0, 1, 16, 64, lane, t3);
    }
    __syncthreads();
    f16x8 d4[4];
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 acc = layer_tile_b<1>(Wrs, FW_NFRAG + BW_L5, it, lane16, &d5);
#pragma unroll
      for (int r = 0; r < 16; r++) d4[2 * it + (r >> 3)][r & 7] = (m4 >> (16 * it + r)) & 1u ? (_Float16)acc[r] : (_Float16)0;
      stage_chunk64(stage, FU_DY, d4[2 * it], 2 * it, h, col);
      stage_chunk64(stage, FU_DY, d4[2 * it + 1], 2 * it + 1, h, col);
    }
    __syncthreads();                                                   // W4: dY = d4, X = h3
    wgrad_tile64(stage, FU_XH3, wave, 0, 64, 64, lane, t0);
    wgrad_tile64(stage, FU_XH3, wave, 1, 64, 64, lane, t1);
    __syncthreads();
    f16x8 d3[4];
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 acc = layer_tile_b<4>(Wrs, FW_NFRAG + BW_L4, it, lane16, d4);
#pragma unroll
      for (int r = 0; r < 16; r++) d3[2 * it + (r >> 3)][r & 7] = (m3 >> (16 * it + r)) & 1u ? (_Float16)acc[r] : (_Float16)0;
      stage_chunk64(stage, FU_DY, d3[2 * it], 2 * it, h, col);
      stage_chunk64(stage, FU_DY, d3[2 * it + 1], 2 * it + 1, h, col);
    }
    __syncthreads();                                                   // W3: dY = d3, X = cin (32 rows)
    if (wave == 1) {
      wgrad_tile64(stage, FU_XC, 0, 0, 64, 32, lane, t4);
      wgrad_tile64(stage, FU_XC, 1, 0, 64, 32, lane, t5);
    }
    __syncthreads();
    f16x8 dd;
    {
      const f32x16 acc = layer_tile_b<4>(Wrs, FW_NFRAG + BW_L3, 0, lane16, d3);
#pragma unroll
      for (int r = 0; r < 8; r++) dd[r] = (_Float16)acc[r];
      if (h == 0) dd[0] = (_Float16)((float)dd[0] + (float)go[3]);
    }
    stage_chunk64(stage, FU_DY, dd, 0, h, col);                        // W2: dY = dd (16 rows), X = h1
    __syncthreads();
    if (wave == 0) {
      wgrad_tile64(stage, FU_XH1, 0, 0, 16, 64, lane, t2);
      wgrad_tile64(stage, FU_XH1, 0, 1, 16, 64, lane, t3);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const f32x16 acc = layer_tile_b<1>(Wrs, FW_NFRAG + BW_L2, it, lane16, &dd);
      f16x8 lo8, hi8;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        lo8[r] = (m1 >> (16 * it + r)) & 1u ? (_Float16)acc[r] : (_Float16)0;
        hi8[r] = (m1 >> (16 * it + 8 + r)) & 1u ? (_Float16)acc[8 + r] : (_Float16)0;
      }
      stage_chunk64(stage, FU_DY, lo8, 2 * it, h, col);
      stage_chunk64(stage, FU_DY, hi8, 2 * it + 1, h, col);
    }
    __syncthreads();                                                   // W1: dY = d1, X = the features
    if (wave == 0) {
      wgrad_tile64(stage, FU_XF, 0, 0, 64, 32, lane, t4);
      wgrad_tile64(stage, FU_XF, 1, 0, 64, 32, lane, t5);
    }
  }
  float* P = a.partial + (long)blockIdx.x * W_TOTAL;
  wgrad_store(P, W4_OFF, wave, 0, 64, 64, lane, t0);
  wgrad_store(P, W4_OFF, wave, 1, 64, 64, lane, t1);
  if (wave == 0) {
    wgrad_store(P, W2_OFF, 0, 0, 16, 64, lane, t2);
    wgrad_store(P, W2_OFF, 0, 1, 16, 64, lane, t3);
    wgrad_store(P, W1_OFF, 0, 0, 64, 32, lane, t4);
    wgrad_store(P, W1_OFF, 1, 0, 64, 32, lane, t5);
  } else {
    wgrad_store(P, W5_OFF, 0, 0, 16, 64, lane, t2);
    wgrad_store(P, W5_OFF, 0, 1, 16, 64, lane, t3);
    wgrad_store(P, W3_OFF, 0, 0, 64, 32, lane, t4);
    wgrad_store(P, W3_OFF, 1, 0, 64, 32, lane, t5);
  }
}

extern "C" size_t ns_ngp_mlp_fragment_table_bytes(void) { return (size_t)(FW_NFRAG + BW_NFRAG) * 64 * sizeof(f16x8); }

// packed MFMA fragment table of the current weights (forward + transposed), for ns_ngp_mlp_wgrad_recompute_n
extern "C" int ns_ngp_mlp_pack_fragments(const void* weights, void* frags, void* stream) {
  NS_REQUIRE(weights && frags && ((uintptr_t)frags % 16) == 0, "ns_ngp_mlp_pack_fragments: null / unaligned pointer");
  hipLaunchKernelGGL(ngp_mlp_pack_frags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const _Float16*)weights, (f16x8*)frags);
  NS_CHECK_LAUNCH("ngp_mlp_pack_frags_kernel");
  return NS_OK;
}

// weight gradients alone, forward and backward chains recomputed on chip from the features and the loss gradient: ADDS to
// grad_weights; partial_ws: wgs * 10240 floats
// ---------------------------------------------------------------------------------------------
// The MLP's optimiser step in ONE launch (round 4): slab reduce + Adam + fragment tables.  The side stream of the training step
// ran  weight gradients -> ngp_mlp_wgrad_reduce_kernel -> ngp_adam_kernel -> ngp_mlp_pack_frags_kernel : three launches of
// microseconds of work each (11 + 13 + 21 us inside the pipeline, most of it waiting for a slot) at the tail of a chain that must
// end before the next step's forward pass.  Every output depends on ONE weight: a lane owns weight i, adds its slabs in the order
// of the reduce kernel (16 interleaved partial sums, then their sum: the same bits), applies adam_apply() (the one definition,
// ngp_adam.h), writes the f32 master and the f16 copy, and drops the f16 value into its two places in the fragment tables --
// forward table: A = W [nout][nin], transposed table: A = W^T -- by inverting fill_frags():
//   element (row, k) of A lives in fragment (row / 32) nchunk + k / 16, lane (row % 32) + 32 h, element q,
//   with k % 16 = 4 h + (q & 3) + 8 (q >> 2)   <=>   h = bit 2 of k, q = (k & 3) | (bit 3 of k) << 2.
// Rows of a fragment beyond a layer's size are zeros written once by ngp_mlp_pack_frags_kernel and never touched here.
// ---------------------------------------------------------------------------------------------
struct MlpStepArgs {
  const float* partial;   // [slabs][W_TOTAL]
  int slabs;
  float *grad, *master, *m1, *m2;
  _Float16 *hp, *frags;   // f16 weights; fragment tables [FW_NFRAG + BW_NFRAG][64][8]
  float c1, c2, lr, beta1, beta2, eps, l2, inv_grad_scale;
  const int* ctl;
};

__device__ __forceinline__ void frag_place(_Float16* frags, int first, int nchunk, int row, int k, _Float16 v) {
  const int f = first + (row >> 5) * nchunk + (k >> 4);
  const int h = (k >> 2) & 1, q = (k & 3) | ((k >> 3) & 1) << 2;
  frags[((long)f * 64 + (row & 31) + 32 * h) * 8 + q] = v;
}

__global__ __launch_bounds__(256) void ngp_mlp_step_kernel(MlpStepArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W_TOTAL) return;
  // ---- the slabs, in ngp_mlp_wgrad_reduce_kernel's order: 16 interleaved partial sums (slab k belongs to sum k % 16), then
  //      their sum.  Up to 64 slabs (the trainer's: 64 weight-gradient workgroups) ALL loads leave before the first add
  //      (the loop below is 16 dependent load rounds).  Inside the pipeline the kernel takes 37-38 us either way: its 40
  //      workgroups wait for CU slots behind the accumulate pass of the table gradient, not for their loads ----
  float t = 0.0f;
  if (a.slabs <= 64) {
    float v[4][16];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int grp = 0; grp < 16; grp++) {
        const int k = grp + 16 * u;
        v[u][grp] = k < a.slabs ? a.partial[(long)k * W_TOTAL + i] : 0.0f;
      }
#pragma unroll
    for (int grp = 0; grp < 16; grp++) {
      float s = 0.0f;
#pragma unroll
      for (int u = 0; u < 4; u++) s += v[u][grp];
      t += s;
    }
  } else {
    for (int grp = 0; grp < 16; grp++) {
      float s = 0.0f;
      for (int k0 = grp; k0 < a.slabs; k0 += 16 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int k = k0 + 16 * u;
          v[u] = k < a.slabs ? a.partial[(long)k * W_TOTAL + i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
      }
      t += s;
    }
  }
  // ---- Adam, as ngp_adam_kernel ----
  const float c1 = a.ctl ? __int_as_float(a.ctl[NS_CTL_C1]) : a.c1, c2 = a.ctl ? __int_as_float(a.ctl[NS_CTL_C2]) : a.c2;
  const float g = (a.grad[i] + t) * a.inv_grad_scale;
  a.grad[i] = 0.0f;
  float p = a.master[i];
  if (!(g == 0.0f && a.l2 == 0.0f)) {
    float m1 = a.m1[i], m2 = a.m2[i];
    p = adam_apply(p, g, a.l2, m1, m2, c1, c2, a.lr, a.beta1, a.beta2, a.eps);
    a.m1[i] = m1;
    a.m2[i] = m2;
    a.master[i] = p;
  }
  const _Float16 hv = (_Float16)p;
  a.hp[i] = hv;
  // ---- the weight's two places in the fragment tables ----
  int off, nin, fw, bw, nout;
  if (i < W2_OFF)      { off = W1_OFF; nout = 64; nin = 32; fw = FW_L1; bw = BW_L1; }
  else if (i < W3_OFF) { off = W2_OFF; nout = 16; nin = 64; fw = FW_L2; bw = BW_L2; }
  else if (i < W4_OFF) { off = W3_OFF; nout = 64; nin = 32; fw = FW_L3; bw = BW_L3; }
  else if (i < W5_OFF) { off = W4_OFF; nout = 64; nin = 64; fw = FW_L4; bw = BW_L4; }
  else                 { off = W5_OFF; nout = 16; nin = 64; fw = FW_L5; bw = BW_L5; }
  const int o = (i - off) / nin, k = (i - off) - o * nin;
  frag_place(a.frags, fw, nin / 16, o, k, hv);                              // A = W:   row o, column k
  frag_place(a.frags + (long)FW_NFRAG * 64 * 8, bw, nout / 16, k, o, hv);   // A = W^T: row k, column o
}

// slabs the weight-gradient launch of ns_ngp_mlp_wgrad_recompute_n / _partials_n fills for a workspace of `wgs` slabs
static int wgrad_slabs(int wgs, long N, bool& staged) {
  static const bool st = [] { const char* e = ns_variant_env("NS_NGP_WGRAD"); return e != nullptr && e[0] == 's'; }();
  // The matrix-core kernel fetches WHOLE 32-sample tiles of featT [32, N] (values past the live count are masked, the loads
  // are not): with a row stride that is not a multiple of the tile, the last tile of the last row would read past the array
  // (ADVICE r04).  Such budgets (the trainer's 2^18 never is one) take the staged kernel, which loads per 8 samples.
  staged = st || (N % 32 != 0);
  if (staged) return wgs;
  static const int cus = [] { const char* e = ns_variant_env("NS_NGP_WGRAD_WGS"); return e ? atoi(e) : 64; }();
  const long tiles4 = (N / 32 + 3) / 4;
  return (int)std::max(1L, std::min((long)std::min(wgs, cus), tiles4));
}

extern "C" int ns_ngp_mlp_wgrad_slabs(int wgs, long N) {
  bool staged;
  return wgrad_slabs(wgs, N, staged);
}

static int wgrad_partials_launch(const void* frags, const void* featT, const float* dirs, const void* dLdout, float* partial_ws,
                                 int wgs, long N, const int* n_dev, void* stream, int& slabs) {
  MlpWgradArgs a{(const f16x8*)frags, (const _Float16*)featT, dirs, (const _Float16*)dLdout, partial_ws, N, n_dev};
  bool staged;
  slabs = wgrad_slabs(wgs, N, staged);
  if (staged) {
    hipLaunchKernelGGL(ngp_mlp_wgrad_recompute_kernel, dim3(slabs), dim3(128), 0, (hipStream_t)stream, a);
    NS_CHECK_LAUNCH("ngp_mlp_wgrad_recompute_kernel");
    return NS_OK;
  }
  // 4-wave workgroups, every wave holding all twelve accumulator tiles: a workgroup takes a whole CU's registers.  64 of them
  // (a quarter of the chip for ~4 x as long, still well inside the step) instead of one per CU: in the step this kernel runs
  // next to the table gradient's scatter, which then keeps 192 CUs to itself -- stand-alone step unchanged (0.275 ms), pipeline
  // 129-131 -> 134-137 frames/s (32: 133, 96: 136, 128: 131).  NS_NGP_WGRAD_WGS overrides.
  NS_REQUIRE(((uintptr_t)featT % 8) == 0 && ((uintptr_t)dLdout % 8) == 0 && ((uintptr_t)partial_ws % 16) == 0,
             "ns_ngp_mlp_wgrad_recompute: featT / dLdout must be 8-byte, partial_ws 16-byte aligned");
  return ngp_mlp_wgrad_tr_launch(a, slabs, (hipStream_t)stream);
}

extern "C" int ns_ngp_mlp_wgrad_recompute_n(const void* frags, const void* featT, const float* dirs, const void* dLdout,
                                            float* partial_ws, int wgs, float* grad_weights, long N, const int* n_dev,
                                            void* stream) {
  NS_REQUIRE(frags && featT && dirs && dLdout && partial_ws && grad_weights, "ns_ngp_mlp_wgrad_recompute: null pointer");
  NS_REQUIRE(wgs >= 1 && wgs <= 65535 && N % 8 == 0, "ns_ngp_mlp_wgrad_recompute: 1 <= wgs <= 65535 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  int slabs = 0;
  const int rc = wgrad_partials_launch(frags, featT, dirs, dLdout, partial_ws, wgs, N, n_dev, stream, slabs);
  if (rc != NS_OK) return rc;
  hipLaunchKernelGGL(ngp_mlp_wgrad_reduce_kernel, dim3(W_TOTAL / 16), dim3(256), 0, (hipStream_t)stream, partial_ws, slabs, grad_weights);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_reduce_kernel");
  return NS_OK;
}

// the weight-gradient launch alone: ns_ngp_mlp_wgrad_slabs(wgs, N) slabs of partial_ws are left for ns_ngp_mlp_reduce /
// ns_ngp_mlp_step_fused
extern "C" int ns_ngp_mlp_wgrad_partials_n(const void* frags, const void* featT, const float* dirs, const void* dLdout,
                                           float* partial_ws, int wgs, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(frags && featT && dirs && dLdout && partial_ws, "ns_ngp_mlp_wgrad_partials: null pointer");
  NS_REQUIRE(wgs >= 1 && wgs <= 65535 && N % 8 == 0, "ns_ngp_mlp_wgrad_partials: 1 <= wgs <= 65535 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  int slabs = 0;
  return wgrad_partials_launch(frags, featT, dirs, dLdout, partial_ws, wgs, N, n_dev, stream, slabs);
}

extern "C" int ns_ngp_mlp_reduce(const float* partial_ws, int slabs, float* grad_weights, void* stream) {
  NS_REQUIRE(partial_ws && grad_weights && slabs >= 1, "ns_ngp_mlp_reduce: null pointer or no slabs");
  hipLaunchKernelGGL(ngp_mlp_wgrad_reduce_kernel, dim3(W_TOTAL / 16), dim3(256), 0, (hipStream_t)stream, partial_ws, slabs, grad_weights);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_reduce_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_mlp_step_fused(const float* partial_ws, int slabs, float* grad_weights, float* master, void* half_params,
                                     float* m1, float* m2, void* frags, int step, float lr, float beta1, float beta2, float eps,
                                     float l2, float grad_scale, const int* ctl, void* stream) {
  NS_REQUIRE(partial_ws && grad_weights && master && half_params && m1 && m2 && frags, "ns_ngp_mlp_step_fused: null pointer");
  NS_REQUIRE(slabs >= 1 && (ctl || step >= 1) && grad_scale > 0.0f, "ns_ngp_mlp_step_fused: slabs >= 1, step >= 1 (or ctl), grad_scale > 0");
  MlpStepArgs a;
  a.partial = partial_ws;
  a.slabs = slabs;
  a.grad = grad_weights;
  a.master = master;
  a.m1 = m1;
  a.m2 = m2;
  a.hp = (_Float16*)half_params;
  a.frags = (_Float16*)frags;
  a.c1 = 1.0f - powf(beta1, (float)(step < 1 ? 1 : step));
  a.c2 = 1.0f - powf(beta2, (float)(step < 1 ? 1 : step));
  a.lr = lr;
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.l2 = l2;
  a.inv_grad_scale = 1.0f / grad_scale;
  a.ctl = ctl;
  hipLaunchKernelGGL(ngp_mlp_step_kernel, dim3(W_TOTAL / 256), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_step_kernel");
  return NS_OK;
}

// activation gradients AND weight gradients from the features alone (forward recomputed on chip): writes dLdfeatT [32,N]
// and ADDS the weight gradients to grad_weights; partial_ws holds `wgs` slabs of W_TOTAL floats (wgs = launched workgroups)
extern "C" int ns_ngp_mlp_backward_fused_n(const void* weights, const void* featT, const float* dirs, const void* dLdout,
                                           void* dLdfeatT, float* partial_ws, int wgs, float* grad_weights, long N,
                                           const int* n_dev, void* stream) {
  NS_REQUIRE(weights && featT && dirs && dLdout && dLdfeatT && partial_ws && grad_weights, "ns_ngp_mlp_backward_fused: null pointer");
  NS_REQUIRE(wgs >= 1 && wgs <= 65535 && N % 8 == 0, "ns_ngp_mlp_backward_fused: 1 <= wgs <= 65535 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  MlpFusedArgs a{(const _Float16*)weights, (const _Float16*)featT, dirs, (const _Float16*)dLdout, (_Float16*)dLdfeatT, partial_ws, N, n_dev};
  hipLaunchKernelGGL(ngp_mlp_bwd_fused_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_mlp_bwd_fused_kernel");
  hipLaunchKernelGGL(ngp_mlp_wgrad_reduce_kernel, dim3(W_TOTAL / 16), dim3(256), 0, (hipStream_t)stream, partial_ws, wgs, grad_weights);
  NS_CHECK_LAUNCH("ngp_mlp_wgrad_reduce_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_mlp_wgrad_n(const void* featT, const void* h1T, const void* cinT, const void* h3T, const void* h4T,
                                  const void* d5T, const void* d4T, const void* d3T, const void* ddT, const void* d1T,
                                  float* partial_ws, int ksplit, float* grad_weights, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(featT && h1T && cinT && h3T && h4T && d5T && d4T && d3T && ddT && d1T && partial_ws && grad_weights,
             "ns_ngp_mlp_wgrad: null pointer");
  NS_REQUIRE(ksplit >= 1 && N % 8 == 0, "ns_ngp_mlp_wgrad: ksplit >= 1 and N a multiple of 8 are required");
  if (N <= 0) return NS_OK;
  return mlp_wgrad_launch(featT, h1T, cinT, h3T, h4T, d5T, d4T, d3T, ddT, d1T, partial_ws, ksplit, grad_weights, N, n_dev,
                          (hipStream_t)stream);
}
