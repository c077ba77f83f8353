// ngp_mlp_common.h -- layouts shared by the MLP kernels (csrc/ngp_mlp.hip, csrc/ngp_mlp_wgrad.hip): packed weight offsets, the
// direction encoding, the MFMA fragment conventions of v_mfma_f32_32x32x16_f16 and the sample-count helpers.
#pragma once
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
#pragma clang fp contract(off)          // (see sh_term)
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

// The MFMA kernels want, per lane half h, the 8 terms {(q & 3) + 8 (q >> 2) + 4 h}.  Written as `h ? sh[lo + 4] : sh[lo]` over
// the array above the compiler turned the select into an INDEXED load: the 16 terms went to scratch (or to 16 KB of LDS per
// workgroup, where the private array could be promoted) and came back with a per-lane offset.  One term by compile-time index
// instead (the switch folds away after unrolling, the shared products are CSE'd): no array at all.
__device__ __forceinline__ float sh_term(int k, float x, float y, float z) {
  // No contraction: with the default (fast) mode whether `c * (z * z) - d` becomes an fma depends on what ELSE the kernel around
  // this inline function does with z * z -- round 6 changed the ReLU of the forward kernel and 3 of 262 144 outputs moved by an
  // ulp, because the direction encoding had been compiled differently.  Separately rounded products are what the oracle
  // computes (-ffp-contract=off) and what every kernel that includes this header now computes, whatever surrounds the call.
#pragma clang fp contract(off)
  switch (k) {
    case 0: return 0.28209479177387814f;
    case 1: return -0.48860251190291987f * y;
    case 2: return 0.48860251190291987f * z;
    case 3: return -0.48860251190291987f * x;
    case 4: return 1.0925484305920792f * (x * y);
    case 5: return -1.0925484305920792f * (y * z);
    case 6: return 0.94617469575755997f * (z * z) - 0.31539156525251999f;
    case 7: return -1.0925484305920792f * (x * z);
    case 8: return 0.54627421529603959f * (x * x) - 0.54627421529603959f * (y * y);
    case 9: return 0.59004358992664352f * y * (-3.0f * (x * x) + (y * y));
    case 10: return 2.8906114426405538f * (x * y) * z;
    case 11: return 0.45704579946446572f * y * (1.0f - 5.0f * (z * z));
    case 12: return 0.3731763325901154f * z * (5.0f * (z * z) - 3.0f);
    case 13: return 0.45704579946446572f * x * (1.0f - 5.0f * (z * z));
    case 14: return 1.4453057213202769f * z * ((x * x) - (y * y));
    default: return 0.59004358992664352f * x * (-(x * x) + 3.0f * (y * y));
  }
}
typedef _Float16 sh_f16x8 __attribute__((ext_vector_type(8)));
// chunk 1 of the colour MLP's input for lane half h (element q = SH term (q & 3) + 8 (q >> 2) + 4 h)
__device__ __forceinline__ sh_f16x8 sh_chunk(float x, float y, float z, int h) {
  sh_f16x8 c;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int lo = (q & 3) + 8 * (q >> 2);
    const float a = sh_term(lo, x, y, z), b = sh_term(lo + 4, x, y, z);
    c[q] = (_Float16)(h ? b : a);
  }
  return c;
}

// ---------------------------------------------------------------------------------------------
// MFMA register chain.
//
// Every layer is computed TRANSPOSED: H^T[unit][sample] = W[unit][k] * X^T[k][sample] with
// v_mfma_f32_32x32x16_f16 (A = 32 weight rows x 16 k, B = 16 k x 32 samples).  The accumulator layout of
// that instruction gives lane (j = lane & 31, h = lane >> 5) the 16 units {4h + (r & 3) + 8 (r >> 2)} of
// sample j -- and the B operand of the next layer wants, per 16-wide k chunk, 8 k values of sample j from
// each half-wave.  A matrix product does not care in which order k is summed, so the k order of every
// chunk is DEFINED as "what the accumulator already holds": chunk 0 of a 32-unit tile = registers r 0..7
// (units 4h..4h+3, 8+4h..8+4h+3), chunk 1 = r 8..15; the weight fragments are gathered into LDS in that
// same order once per workgroup.  A layer's output becomes the next layer's input by ReLU + cvt_f16 in
// place: no LDS round trip, no shuffles, no transposes between the five layers.
//
// A wave handles 64 samples per iteration as two column tiles: tile 0 = even samples, tile 1 = odd ones, so
// lane j owns samples (2j, 2j+1) and every unit-major load / store is one dword per lane = 128 contiguous
// bytes per half-wave.  The kernels are HBM-bound on the saved activations (forward ~0.5 KB, backward
// ~0.9 KB per sample); the 40 MFMAs per 32 samples are a few microseconds per 2^18 samples.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define MLP_ITERS 2  // 64-sample iterations per wave

// fragment tables: (weight offset, rows of A, columns of A (= K), first fragment); A = W (forward) or W^T (backward)
// forward : L1 W1[64][32]  L2 W2[16][64]  L3 W3[64][32]  L4 W4[64][64]  L5 W5[16][64]
#define FW_L1 0
#define FW_L2 4
#define FW_L3 8
#define FW_L4 12
#define FW_L5 20
#define FW_NFRAG 24
// backward: L5^T [64][16]  L4^T [64][64]  L3^T [32][64]  L2^T [64][16]  L1^T [32][64]
#define BW_L5 0
#define BW_L4 2
#define BW_L3 10
#define BW_L2 14
#define BW_L1 16
#define BW_NFRAG 20

// k (column of A) that lane half h supplies as element q of chunk cc
__device__ __forceinline__ int frag_k(int cc, int h, int q) { return 16 * cc + 4 * h + (q & 3) + 8 * (q >> 2); }
// unit (row of D) that lane half h holds in accumulator register r of row tile it
__device__ __forceinline__ int acc_unit(int it, int h, int r) { return 32 * it + 4 * h + (r & 3) + 8 * (r >> 2); }

// Unit-major addressing [unit][sample], split so that the compiler keeps ONE 32-bit lane offset for every row of every
// tensor: element (unit, sample) with unit = u + 4 h (u = the wave-uniform part of acc_unit / frag_k) lives at
//   (base + u N) [uniform: scalar registers]  +  2 (4 h N + sample) bytes [per lane: one VGPR]
// -> `global_load/store_dword v, v_off, s[base:base+1]`.  Written as 64-bit per-row addresses the same accesses cost two
// address VGPRs and a 64-bit multiply-add per row and pushed the backward kernel to 240 VGPRs.
__device__ __forceinline__ int urow(int it, int r) { return 32 * it + (r & 3) + 8 * (r >> 2); }
__device__ __forceinline__ int ufrag(int cc, int q) { return 16 * cc + (q & 3) + 8 * (q >> 2); }
__device__ __forceinline__ uint32_t lane_bytes(int h, long N, long np) { return (uint32_t)((4 * (long)h * N + np) * 2); }
__device__ __forceinline__ uint32_t* um_at(_Float16* row, uint32_t boff) {
  return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(row) + boff);
}
__device__ __forceinline__ const uint32_t* um_at(const _Float16* row, uint32_t boff) {
  return reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(row) + boff);
}

// gather the fragments of one layer into LDS: frag (it, cc) -> Wf[(first + it * nchunk + cc) * 64 + lane]
template <bool TRANSPOSED>
__device__ __forceinline__ void fill_frags(f16x8* Wf, const _Float16* __restrict__ W, int woff, int nout, int nin,
                                           int first) {
  const int rows = TRANSPOSED ? nin : nout, cols = TRANSPOSED ? nout : nin;  // of A
  const int ntile = (rows + 31) / 32, nchunk = cols / 16;
  for (int e = threadIdx.x; e < ntile * nchunk * 64; e += 256) {
    const int lane = e & 63, f = e >> 6, it = f / nchunk, cc = f % nchunk;
    const int row = 32 * it + (lane & 31), h = lane >> 5;
    f16x8 v;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int k = frag_k(cc, h, q);
      v[q] = row < rows ? (TRANSPOSED ? W[woff + k * nin + row] : W[woff + row * nin + k]) : (_Float16)0;
    }
    Wf[(first + f) * 64 + lane] = v;
  }
}

// one row tile of one layer: acc = sum over chunks A(it, cc) * B(cc)
template <int NCHUNK>
__device__ __forceinline__ f32x16 layer_tile(const f16x8* Wf, int first, int it, int lane, const f16x8* bin) {
  f32x16 acc = (f32x16)0.0f;
#pragma unroll
  for (int cc = 0; cc < NCHUNK; cc++)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[(first + it * NCHUNK + cc) * 64 + lane], bin[cc], acc, 0, 0, 0);
  return acc;
}

// the same with the fragments in GLOBAL memory (the packed table, L1 / L2 resident), read through a BUFFER resource: the address of
// fragment f is (SGPR resource) + (one VGPR: 16 x lane) + (scalar f x 1 KB).  Written as Wf[f * 64 + lane] the compiler built a
// 64-bit per-lane address for every fragment whose offset does not fit the 12-bit immediate, hoisted all of them out of the sample
// loop and kept them live: ~80 of the weight-gradient kernel's registers.
typedef uint32_t mlp_u32x4 __attribute__((ext_vector_type(4)));
template <int NCHUNK>
__device__ __forceinline__ f32x16 layer_tile_b(__amdgpu_buffer_rsrc_t Wf, int first, int it, uint32_t lane16, const f16x8* bin) {
  f32x16 acc = (f32x16)0.0f;
#pragma unroll
  for (int cc = 0; cc < NCHUNK; cc++) {
    const mlp_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(Wf, lane16, (first + it * NCHUNK + cc) * 1024, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), bin[cc], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
  const f16x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ f16x2 unpack2(uint32_t w) { return __builtin_bit_cast(f16x2, w); }


// samples to process: the by-value N, or the device count rounded up to 8 (the tail slots carry zero gradients)
__device__ __forceinline__ long ngp_count(long N, const int* n_dev) {
  if (n_dev == nullptr) return N;
  const long c = ((long)*n_dev + 7) & ~7L;
  return c < N ? c : N;
}

// the exact device count (ngp_count rounds it up to 8): the up to 7 slots between the two carry ZERO upstream gradient, whatever
// the loss-gradient buffer holds there -- the kernels that read dL/dout mask them, so nobody has to clear that buffer per step
__device__ __forceinline__ long ngp_exact(long N, const int* n_dev) {
  if (n_dev == nullptr) return N;
  const long c = (long)*n_dev;
  return c < N ? c : N;
}


struct MlpWgradArgs {
  const f16x8* frags;      // ngp_mlp_pack_frags_kernel's table
  const _Float16* featT;
  const float* dirs;
  const _Float16* dLdout;
  float* partial;          // [gridDim.x][W_TOTAL]
  long N;
  const int* n_dev;
};

// defined in ngp_mlp_wgrad.hip (compiled with the VGPR form of the MFMA builtins): launches the register-only weight-gradient kernel
int ngp_mlp_wgrad_tr_launch(const MlpWgradArgs& a, int wgs, hipStream_t stream);
