// corr_volume.hip -- all-pairs correlation volume + 4-level pyramid in ONE kernel (gfx950, MFMA).
//
// Replaces CorrBlock.__init__ of the reference (networks/modules/corr.py:23-38, 63-72):
//   corr = (fmap1/4)^T (fmap2/4)            torch.matmul -> [HW, HW] per edge, f16 out
//   3 x F.avg_pool2d(corr, 2)               three more full passes over the volume
// i.e. the volume is written once and then re-read 1.33x and the pooled levels written by separate
// launches.  The job is OUTPUT-WRITE bound (K = 128: 96 flop per written byte, far under the f16
// MFMA ridge), so what matters is writing every byte exactly once AND in whole 128-byte lines
// (a first version that stored 16-byte pieces of 256 open lines per wave ran at 0.7 TB/s: with
// thousands of waves the open lines overflow the L2 and reach HBM as masked partial writes):
//
//   * features are channels-last [HW][128] f16, so both MFMA operands (8 consecutive k of one
//     row per lane) are plain 16-byte loads of L2-resident data;
//   * a wave owns 32 source pixels p (MFMA B operand, all of K resident in 32 VGPRs) and walks the
//     target image row by row in chunks of 64 pixels: two v_mfma_f32_32x32x16_f16 tiles whose
//     A-operand rows are PERMUTED target pixels, chosen so that the 2 x 16 accumulators of a lane
//     are 32 consecutive x of one (p, y) row: a lane stores 64 contiguous bytes, the lane pair
//     (l, l+32) a complete 128-byte line -- every volume line is written once, whole;
//   * the 2x2 / 4x4 / 8x8 average pools run in the lane's own registers across the 8 rows of a
//     band (no cross-lane traffic at all) and are stored as 32 / 16 / 8-byte runs;
//   * rounding follows the reference exactly: level 0 = f16(f32 accumulator), level l+1 =
//     f16((a+b+c+d in f32, row-major order)/4) of the ROUNDED level-l values.
#include <cstdlib>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct VolArgs {
  const _Float16* f1;  // [n1][HW][C] channels-last, already divided by 4
  const _Float16* f2;  // [n2][HW][C]
  const int64_t* ii;   // optional frame indices into f1 / f2 (null: edge e uses row e)
  const int64_t* jj;
  _Float16* pyr[4];    // level l: [E][HW][h>>l][w>>l]
  int E, ht, wd, num_levels;
  int tiled;  // levels 0 and 1 in 8x8-tiled slices (see corr_volume_tiled_kernel)
  const int* slot;  // optional: edge e is written to volume index slot[e] of pyr[*] (slot-addressed pools; null: e)
};

__device__ __forceinline__ _Float16 pool4(_Float16 a, _Float16 b, _Float16 c, _Float16 d) {
  float acc = 0.0f;
  acc += (float)a;
  acc += (float)b;
  acc += (float)c;
  acc += (float)d;
  return (_Float16)(acc / 4.0f);
}

// The same pooling on PACKED pairs, two results at a time, for the tiled kernel (round 6).  The plain form costs 13 vector
// instructions per result (the compiler re-rounds the f32 accumulators to f16, converts back, adds 0, ...), 150 of the ~280 a
// wave issues per tile, and the kernel is bound by instruction issue (rocprofv3 --pmc: each SIMD issues 60 % of the time at two
// waves, LDS 21 % busy, matrix core 14 %).  v_fma_mix_f32 reads an f16 HALF of a packed register as an f32 operand:
//   t = lo(p) * 1 + hi(p);  t = lo(q) * 1 + t;  t = hi(q) * 1 + t      == ((0 + a) + b) + c) + d in f32, the same three roundings
//   v_fma_mixlo/hi_f16: half = f16(t * 0.25 + 0)                        == f16(t / 4): the product is exact, ONE rounding to f16;
//                                                                          the +0 addend turns a -0 sum into the +0 that 0 + a gives
// 4 instructions per result, no unpacking, bit-identical (tests/test_corr_gpu.py compares every level with the plain form's bits).
__device__ __forceinline__ float pool_sum(f16x2 p, f16x2 q) {
  float t;
  asm("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(p));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(q), "v"(t));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(q), "v"(t));
  return t;
}
// {pool(p0.lo, p0.hi, q0.lo, q0.hi), pool(p1.lo, p1.hi, q1.lo, q1.hi)}
__device__ __forceinline__ f16x2 pool4x2(f16x2 p0, f16x2 q0, f16x2 p1, f16x2 q1, float quarter) {
  const float t0 = pool_sum(p0, q0), t1 = pool_sum(p1, q1);
  f16x2 d;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(t0), "v"(quarter));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(t1), "v"(quarter));
  return d;
}
__device__ __forceinline__ _Float16 pool4x1(f16x2 p, f16x2 q, float quarter) {
  const float t = pool_sum(p, q);
  f16x2 d;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(t), "v"(quarter));
  return d[0];
}

// f16 values travel in PACKED pairs (one VGPR per two values): the row, its three pooling carries and the
// pooled rows would otherwise take one VGPR per half and push the kernel to one wave per SIMD.
__device__ __forceinline__ f16x2 mk2(_Float16 a, _Float16 b) {
  const f16x2 v = {a, b};
  return v;
}

// store NP consecutive pairs (2*NP halves) at dst (2-byte aligned): 16 / 8 / 4-byte pieces, element-wise at the
// image border (nvalid = number of valid halves from dst on)
template <int NP>
__device__ __forceinline__ void store_pairs(_Float16* dst, const f16x2* v, int nvalid) {
  struct __attribute__((packed, aligned(2))) U8 { f16x8 v; };
  struct __attribute__((packed, aligned(2))) U4 { f16x4 v; };
  struct __attribute__((packed, aligned(2))) U2 { f16x2 v; };
  int k = 0;
#pragma unroll
  for (; k + 4 <= NP; k += 4) {
    if (nvalid - 2 * k >= 8) {
      const f16x8 o = {v[k][0], v[k][1], v[k + 1][0], v[k + 1][1], v[k + 2][0], v[k + 2][1], v[k + 3][0], v[k + 3][1]};
      reinterpret_cast<U8*>(dst + 2 * k)->v = o;
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (2 * k + q < nvalid) dst[2 * k + q] = v[k + (q >> 1)][q & 1];
    }
  }
  if (NP - k >= 2) {
    if (nvalid - 2 * k >= 4) {
      const f16x4 o = {v[k][0], v[k][1], v[k + 1][0], v[k + 1][1]};
      reinterpret_cast<U4*>(dst + 2 * k)->v = o;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (2 * k + q < nvalid) dst[2 * k + q] = v[k + (q >> 1)][q & 1];
    }
    k += 2;
  }
  if (NP - k >= 1) {
    if (nvalid - 2 * k >= 2) {
      reinterpret_cast<U2*>(dst + 2 * k)->v = v[k];
    } else if (2 * k < nvalid) {
      dst[2 * k] = v[k][0];
    }
  }
}

// ---- LDS staging of the target-image rows --------------------------------------------------------
// One row chunk = up to 64 consecutive target pixels x 128 channels = 16 KiB.  The four waves of a
// workgroup (4 x 32 source pixels) all multiply against the same chunk, so it is fetched ONCE per
// workgroup with fully coalesced 1 KiB wave loads (lane = 16 bytes, 16 lanes per pixel row) instead
// of 16 scattered 16-byte loads per lane and wave (which made the first version address-coalescer
// bound).  Row stride 272 B (256 + 16) puts the 16-byte slot of (row r, slot s) at (17 r + s) mod 16,
// which makes both the staging writes and the fragment reads (row = permuted x, see below) conflict
// free.  Two buffers: the loads of the next row are in flight while the current one is multiplied.
#define ROWB 272

// One staged row chunk = 32*NT consecutive target pixels x 128 channels (NT <= 3: 26 KiB with the padding).
template <int C, int NT>
__device__ __forceinline__ void stage_load(const _Float16* __restrict__ F2row, int x0, int wd, int tid, uint4* regs) {
  // 32*NT rows x 256 B = 512*NT pieces of 16 B over 256 lanes: 2*NT per lane; piece id = tid + 256*k
#pragma unroll
  for (int k = 0; k < 2 * NT; k++) {
    const int piece = tid + 256 * k;
    const int r = piece >> 4, sl = piece & 15;
    const int x = x0 + r;
    regs[k] = (x < wd) ? *reinterpret_cast<const uint4*>(F2row + (long)x * C + sl * 8) : make_uint4(0, 0, 0, 0);
  }
}

template <int NT>
__device__ __forceinline__ void stage_store(char* tile, int tid, const uint4* regs) {
#pragma unroll
  for (int k = 0; k < 2 * NT; k++) {
    const int piece = tid + 256 * k;
    const int r = piece >> 4, sl = piece & 15;
    *reinterpret_cast<uint4*>(tile + r * ROWB + sl * 16) = regs[k];
  }
}

// NT tiles (NT*32 target pixels of the staged row) against the wave's 32 source pixels.
// Lane (col, half) ends up with v[0 .. 16*NT) = corr(p, y, x0 + 16*NT*half + k).
template <int C, int NT>
__device__ __forceinline__ void row_chunk(const char* tile, const f16x8* src, int col, int half, f16x2* v) {
  constexpr int KS = C / 16;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = (f32x16)0.0f;
  // A-operand row i of tile t  <->  staged row  xr = (i&3) + 4*(i>>3) + 16*t + 16*NT*((i>>2)&1)
  const char* rows[NT];
#pragma unroll
  for (int t = 0; t < NT; t++)
    rows[t] = tile + ((col & 3) + 4 * (col >> 3) + 16 * t + 16 * NT * ((col >> 2) & 1)) * ROWB + half * 16;
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
#pragma unroll
    for (int t = 0; t < NT; t++) {  // interleave the independent accumulator chains
      const f16x8 tgt = *reinterpret_cast<const f16x8*>(rows[t] + kk * 32);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tgt, src[kk], acc[t], 0, 0, 0);
    }
    // at most two k-steps of LDS fragments in flight (register budget).  (round 5, measured: all 16 fragment reads of a tile
    // issued before its first MFMA, 196 registers instead of 164: 228-230 us per 10-edge launch either way)
    if (kk & 1) __builtin_amdgcn_sched_barrier(0);
  }
  // accumulator r of tile t (this lane's half h): i = (r&3) + 8*(r>>2) + 4*h  ->  x - x0 - 16*NT*h = r + 16*t
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 16; r += 2) v[8 * t + (r >> 1)] = mk2((_Float16)acc[t][r], (_Float16)acc[t][r + 1]);
}

// Rows [r0, r1) (r0 a multiple of 8) x one chunk of 32*NT target columns, for the whole workgroup.
// Software pipeline over the rows with THREE LDS buffers: while row y is multiplied, row y+1 is already in
// LDS and the global loads of row y+2 are in flight (an L2 round trip is about as long as one row of work;
// with two buffers every row waited for it).  One barrier per row.
template <int C, int NT>
__device__ __forceinline__ void rows_chunk(const VolArgs& a, const _Float16* __restrict__ F2, const f16x8* src,
                                           char* lds, int r0, int r1, int x0, int col, int half, bool pok,
                                           _Float16* o0, _Float16* o1, _Float16* o2, _Float16* o3) {
  constexpr int W = 16 * NT;  // consecutive x held by a lane
  constexpr int TILEB = 32 * NT * ROWB;
  const int wd = a.wd, ht = a.ht, tid = threadIdx.x;
  const int w1 = wd >> 1, w2 = wd >> 2, w3 = wd >> 3, h1 = ht >> 1, h2 = ht >> 2, h3 = ht >> 3;
  const int xl = x0 + W * half;  // first x of this lane
  uint4 regs[2 * NT];
  __syncthreads();  // every wave is done reading the buffers (previous chunk)
  stage_load<C, NT>(F2 + (long)r0 * wd * C, x0, wd, tid, regs);
  stage_store<NT>(lds, tid, regs);
  if (r0 + 1 < r1) stage_load<C, NT>(F2 + (long)(r0 + 1) * wd * C, x0, wd, tid, regs);
  __syncthreads();
  f16x2 prev0[W / 2], prev1[W / 4], prev2[W / 8];  // pooling carries: last odd..even row of levels 0, 1, 2
  int buf = 0;
#pragma unroll 1
  for (int y = r0; y < r1; y++) {
    const int yy = y & 7;
    const char* cur = lds + buf * TILEB;
    const int nb = buf == 2 ? 0 : buf + 1;
    if (y + 1 < r1) stage_store<NT>(lds + nb * TILEB, tid, regs);                                // row y+1 -> LDS
    if (y + 2 < r1) stage_load<C, NT>(F2 + (long)(y + 2) * wd * C, x0, wd, tid, regs);           // row y+2 in flight
    buf = nb;
    f16x2 v[W / 2];
    row_chunk<C, NT>(cur, src, col, half, v);
    if (pok) store_pairs<W / 2>(o0 + (long)y * wd + xl, v, wd - xl);
    if (a.num_levels > 1) {
      if (yy & 1) {
        f16x2 l1[W / 4];  // element c of level 1 = pool of pair c of the two level-0 rows
#pragma unroll
        for (int c = 0; c < W / 4; c++)
          l1[c] = mk2(pool4(prev0[2 * c][0], prev0[2 * c][1], v[2 * c][0], v[2 * c][1]),
                      pool4(prev0[2 * c + 1][0], prev0[2 * c + 1][1], v[2 * c + 1][0], v[2 * c + 1][1]));
        if (pok && (y >> 1) < h1) store_pairs<W / 4>(o1 + (long)(y >> 1) * w1 + (xl >> 1), l1, w1 - (xl >> 1));
        if (a.num_levels > 2) {
          if ((yy & 3) == 3) {
            f16x2 l2[W / 8];
#pragma unroll
            for (int c = 0; c < W / 8; c++)
              l2[c] = mk2(pool4(prev1[2 * c][0], prev1[2 * c][1], l1[2 * c][0], l1[2 * c][1]),
                          pool4(prev1[2 * c + 1][0], prev1[2 * c + 1][1], l1[2 * c + 1][0], l1[2 * c + 1][1]));
            if (pok && (y >> 2) < h2) store_pairs<W / 8>(o2 + (long)(y >> 2) * w2 + (xl >> 2), l2, w2 - (xl >> 2));
            if (a.num_levels > 3) {
              if (yy == 7) {
                f16x2 l3[W / 16];
#pragma unroll
                for (int c = 0; c < W / 16; c++)
                  l3[c] = mk2(pool4(prev2[2 * c][0], prev2[2 * c][1], l2[2 * c][0], l2[2 * c][1]),
                              pool4(prev2[2 * c + 1][0], prev2[2 * c + 1][1], l2[2 * c + 1][0], l2[2 * c + 1][1]));
                if (pok && (y >> 3) < h3) store_pairs<W / 16>(o3 + (long)(y >> 3) * w3 + (xl >> 3), l3, w3 - (xl >> 3));
              } else {
#pragma unroll
                for (int c = 0; c < W / 8; c++) prev2[c] = l2[c];
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < W / 4; c++) prev1[c] = l1[c];
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < W / 2; c++) prev0[c] = v[c];
      }
    }
    __syncthreads();  // row y+1 is complete in LDS; the buffer of row y is free for row y+3
  }
}

template <int C, int MAXNT>
__global__ __launch_bounds__(256) void corr_volume_pyramid_kernel(VolArgs a) {
  constexpr int KS = C / 16;  // k-steps of the 32x32x16 MFMA
  __shared__ __attribute__((aligned(16))) char lds[3 * 32 * MAXNT * ROWB];  // three row buffers of 32*MAXNT pixels
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int HW = a.ht * a.wd;
  const int e = blockIdx.y;
  const int p0 = (blockIdx.x * 4 + wave) * 32;
  const long fi = a.ii ? a.ii[e] : e, fj = a.jj ? a.jj[e] : e;
  const int eo = a.slot ? a.slot[e] : e;  // output volume index
  const _Float16* __restrict__ F1 = a.f1 + fi * (long)HW * C;
  const _Float16* __restrict__ F2 = a.f2 + fj * (long)HW * C;

  // B operand of the MFMA = this wave's 32 source pixels, all K, resident in registers.
  // Waves past the end of the image stay alive (they take part in the staging and the barriers).
  const int p = p0 + col;
  const bool pok = p < HW;
  f16x8 src[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    src[kk] = *reinterpret_cast<const f16x8*>(F1 + (long)(pok ? p : 0) * C + kk * 16 + half * 8);
    if (!pok) src[kk] = (f16x8)(_Float16)0;
  }
  const long pp = (long)eo * HW + (pok ? p : 0);
  _Float16* o0 = a.pyr[0] + pp * (long)HW;
  _Float16* o1 = a.num_levels > 1 ? a.pyr[1] + pp * (long)(a.ht >> 1) * (a.wd >> 1) : nullptr;
  _Float16* o2 = a.num_levels > 2 ? a.pyr[2] + pp * (long)(a.ht >> 2) * (a.wd >> 2) : nullptr;
  _Float16* o3 = a.num_levels > 3 ? a.pyr[3] + pp * (long)(a.ht >> 3) * (a.wd >> 3) : nullptr;

  // contiguous range of 8-row bands for this z slice (the row pipeline runs across band boundaries)
  const int nby = (a.ht + 7) >> 3;
  const int b0 = (int)((long)blockIdx.z * nby / gridDim.z), b1 = (int)((long)(blockIdx.z + 1) * nby / gridDim.z);
  const int r0 = b0 * 8, r1 = min(b1 * 8, a.ht);
  if (r0 >= r1) return;  // workgroup-uniform
  int x0 = 0;
  if (MAXNT >= 3)
    for (; x0 + 64 < a.wd; x0 += 96) rows_chunk<C, 3>(a, F2, src, lds, r0, r1, x0, col, half, pok, o0, o1, o2, o3);
  if (MAXNT >= 2)
    for (; x0 + 32 < a.wd; x0 += 64) rows_chunk<C, 2>(a, F2, src, lds, r0, r1, x0, col, half, pok, o0, o1, o2, o3);
  for (; x0 < a.wd; x0 += 32) rows_chunk<C, 1>(a, F2, src, lds, r0, r1, x0, col, half, pok, o0, o1, o2, o3);
}

// ---------------------------------------------------------------------------------------------
// TILED variant.  Levels 0 and 1 are stored as 8x8 tiles of 128 bytes (one cache line):
//     slice(e, p, l) = [ceil(h_l/8)][ceil(w_l/8)][8][8] f16         (levels 2, 3 stay row-major: they are 1-5 lines)
// Why: (1) the lookup's 8x8 tap window then touches <= 2x2 lines instead of 8-9 (row pitch 160 B), which is what
// its HBM traffic is made of; (2) this kernel can walk the target image TILE BY TILE: the 64 target pixels of a
// tile are the 64 A-operand rows of two MFMAs, lane (p, half) ends up with rows 4*half..4*half+3 of the tile =
// 64 contiguous bytes of the output line (the lane pair writes the whole line), and ALL pooling for the tile
// (4x4 level-1, 2x2 level-2, 1 level-3 value) happens in the lane's registers at once -- no carries across rows,
// and no row-parity control flow.  Tiles are visited in 2x2 groups = one level-1 tile, one 4x4 level-2 block and one
// 2x2 level-3 block per source pixel, and the OUTPUT is batched per tile / per group into whole 128-byte lines
// (through a wave-private LDS transposition) because the kernel is bound by the number of L2 write requests, not
// by bytes: 335 -> 230 us for 10 edges when the 16-byte-per-lane stores became 8-lanes-per-line stores.
// ---------------------------------------------------------------------------------------------
typedef uint32_t vol_u32x4 __attribute__((ext_vector_type(4)));

// Raw-buffer addressing for the tile loop (round 5).  The loop prefetches the target tile two iterations ahead and ends every
// iteration with stores nobody in the kernel reads.  gfx950 has ONE counter for vector loads and stores (vmcnt, in issue order),
// so "the prefetched tile has arrived" is `s_waitcnt vmcnt(n)` with n = the memory instructions issued after those loads.  With
// flat loads / stores under per-lane `if`s the compiler branches around them (s_cbranch_execz), can no longer count, and emits
// vmcnt(0): every iteration then also waited for the PREVIOUS tile's stores to be acknowledged by the L2.  Buffer instructions
// need no branch: a lane outside the image gets an offset past num_records (loads return 0, stores are dropped), so the level-0
// stores are exactly four instructions on every path and the compiler's own count comes out as vmcnt(7..4): the tile's stores
// stay in flight behind the next tile's MFMAs.  Measured on one box, same bits: 235-237 -> 228-230 us per 10-edge launch --
// 3 %, so the store acknowledgement was a small part of the tile iteration, not the chain that bounds it (two workgroups
// of 72 KB of LDS per CU already overlap one another's waits).
#define VOL_OOB 0x80000000u

__device__ __forceinline__ void stage_tile_load(__amdgpu_buffer_rsrc_t F2, int ty, int tx, int ht, int wd, int tid,
                                                uint4* regs) {
  // 64 pixels x 16 pieces of 16 B; LDS row o = 8 * (row in tile) + (column in tile)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int piece = tid + 256 * k;
    const int o = piece >> 4, sl = piece & 15;
    const int y = 8 * ty + (o >> 3), x = 8 * tx + (o & 7);
    const uint32_t off = (y < ht && x < wd) ? (uint32_t)(y * wd + x) * 256u + sl * 16u : VOL_OOB;
    const vol_u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(F2, off, 0, 0);
    regs[k] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

struct TileWalk {  // tiles of a z-slice in 2x2-group order, skipping the ones outside the tile grid
  int t, t_end, ngx, nty, ntx;
  int gy, gx, sub;  // the group and the tile inside it that t names (kept incrementally: a division per tile and walker was
                    // ~60 of the ~160 scalar instructions of a tile iteration)
  __device__ __forceinline__ void start(int t0) {  // t0: a multiple of 4
    t = t0;
    gy = (t0 >> 2) / ngx;
    gx = (t0 >> 2) % ngx;
    sub = 0;
  }
  __device__ __forceinline__ bool next(int& ty, int& tx) {  // advances to the next valid tile; false at the end
    while (t < t_end) {
      ty = 2 * gy + (sub >> 1);
      tx = 2 * gx + (sub & 1);
      const bool ok = ty < nty && tx < ntx;
      t++;
      if (++sub == 4) {
        sub = 0;
        if (++gx == ngx) {
          gx = 0;
          gy++;
        }
      }
      if (ok) return true;
    }
    return false;
  }
};

// BAND (round 6): levels 2 and 3 are row-major and small (a 4x4 / 2x2 block per source pixel and tile group), and rounds 3-5 stored
// each block from registers as it completed: 8- and 4-byte pieces, every one its own partial-line write -- 48 of the 88 write
// requests of a tile, 757 MB at the memory for 611 MB of pyramid, 45 of the 205 us of a 10-edge launch
// (tools/vol_levels_bench.py: 4 / 3 / 2 / 1 levels = 205 / 211 / 178 / 155 us).  BAND = 1 collects the blocks of up to VOL_RUN
// consecutive groups of one band (two tile rows) per wave in LDS and writes them when the band -- or the workgroup's share of it --
// ends: at 60x80 a band is the whole width, i.e. 4 level-2 rows = 160 contiguous bytes and 2 level-3 rows = 40 bytes per source
// pixel, stored as runs of consecutive lanes.  The LDS for it comes from the staging: ONE tile buffer and a second barrier per
// tile (measured alone, 3 workgroups per CU: 202 vs 203 us -- neither a gain nor a loss).
#define VOL_RUN 5        // groups per run: 4 x 20 level-2 values, 2 x 10 level-3 values per source pixel
#define VOL_P2 168       // bytes per source pixel in the level-2 run buffer (160 + 8)
#define VOL_P3 44        // level 3 (40 + 4)

template <int C, int BAND>
__global__ __launch_bounds__(256) void corr_volume_tiled_kernel(VolArgs a) {
  static_assert(C == 128, "staging assumes 128 channels");
  constexpr int KS = C / 16;
  constexpr int TILEB = 64 * ROWB;
  constexpr int XPITCH = 144;  // 128-byte output line + 16: conflict-free 16-byte slots for both access patterns
  // BAND = 0: two staging buffers and one barrier per tile: iteration t fills the buffer that was read in iteration t-1 (all
  // waves are past that barrier) and reads the one filled during t-1.  BAND = 1: one buffer, filled between two barriers.
  __shared__ __attribute__((aligned(16))) char lds[(BAND ? 1 : 2) * TILEB];
  // per wave: 32 level-0 lines of the current tile + 32 level-1 lines of the current 2x2 tile group
  __shared__ __attribute__((aligned(16))) char xpose[4][2][32 * XPITCH];
  __shared__ __attribute__((aligned(16))) char runbuf[BAND ? 4 : 1][BAND ? 32 * (VOL_P2 + VOL_P3) : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int ht = a.ht, wd = a.wd, HW = ht * wd;
  const int e = blockIdx.y;
  const int p0 = (blockIdx.x * 4 + wave) * 32;
  const long fi = a.ii ? a.ii[e] : e, fj = a.jj ? a.jj[e] : e;
  const int eo = a.slot ? a.slot[e] : e;  // output volume index
  const _Float16* __restrict__ F1 = a.f1 + fi * (long)HW * C;
  const __amdgpu_buffer_rsrc_t F2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.f2 + fj * (long)HW * C), 0, HW * C * 2, 0x00027000);
  const int p = p0 + col;
  const bool pok = p < HW;
  f16x8 src[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    src[kk] = *reinterpret_cast<const f16x8*>(F1 + (long)(pok ? p : 0) * C + kk * 16 + half * 8);
    if (!pok) src[kk] = (f16x8)(_Float16)0;
  }
  const int h1 = ht >> 1, w1 = wd >> 1, h2 = ht >> 2, w2 = wd >> 2, h3 = ht >> 3, w3 = wd >> 3;
  const int nty0 = (ht + 7) >> 3, ntx0 = (wd + 7) >> 3, nty1 = (h1 + 7) >> 3, ntx1 = (w1 + 7) >> 3;
  const long slice0 = (long)nty0 * ntx0 * 64, slice1 = (long)nty1 * ntx1 * 64;
  const long pp = (long)eo * HW + (pok ? p : 0);
  _Float16* o2 = a.num_levels > 2 ? a.pyr[2] + pp * (long)h2 * w2 : nullptr;
  _Float16* o3 = a.num_levels > 3 ? a.pyr[3] + pp * (long)h3 * w3 : nullptr;
  // this edge's level-0 / level-1 slices as buffers of exactly HW lines-of-tiles: source pixels >= HW fall off the end
  // (the host checks that a slice set stays under 2 GiB)
  const __amdgpu_buffer_rsrc_t O0 =
      __builtin_amdgcn_make_buffer_rsrc(a.pyr[0] + (long)eo * HW * slice0, 0, (int)(HW * slice0 * 2), 0x00027000);
  const __amdgpu_buffer_rsrc_t O1 = __builtin_amdgcn_make_buffer_rsrc(
      a.num_levels > 1 ? a.pyr[1] + (long)eo * HW * slice1 : a.pyr[0], 0, a.num_levels > 1 ? (int)(HW * slice1 * 2) : 0, 0x00027000);
  char* xp0 = xpose[wave][0];
  char* xp1 = xpose[wave][1];

  TileWalk ld;   // walks ahead of the compute: issues the global loads
  ld.ngx = (ntx0 + 1) >> 1; ld.nty = nty0; ld.ntx = ntx0;
  const int ngroups = ((nty0 + 1) >> 1) * ld.ngx;
  ld.start(4 * (int)((long)blockIdx.z * ngroups / gridDim.z));
  ld.t_end = 4 * (int)((long)(blockIdx.z + 1) * ngroups / gridDim.z);
  TileWalk cp = ld;  // compute cursor
  uint4 regs[4];
  int lty, ltx, ty, tx;
  if (!ld.next(lty, ltx)) return;  // workgroup-uniform: empty slice
  stage_tile_load(F2, lty, ltx, ht, wd, tid, regs);
  stage_store<2>(lds, tid, regs);
  bool have_next = ld.next(lty, ltx);
  if (have_next) stage_tile_load(F2, lty, ltx, ht, wd, tid, regs);
  lds_barrier();  // (not __syncthreads(): the second tile's loads stay in flight)
  int buf = 0;
  f16x4 l2g[2] = {(f16x4)(_Float16)0, (f16x4)(_Float16)0};  // level-2 rows (half, 2 + half) of the group, 4 columns
  f16x2 l3g[2] = {(f16x2)(_Float16)0, (f16x2)(_Float16)0};  // level-3 rows 0, 1 of the group (held by half == 0)
  const float quarter = 0.25f;
  int run_n = 0, run_gx0 = 0, run_gy = 0;  // BAND: groups in the run buffers, the run's first group column, its band
  while (cp.next(ty, tx)) {
    const char* cur = lds + (BAND ? 0 : buf) * TILEB;
    f16x2 v[16];  // v[k] = tile-local offsets 32*half + 2k, +1  (rows 4*half .. 4*half+3, 8 columns each)
    row_chunk<C, 2>(cur, src, col, half, v);
    if (BAND) lds_barrier();  // every wave has its fragments: the buffer takes the next tile below
#pragma unroll
    for (int k = 0; k < 16; k++) asm("" : "+v"(v[k]));  // (the packed pairs are values, not recipes: without this the compiler
                                                        //  converts half of the accumulators a second time for the pooling)
    const int dy = ty & 1, dx = tx & 1, gy = ty >> 1, gx = tx >> 1;
    // last tile of its 2x2 group in walk order (the others are outside the tile grid)?
    const bool last_in_group = !((dx == 0 && tx + 1 < ntx0) || (dy == 0 && ty + 1 < nty0));
    {
      // level 0.  The L2 takes ~190 G write requests/s and that, not bytes, bounds this kernel: a lane's 16-byte store
      // is its own request (the lines of different source pixels are 10 KB apart).  So the wave's 32 lines (4 KB) go
      // through a wave-private LDS area and leave as 4 stores in which 8 consecutive lanes cover one whole 128-byte
      // line: 32 requests per tile instead of 256.
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f16x8 o = {v[4 * q][0], v[4 * q][1], v[4 * q + 1][0], v[4 * q + 1][1],
                         v[4 * q + 2][0], v[4 * q + 2][1], v[4 * q + 3][0], v[4 * q + 3][1]};
        *reinterpret_cast<f16x8*>(xp0 + col * XPITCH + 64 * half + 16 * q) = o;
      }
    }
    f16x2 l1[2][2];
    if (a.num_levels > 1) {
      // level 1: rows rr = 0,1 <- tile rows (4h + 2rr, 4h + 2rr + 1); v index of (row R, col c) = 4 (R - 4h) + c / 2
#pragma unroll
      for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
          const f16x2 u0 = v[8 * rr + 2 * cc], u1 = v[8 * rr + 2 * cc + 1];          // upper row, columns 4cc..4cc+3
          const f16x2 d0_ = v[8 * rr + 4 + 2 * cc], d1_ = v[8 * rr + 4 + 2 * cc + 1];  // lower row
          l1[rr][cc] = pool4x2(u0, d0_, u1, d1_, quarter);
        }
      // into the group's level-1 line of this source pixel: row 4 dy + 2 half + rr, columns 4 dx .. 4 dx + 3
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
        const f16x4 o = {l1[rr][0][0], l1[rr][0][1], l1[rr][1][0], l1[rr][1][1]};
        *reinterpret_cast<f16x4*>(xp1 + col * XPITCH + (4 * dy + 2 * half + rr) * 16 + 8 * dx) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private areas: LDS ops of one wave complete in order
    {
      const uint32_t tile_off = (uint32_t)(ty * ntx0 + tx) * 128u;  // bytes
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int line = 8 * k + (lane >> 3), piece = lane & 7;
        const vol_u32x4 d = *reinterpret_cast<const vol_u32x4*>(xp0 + line * XPITCH + 16 * piece);
        const uint32_t pl = p0 + line;  // pl >= HW: past num_records, dropped
        __builtin_amdgcn_raw_buffer_store_b128(d, O0, pl * (uint32_t)(slice0 * 2) + tile_off + 16u * piece, 0, 0);
      }
    }
    // Staging for the next two tiles sits HERE, right behind the four level-0 stores that every path issues: "tile t+1 has
    // arrived" is then vmcnt(4) -- everything older than this tile's own level-0 stores, which stay in flight.  The
    // group's level-1 / 2 / 3 stores (every fourth tile) follow the prefetch, so the wait never covers stores issued less than
    // a whole iteration ago.  buf ^ 1 was read during tile t-1; every wave is past that iteration's barrier.
    if (have_next) {
      stage_store<2>(lds + (BAND ? 0 : (buf ^ 1)) * TILEB, tid, regs);    // tile t+1 -> LDS
      have_next = ld.next(lty, ltx);
      if (have_next) stage_tile_load(F2, lty, ltx, ht, wd, tid, regs);    // tile t+2 in flight
    }
    buf ^= 1;
    if (a.num_levels > 1 && last_in_group && gy < nty1 && gx < ntx1) {
      // the group's level-1 tile of every source pixel is complete (parts from tiles outside the grid are padding)
      const uint32_t tile_off = (uint32_t)(gy * ntx1 + gx) * 128u;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int line = 8 * k + (lane >> 3), piece = lane & 7;
        const vol_u32x4 d = *reinterpret_cast<const vol_u32x4*>(xp1 + line * XPITCH + 16 * piece);
        const uint32_t pl = p0 + line;
        __builtin_amdgcn_raw_buffer_store_b128(d, O1, pl * (uint32_t)(slice1 * 2) + tile_off + 16u * piece, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next tile overwrites the areas
    if (a.num_levels > 2) {
      // level 2: this tile gives one row (2 dy + half of the group's 4) and two columns (2 dx, 2 dx + 1); kept in
      // registers until the group is complete, then two 8-byte stores per lane instead of four 4-byte ones
      const f16x2 l2 = pool4x2(l1[0][0], l1[1][0], l1[0][1], l1[1][1], quarter);
#pragma unroll
      for (int ry = 0; ry < 2; ry++)
        if (dy == ry) {
          if (dx == 0) { l2g[ry][0] = l2[0]; l2g[ry][1] = l2[1]; }
          else         { l2g[ry][2] = l2[0]; l2g[ry][3] = l2[1]; }
        }
      if (a.num_levels > 3) {
        // level 3: the tile's single value needs both halves' level-2 rows
        const uint32_t mine = __builtin_bit_cast(uint32_t, l2);
        const f16x2 other = __builtin_bit_cast(f16x2, (uint32_t)__shfl_xor((int)mine, 32));
        const _Float16 l3 = pool4x1(l2, other, quarter);  // meaningful on half == 0
#pragma unroll
        for (int ry = 0; ry < 2; ry++)
          if (dy == ry) l3g[ry][dx] = l3;
      }
      struct __attribute__((packed, aligned(2))) U4 { f16x4 v; };
      struct __attribute__((packed, aligned(2))) U2 { f16x2 v; };
      if (BAND && last_in_group) {
        // the group's blocks into the run buffers (rows / columns outside the image are dropped when the run is written)
        char* b2 = runbuf[wave];
        char* b3 = b2 + 32 * VOL_P2;
        if (run_n == 0) { run_gx0 = gx; run_gy = gy; }
#pragma unroll
        for (int ry = 0; ry < 2; ry++) {
          *reinterpret_cast<f16x4*>(b2 + col * VOL_P2 + (2 * ry + half) * (VOL_RUN * 8) + run_n * 8) = l2g[ry];
          if (a.num_levels > 3 && half == 0) *reinterpret_cast<f16x2*>(b3 + col * VOL_P3 + ry * (VOL_RUN * 4) + run_n * 4) = l3g[ry];
        }
        run_n++;
        // the run ends with the band, with this workgroup's share of the slice, or when the buffer is full
        if (run_n == VOL_RUN || gx + 1 == cp.ngx || 4 * (((cp.t - 1) >> 2) + 1) >= cp.t_end) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const long h2w2 = (long)h2 * w2, h3w3 = (long)h3 * w3;
#pragma unroll 2
          for (int it = 0; it < (32 * 4 * VOL_RUN) / 64; it++) {
            const int q = it * 64 + lane, px = q / (4 * VOL_RUN), rem = q % (4 * VOL_RUN), r = rem / VOL_RUN, g = rem % VOL_RUN;
            const int Y2 = 4 * run_gy + r, X2 = 4 * (run_gx0 + g);
            if (g < run_n && p0 + px < HW && Y2 < h2 && X2 < w2) {
              const f16x4 d = *reinterpret_cast<const f16x4*>(b2 + px * VOL_P2 + r * (VOL_RUN * 8) + g * 8);
              _Float16* d2p = a.pyr[2] + ((long)eo * HW + p0 + px) * h2w2 + (long)Y2 * w2 + X2;
              if (X2 + 4 <= w2) {
                reinterpret_cast<U4*>(d2p)->v = d;
              } else {
#pragma unroll
                for (int c = 0; c < 4; c++)
                  if (X2 + c < w2) d2p[c] = d[c];
              }
            }
          }
          if (a.num_levels > 3) {
#pragma unroll 1
            for (int it = 0; it < (32 * 2 * VOL_RUN) / 64; it++) {
              const int q = it * 64 + lane, px = q / (2 * VOL_RUN), rem = q % (2 * VOL_RUN), ry = rem / VOL_RUN, g = rem % VOL_RUN;
              const int Y3 = 2 * run_gy + ry, X3 = 2 * (run_gx0 + g);
              if (g < run_n && p0 + px < HW && Y3 < h3 && X3 < w3) {
                const f16x2 d = *reinterpret_cast<const f16x2*>(b3 + px * VOL_P3 + ry * (VOL_RUN * 4) + g * 4);
                _Float16* d3p = a.pyr[3] + ((long)eo * HW + p0 + px) * h3w3 + (long)Y3 * w3 + X3;
                if (X3 + 2 <= w3) {
                  reinterpret_cast<U2*>(d3p)->v = d;
                } else {
                  d3p[0] = d[0];
                }
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next run overwrites the buffers)
          run_n = 0;
        }
      }
      if (!BAND && last_in_group && pok) {
#pragma unroll
        for (int ry = 0; ry < 2; ry++) {
          const int Y2 = 4 * gy + 2 * ry + half, X2 = 4 * gx;
          if (Y2 < h2 && (2 * gy + ry) < nty0) {
            _Float16* d2p = o2 + (long)Y2 * w2 + X2;
            if (X2 + 4 <= w2) {
              reinterpret_cast<U4*>(d2p)->v = l2g[ry];
            } else {
#pragma unroll
              for (int c = 0; c < 4; c++)
                if (X2 + c < w2) d2p[c] = l2g[ry][c];
            }
          }
          if (a.num_levels > 3 && half == 0) {
            const int Y3 = 2 * gy + ry, X3 = 2 * gx;
            if (Y3 < h3) {
              _Float16* d3p = o3 + (long)Y3 * w3 + X3;
              if (X3 + 2 <= w3) {
                reinterpret_cast<U2*>(d3p)->v = l3g[ry];
              } else if (X3 < w3) {
                d3p[0] = l3g[ry][0];
              }
            }
          }
        }
      }
    }
    // LDS-only barrier: the two sides communicate through the staging buffers alone, and __syncthreads() would drain vmcnt
    // (round 4 tried only this and measured nothing, 243.5 vs 242.1 us: the vmcnt(0) in front of the staging store at the top
    // of the loop, see VOL_OOB above, did the same draining one instruction later)
    lds_barrier();
  }
}

extern "C" int ns_corr_volume_pyramid(const void* fmap1, const void* fmap2, const int64_t* ii, const int64_t* jj,
                                      void* const* pyr_host, int num_levels, int E, int C, int ht, int wd,
                                      int tiled, void* stream) {
  return ns_corr_volume_pyramid_slots(fmap1, fmap2, ii, jj, pyr_host, num_levels, E, C, ht, wd, tiled, nullptr, stream);
}

extern "C" int ns_corr_volume_pyramid_slots(const void* fmap1, const void* fmap2, const int64_t* ii, const int64_t* jj,
                                            void* const* pyr_host, int num_levels, int E, int C, int ht, int wd,
                                            int tiled, const int* slot, void* stream) {
  if (E == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(fmap1 && fmap2 && pyr_host, "ns_corr_volume_pyramid: null pointer");
  NS_REQUIRE(num_levels >= 1 && num_levels <= 4, "ns_corr_volume_pyramid: num_levels=%d not in 1..4", num_levels);
  NS_REQUIRE(E >= 0 && ht > 0 && wd > 0, "ns_corr_volume_pyramid: bad shape");
  NS_REQUIRE((ii == nullptr) == (jj == nullptr), "ns_corr_volume_pyramid: pass both index arrays or neither");
  if (C != 128) {
    ns_set_error("ns_corr_volume_pyramid: built for C=128 feature channels (the reference's setting, "
                 "visual_frontend.py:134), got %d", C);
    return NS_ENOSUP;
  }
  if (E == 0) return NS_OK;
  VolArgs a;
  a.f1 = (const _Float16*)fmap1;
  a.f2 = (const _Float16*)fmap2;
  a.ii = ii;
  a.jj = jj;
  for (int l = 0; l < 4; l++) {
    a.pyr[l] = (_Float16*)(l < num_levels ? pyr_host[l] : nullptr);
    NS_REQUIRE(l >= num_levels || a.pyr[l] != nullptr, "ns_corr_volume_pyramid: pyr[%d] is null", l);
  }
  a.E = E;
  a.ht = ht;
  a.wd = wd;
  a.num_levels = num_levels;
  a.tiled = tiled ? 1 : 0;
  a.slot = slot;
  const int HW = ht * wd;
  // The sweep over the target image is latency bound per wave (16 dependent 16-byte loads per 8x8
  // block); split it over 8-row bands until ~4 workgroups per CU are in flight.
  const int nby = (ht + 7) / 8;
  int zs = ns_cdiv(1024, (long)ns_cdiv(HW, 128) * E);
  if (zs > nby) zs = nby;
  if (zs < 1) zs = 1;
  if (tiled) {
    // the tiled kernel addresses one edge's level-0 slices with 32-bit buffer offsets
    const long slice_bytes = (long)HW * nby * ((wd + 7) / 8) * 128;
    if (slice_bytes >= (1L << 31)) {
      ns_set_error("ns_corr_volume_pyramid: a %dx%d grid makes %ld-byte level-0 volumes per edge; the tiled layout is built "
                   "for < 2 GiB per edge", ht, wd, slice_bytes);
      return NS_ENOSUP;
    }
    const int ngroups = ((nby + 1) / 2) * ((((wd + 7) / 8) + 1) / 2);
    // z slices: every workgroup walks ceil(ngroups / zt) groups of 2x2 tiles, and the device runs them in ROUNDS of `slots`
    // resident workgroups (two per CU: 72 KB of LDS, 4 waves of ~230 registers).  Round 4 aimed at ~1024 workgroups whatever
    // that meant in rounds -- E = 10 at 60x80: zt = 3, 1140 workgroups = 2.2 rounds, the last one on a fifth of the chip for
    // as long as the first two.  Pick the zt that minimises rounds x (tiles per workgroup + prologue): same work, same bits.
    static const int slots = [] {
      int dev = 0, cus = 256;
      hipDeviceProp_t pr;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
        cus = pr.multiProcessorCount;
      return 2 * cus;
    }();
    const long per_z = (long)ns_cdiv(HW, 128) * E;
    int zt = 1;
    long best = -1;
    for (int z = 1; z <= ngroups; z++) {
      const long rounds = (per_z * z + slots - 1) / slots;
      const long cost = rounds * (4L * ((ngroups + z - 1) / z) + 3);      // (+3: source fragments + the first tile's staging)
      if (best < 0 || cost < best) {
        best = cost;
        zt = z;
      }
    }
#ifdef NS_TEST_VARIANTS
    if (ns_variant_env("NS_VOL_Z")) zt = atoi(ns_variant_env("NS_VOL_Z"));
    if (ns_variant_env("NS_VOL_NO_BAND"))
      hipLaunchKernelGGL((corr_volume_tiled_kernel<128, 0>), dim3(ns_cdiv(HW, 128), E, zt), dim3(256), 0, (hipStream_t)stream, a);
    else
#endif
    hipLaunchKernelGGL((corr_volume_tiled_kernel<128, 1>), dim3(ns_cdiv(HW, 128), E, zt), dim3(256), 0, (hipStream_t)stream, a);
    NS_CHECK_LAUNCH("corr_volume_tiled_kernel");
    return NS_OK;
  }
  dim3 grid(ns_cdiv(HW, 128), E, zs);
#ifdef NS_TEST_VARIANTS
  static const int mode = ns_variant_env("NS_VOL_NT") ? atoi(ns_variant_env("NS_VOL_NT")) : 2;  // tuning switch: widest column chunk
  if (mode >= 3) {
    hipLaunchKernelGGL((corr_volume_pyramid_kernel<128, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (mode == 2) {
    hipLaunchKernelGGL((corr_volume_pyramid_kernel<128, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((corr_volume_pyramid_kernel<128, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
#else
  hipLaunchKernelGGL((corr_volume_pyramid_kernel<128, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
#endif
  NS_CHECK_LAUNCH("corr_volume_pyramid_kernel");
  return NS_OK;
}
