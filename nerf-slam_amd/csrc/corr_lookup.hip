// corr_lookup.hip -- 4D correlation-volume lookup for gfx950 (MI355X).
//
// Replaces corr_index_forward_kernel / corr_index_backward_kernel of the reference
// (src/correlation_kernels.cu:20-70, 73-124) and fuses the 4-level loop of CorrBlock.__call__
// (networks/modules/corr.py:40-50) into one launch.
//
// Design (HBM-bound gather): one lane per (edge, pixel).  The 8x8 tap window of a pixel is
// eight 16-byte runs of its private [h2,w2] slice, so each lane issues eight unaligned
// global_load_dwordx4 up front (8 x 16 B in flight per lane, 8 KiB per wave) instead of the
// reference's 64 two-byte loads, masks the out-of-image columns with four ANDs per row, and
// interpolates in packed f16.  The 49 outputs are written channel by channel, each wave store
// covering 128 contiguous bytes of one channel plane; the reference's zero-fill pass and its
// 256 global read-modify-writes per pixel do not exist.
//
// Numerics (f16): the reference accumulates in c10::Half -- every product and every running sum
// is rounded to half (float op, then round; double rounding is innocuous for +,* at 24 >= 2*11+2
// bits) in the loop order tap(a,b), tap(a,b+1), tap(a+1,b), tap(a+1,b+1) for output (a,b), and
// out-of-image taps are skipped.  Adding a +0 product is the identity on every value the
// accumulator can hold (it can never be -0), so masking taps to +0 reproduces the skip and the
// result is bit-identical to the reference's arithmetic.  FP contraction is off in this file.
#include <cstdlib>

#include "common.h"
#include <hip/hip_fp16.h>

#pragma clang fp contract(off)

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(2))) Run16 {
  uint32_t d[4];
};

struct LookupLevels {
  const _Float16* vol[4];
  int h2[4];
  int w2[4];
  float scale[4];
  long slice_elems[4];  // h2*w2
  long total_elems[4];  // E*HW1*h2*w2
  int ntx[4];           // > 0: the level is stored as 8x8 tiles, ntx tiles per tile row (slice_elems = nty*ntx*64)
  int num_levels;
};

__device__ __forceinline__ h2_t as_h2(uint32_t u) { return __builtin_bit_cast(h2_t, u); }
__device__ __forceinline__ uint32_t as_u32(h2_t h) { return __builtin_bit_cast(uint32_t, h); }

// One (edge,pixel,level).  out points at channel 0 of this pixel; channel stride = HW1 elements.
__device__ __forceinline__ void lookup_r3_f16(const _Float16* __restrict__ vol, long slice_off, long total,
                                              int h2, int w2, int ntx, float x0, float y0,
                                              _Float16* __restrict__ out, long HW1) {
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  // non-finite or far-away coordinates: every tap is outside -> zeros
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const float dx = sane ? x0 - fx0 : 0.0f, dy = sane ? y0 - fy0 : 0.0f;
  const int xb = sane ? (int)fx0 - 3 : -100000;
  const int yb = sane ? (int)fy0 - 3 : -100000;

  // column masks: half i of the run is a real tap iff 0 <= xb+i < w2
  uint32_t cm[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int xl = xb + 2 * k, xh = xl + 1;
    cm[k] = ((xl >= 0 && xl < w2) ? 0x0000ffffu : 0u) | ((xh >= 0 && xh < w2) ? 0xffff0000u : 0u);
  }
  const bool any_col = (xb > -8) && (xb < w2);

  uint32_t row[8][4];
  if (ntx > 0) {
    // TILED slice (corr_volume.hip): row y1 of the window lives in tile row y1 >> 3; its 8 taps straddle the two
    // tiles tx0, tx0 + 1 -> two ALIGNED 16-byte loads (both inside the slice or replaced by a safe address and
    // zeroed) and a funnel shift by (xb & 7) halves.  The window touches <= 2 x 2 cache lines instead of 8-9.
    const int tx0 = xb >> 3, c0 = xb & 7;
    const bool okA = tx0 >= 0 && tx0 < ntx, okB = tx0 + 1 >= 0 && tx0 + 1 < ntx;
    const uint32_t sel1 = (c0 & 2) ? ~0u : 0u, sel2 = (c0 & 4) ? ~0u : 0u, sh16 = (c0 & 1) ? 16u : 0u;
    u32x4 A[8], B[8];
    bool rvt[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int y1 = yb + j;
      rvt[j] = any_col && y1 >= 0 && y1 < h2;
      const long base = slice_off + ((long)(y1 >> 3) * ntx + tx0) * 64 + (y1 & 7) * 8;
      A[j] = *reinterpret_cast<const u32x4*>(vol + ((rvt[j] && okA) ? base : slice_off));
      B[j] = *reinterpret_cast<const u32x4*>(vol + ((rvt[j] && okB) ? base + 64 : slice_off));
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint32_t D[9];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        D[k] = (rvt[j] && okA) ? A[j][k] : 0u;
        D[4 + k] = (rvt[j] && okB) ? B[j][k] : 0u;
      }
      D[8] = 0u;
      uint32_t Es[7], F[5];
#pragma unroll
      for (int i = 0; i < 7; i++) Es[i] = (D[i + 1] & sel1) | (D[i] & ~sel1);      // shift by one dword  (c0 & 2)
#pragma unroll
      for (int i = 0; i < 5; i++) F[i] = (Es[i + 2] & sel2) | (Es[i] & ~sel2);     // shift by two dwords (c0 & 4)
#pragma unroll
      for (int k = 0; k < 4; k++) row[j][k] = __builtin_amdgcn_alignbit(F[k + 1], F[k], sh16) & cm[k];  // one half (c0 & 1)
    }
  } else {
  // Issue all eight row loads unconditionally (no per-row branches, 8 x 16 B in flight per lane):
  // rows that are not real taps read the start of the pixel's own slice and are zeroed afterwards.
  bool rv[8];
  bool need_slow = false;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int y1 = yb + j;
    const long g0 = slice_off + (long)y1 * w2 + xb;
    const bool valid = any_col && y1 >= 0 && y1 < h2;
    const bool inb = g0 >= 0 && g0 + 8 <= total;
    rv[j] = valid && inb;
    need_slow |= valid && !inb;
    const Run16 r = *reinterpret_cast<const Run16*>(vol + (rv[j] ? g0 : slice_off));
    row[j][0] = r.d[0];
    row[j][1] = r.d[1];
    row[j][2] = r.d[2];
    row[j][3] = r.d[3];
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
#pragma unroll
    for (int k = 0; k < 4; k++) row[j][k] = rv[j] ? (row[j][k] & cm[k]) : 0u;
  }
  if (__builtin_expect(need_slow, 0)) {
    // a run pokes outside the tensor (first / last slice only): element-wise, bounds-checked
    const uint16_t* v16 = reinterpret_cast<const uint16_t*>(vol);
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
      const int y1 = yb + j;
      const long g0 = slice_off + (long)y1 * w2 + xb;
      const bool valid = any_col && y1 >= 0 && y1 < h2;
      if (valid && !(g0 >= 0 && g0 + 8 <= total)) {
        uint32_t t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int xl = xb + 2 * k;
          const uint32_t lo = (xl >= 0 && xl < w2) ? v16[g0 + 2 * k] : 0u;
          const uint32_t hi = (xl + 1 >= 0 && xl + 1 < w2) ? v16[g0 + 2 * k + 1] : 0u;
          t[k] = lo | (hi << 16);
        }
        // static indexing only (runtime-indexed register arrays would spill to scratch)
#pragma unroll
        for (int jj = 0; jj < 8; jj++)
          if (jj == j) {
            row[jj][0] = t[0]; row[jj][1] = t[1]; row[jj][2] = t[2]; row[jj][3] = t[3];
          }
      }
    }
  }
  }

  // weights, rounded to half exactly as `scalar_t(dx*dy)` etc. (correlation_kernels.cu:56-65)
  // The f32 product must be rounded to f32 BEFORE the conversion (the reference converts a float
  // expression); without the empty asm the compiler fuses mul+cvt into v_fma_mixlo_f16, which
  // rounds once and differs from the reference in double-rounding cases.
  float p00 = (1.0f - dx) * (1.0f - dy), p01 = (1.0f - dx) * dy, p10 = dx * (1.0f - dy), p11 = dx * dy;
  asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));
  const _Float16 w00 = (_Float16)p00, w01 = (_Float16)p01, w10 = (_Float16)p10, w11 = (_Float16)p11;
  const h2_t W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};

  // shifted rows: S[j][a] = T[a+1][j]
  uint32_t sh[8][4];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    sh[j][0] = __builtin_amdgcn_alignbit(row[j][1], row[j][0], 16);
    sh[j][1] = __builtin_amdgcn_alignbit(row[j][2], row[j][1], 16);
    sh[j][2] = __builtin_amdgcn_alignbit(row[j][3], row[j][2], 16);
    sh[j][3] = row[j][3] >> 16;
  }

  uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
#pragma unroll
  for (int b = 0; b < 7; b++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h2_t acc = as_h2(row[b][k]) * W00;           // tap (a  , b  )
      acc = acc + as_h2(row[b + 1][k]) * W01;      // tap (a  , b+1)
      acc = acc + as_h2(sh[b][k]) * W10;           // tap (a+1, b  )
      acc = acc + as_h2(sh[b + 1][k]) * W11;       // tap (a+1, b+1)
      const uint32_t u = as_u32(acc);
      const int a0 = 2 * k;
      o16[(long)(a0 * 7 + b) * HW1] = (uint16_t)(u & 0xffffu);
      if (a0 + 1 < 7) o16[(long)((a0 + 1) * 7 + b) * HW1] = (uint16_t)(u >> 16);
    }
  }
}

// Fused pyramid lookup: grid = (ceil(E*HW1/256), num_levels)
__global__ __launch_bounds__(256) void corr_lookup_pyramid_kernel(LookupLevels L, const float* __restrict__ coords,
                                                                  int interleaved, _Float16* __restrict__ out,
                                                                  int E, int HW1) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)E * HW1) return;
  const int lvl = blockIdx.y;
  const int n = (int)(idx / HW1);
  const int p = (int)(idx - (long)n * HW1);
  float cx, cy;
  if (interleaved) {
    const float2 c = *reinterpret_cast<const float2*>(coords + 2 * idx);
    cx = c.x;
    cy = c.y;
  } else {
    cx = coords[((long)n * 2 + 0) * HW1 + p];
    cy = coords[((long)n * 2 + 1) * HW1 + p];
  }
  const float s = L.scale[lvl];
  _Float16* o = out + ((long)n * (L.num_levels * 49) + lvl * 49) * HW1 + p;
  lookup_r3_f16(L.vol[lvl], idx * L.slice_elems[lvl], L.total_elems[lvl], L.h2[lvl], L.w2[lvl], L.ntx[lvl], cx * s, cy * s,
                o, HW1);
}

// ---------------------------------------------------------------------------------------------
// Cooperative variant of the fused pyramid lookup: EIGHT lanes per (edge, pixel, level), one per window row.
//
// The one-lane-per-pixel kernel above issues 8 (row-major) or 16 (tiled) 16-byte loads per lane, each its own L2
// request because every lane reads its own slice: at ~15 M requests per launch it sits on the L2 request rate
// (~140 G/s), not on bytes.  Here lane r of a pixel's 8-lane group owns ONE window row:
//   tiled level   : tile row r of the window's tile column pair -- rows yb..yb+7 wrap around the tile height, so the
//                   lanes r >= (yb & 7) read the upper tile pair and the others the lower one: 2 aligned loads per
//                   lane, and the 8 lanes of a pixel together read whole tile lines (2-4 L2 requests per pixel);
//   row-major level: window row r, one unaligned 16-byte load.
// The neighbouring row for the bilinear blend comes from the next lane of the group (one ds_bpermute per dword),
// every lane computes the 7 outputs of its row with the same packed-f16 operation order as above (bit-identical),
// and the 49 x 64 outputs of the workgroup's 64 pixels are staged in LDS and written as whole 128-byte channel-plane
// lines (8 lanes x 16 B), as before.
// ---------------------------------------------------------------------------------------------
#define COOP_PITCH 72  // halves per LDS plane row (64 pixels + pad; 144 B keeps 16-byte reads aligned)

// one level of one (edge, pixel) by its 8-lane group: lane r owns one window row; the 49 outputs of the pixel land in
// ob[(a * 7 + b) * CS + pl * PS] (plain kernel: 49 planes of 64 pixels, CS = COOP_PITCH, PS = 1; encoder-fused kernel: 64 pixel
// rows of 208 channels, CS = 1, PS = ENC_PP).  In three parts since round 6 -- geometry, loads, blend -- so that the fused kernel
// can have the NEXT level's (and the next pixel group's) loads in flight while it blends this one; the plain kernel calls them
// back to back (same instructions as the one-piece form it replaces, same bits).
struct CoopGeo {
  const _Float16* vol;
  long slice_off, total;
  float dx, dy;
  int h2, w2, ntx, xb, yb, j, y1;
  bool any_col, rowvalid;
};
struct CoopRaw {
  u32x4 A, B;   // tiled level: the two 8-value tile rows under the window row; row-major level: A = the 8-value run
};
__device__ __forceinline__ CoopGeo coop_geo(const LookupLevels& L, int lvl, float cx, float cy, bool live, long vidx, int t) {
  CoopGeo G;
  const int r = t & 7;
  const float sc = L.scale[lvl];
  const float x0 = cx * sc, y0 = cy * sc;
  G.h2 = L.h2[lvl];
  G.w2 = L.w2[lvl];
  G.ntx = L.ntx[lvl];
  G.vol = L.vol[lvl];
  G.slice_off = vidx * L.slice_elems[lvl];
  G.total = L.total_elems[lvl];
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const bool sane = live && (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  G.dx = sane ? x0 - fx0 : 0.0f;
  G.dy = sane ? y0 - fy0 : 0.0f;
  G.xb = sane ? (int)fx0 - 3 : -100000;
  G.yb = sane ? (int)fy0 - 3 : -100000;
  G.any_col = (G.xb > -8) && (G.xb < G.w2);
  // the window row this lane owns
  G.j = G.ntx > 0 ? ((r - G.yb) & 7) : r;
  G.y1 = G.yb + G.j;
  G.rowvalid = G.any_col && G.y1 >= 0 && G.y1 < G.h2;
  return G;
}
// the loads of the lane's window row: unconditional, from an address that is always inside the tensor
__device__ __forceinline__ void coop_issue(const CoopGeo& G, CoopRaw& raw) {
  if (G.ntx > 0) {
    const int tx0 = G.xb >> 3;
    const bool okA = G.rowvalid && tx0 >= 0 && tx0 < G.ntx, okB = G.rowvalid && tx0 + 1 >= 0 && tx0 + 1 < G.ntx;
    const long base = G.slice_off + ((long)(G.y1 >> 3) * G.ntx + tx0) * 64 + (G.y1 & 7) * 8;
    raw.A = *reinterpret_cast<const u32x4*>(G.vol + (okA ? base : G.slice_off));
    raw.B = *reinterpret_cast<const u32x4*>(G.vol + (okB ? base + 64 : G.slice_off));
  } else {
    const long g0 = G.slice_off + (long)G.y1 * G.w2 + G.xb;
    const bool inb = g0 >= 0 && g0 + 8 <= G.total;
    raw.A = (u32x4)0u;
    if (G.total >= 8) {      // (uniform)
      const long safe = G.slice_off + 8 <= G.total ? G.slice_off : G.total - 8;
      const Run16 rr = *reinterpret_cast<const Run16*>(G.vol + ((G.rowvalid && inb) ? g0 : safe));
      raw.A = (u32x4){rr.d[0], rr.d[1], rr.d[2], rr.d[3]};
    }
  }
}
template <int CS, int PS>
__device__ __forceinline__ void coop_finish(const CoopGeo& G, const CoopRaw& raw, int t, uint16_t* __restrict__ ob) {
  const int r = t & 7, pl = t >> 3;
  const int w2 = G.w2, xb = G.xb, j = G.j;
  const float dx = G.dx, dy = G.dy;
  uint32_t cm[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int xl = xb + 2 * k, xh = xl + 1;
    cm[k] = ((xl >= 0 && xl < w2) ? 0x0000ffffu : 0u) | ((xh >= 0 && xh < w2) ? 0xffff0000u : 0u);
  }
  uint32_t row[4];
  if (G.ntx > 0) {
    const int tx0 = xb >> 3, c0 = xb & 7;
    const bool okA = G.rowvalid && tx0 >= 0 && tx0 < G.ntx, okB = G.rowvalid && tx0 + 1 >= 0 && tx0 + 1 < G.ntx;
    uint32_t D[9];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      D[k] = okA ? raw.A[k] : 0u;
      D[4 + k] = okB ? raw.B[k] : 0u;
    }
    D[8] = 0u;
    const uint32_t sel1 = (c0 & 2) ? ~0u : 0u, sel2 = (c0 & 4) ? ~0u : 0u, sh16 = (c0 & 1) ? 16u : 0u;
    uint32_t Es[7], F[5];
#pragma unroll
    for (int i = 0; i < 7; i++) Es[i] = (D[i + 1] & sel1) | (D[i] & ~sel1);
#pragma unroll
    for (int i = 0; i < 5; i++) F[i] = (Es[i + 2] & sel2) | (Es[i] & ~sel2);
#pragma unroll
    for (int k = 0; k < 4; k++) row[k] = __builtin_amdgcn_alignbit(F[k + 1], F[k], sh16) & cm[k];
  } else {
    const long g0 = G.slice_off + (long)G.y1 * w2 + xb;
    const bool inb = g0 >= 0 && g0 + 8 <= G.total;
    if (G.rowvalid && inb) {
#pragma unroll
      for (int k = 0; k < 4; k++) row[k] = raw.A[k] & cm[k];
    } else if (G.rowvalid) {  // the run pokes outside the tensor (first / last slice only): element-wise, bounds-checked
      const uint16_t* v16 = reinterpret_cast<const uint16_t*>(G.vol);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int xl = xb + 2 * k;
        const uint32_t lo = (xl >= 0 && xl < w2) ? v16[g0 + 2 * k] : 0u;
        const uint32_t hi = (xl + 1 >= 0 && xl + 1 < w2) ? v16[g0 + 2 * k + 1] : 0u;
        row[k] = lo | (hi << 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) row[k] = 0u;
    }
  }
  // next window row from the next lane of the group
  uint32_t nxt[4];
  const int srcl = (t & 63 & ~7) | ((r + 1) & 7);
#pragma unroll
  for (int k = 0; k < 4; k++) nxt[k] = (uint32_t)__shfl((int)row[k], srcl);
  float p00 = (1.0f - dx) * (1.0f - dy), p01 = (1.0f - dx) * dy, p10 = dx * (1.0f - dy), p11 = dx * dy;
  asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));  // no mul+cvt fusion (see lookup_r3_f16)
  const _Float16 w00 = (_Float16)p00, w01 = (_Float16)p01, w10 = (_Float16)p10, w11 = (_Float16)p11;
  const h2_t W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
  uint32_t sh[4], shn[4];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    sh[k] = __builtin_amdgcn_alignbit(row[k + 1], row[k], 16);
    shn[k] = __builtin_amdgcn_alignbit(nxt[k + 1], nxt[k], 16);
  }
  sh[3] = row[3] >> 16;
  shn[3] = nxt[3] >> 16;
  if (j < 7) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h2_t acc = as_h2(row[k]) * W00;        // tap (a  , b  )
      acc = acc + as_h2(nxt[k]) * W01;       // tap (a  , b+1)
      acc = acc + as_h2(sh[k]) * W10;        // tap (a+1, b  )
      acc = acc + as_h2(shn[k]) * W11;       // tap (a+1, b+1)
      const uint32_t u = as_u32(acc);
      const int a0 = 2 * k;
      ob[(a0 * 7 + j) * CS + pl * PS] = (uint16_t)(u & 0xffffu);
      if (a0 + 1 < 7) ob[((a0 + 1) * 7 + j) * CS + pl * PS] = (uint16_t)(u >> 16);
    }
  }
}
__device__ __forceinline__ void coop_level(const LookupLevels& L, int lvl, float cx, float cy, bool live, long vidx, int t,
                                           uint16_t* __restrict__ ob) {
  const CoopGeo G = coop_geo(L, lvl, cx, cy, live, vidx, t);
  CoopRaw raw;
  coop_issue(G, raw);
  coop_finish<COOP_PITCH, 1>(G, raw, t, ob);
}

__global__ __launch_bounds__(512) void corr_lookup_coop_kernel(LookupLevels L, const float* __restrict__ coords,
                                                               int interleaved, _Float16* __restrict__ out, int E,
                                                               int HW1, const int* __restrict__ slot) {
  __shared__ __attribute__((aligned(16))) uint16_t ob[49 * COOP_PITCH];
  const int t = threadIdx.x, pl = t >> 3;
  const int lvl = blockIdx.y;
  const long idx0 = (long)blockIdx.x * 64;
  const long total_px = (long)E * HW1;
  const long idx = idx0 + pl;
  const bool live = idx < total_px;
  const long idc = live ? idx : total_px - 1;
  const int n = (int)(idc / HW1);
  const int p = (int)(idc - (long)n * HW1);
  float cx, cy;
  if (interleaved) {
    const float2 c = *reinterpret_cast<const float2*>(coords + 2 * idc);
    cx = c.x;
    cy = c.y;
  } else {
    cx = coords[((long)n * 2 + 0) * HW1 + p];
    cy = coords[((long)n * 2 + 1) * HW1 + p];
  }
  // slot-addressed pools: edge n reads volume index slot[n] (the edge list is reordered / shrunk without moving volumes)
  const long vidx = slot ? (long)slot[n] * HW1 + p : idc;
  coop_level(L, lvl, cx, cy, live, vidx, t, ob);
  __syncthreads();
  // 49 channel planes x 64 pixels -> whole lines where the 8 pixels of a piece are consecutive in one plane
  uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
  for (int q = t; q < 49 * 8; q += 512) {
    const int ch = q >> 3, piece = q & 7;
    const long i0 = idx0 + piece * 8;
    if (i0 >= total_px) continue;
    const int n0 = (int)(i0 / HW1);
    const int q0 = (int)(i0 - (long)n0 * HW1);
    const uint16_t* src = ob + ch * COOP_PITCH + piece * 8;
    if (q0 + 8 <= HW1 && i0 + 8 <= total_px) {
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      Run16 w;
      w.d[0] = v.x; w.d[1] = v.y; w.d[2] = v.z; w.d[3] = v.w;
      *reinterpret_cast<Run16*>(o16 + ((long)n0 * (L.num_levels * 49) + lvl * 49 + ch) * HW1 + q0) = w;
    } else {
      for (int e = 0; e < 8; e++) {
        const long ie = i0 + e;
        if (ie >= total_px) break;
        const int ne = (int)(ie / HW1);
        const int pe = (int)(ie - (long)ne * HW1);
        o16[((long)ne * (L.num_levels * 49) + lvl * 49 + ch) * HW1 + pe] = src[e];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Lookup + correlation encoder in ONE launch (round 4; SURVEY 8(f) row 2: "fuse corr-encoder 1x1 conv onto the K12 output").
// Reference chain: CorrBlock.__call__ (networks/modules/corr.py:40-50) -> UpdateModule.corr_encoder[0:2] = Conv2d(196, 128, 1)
// + ReLU (networks/droid_net.py:83-87, applied at :133).  Unfused, the lookup writes [E,196,ht,wd] f16 (90 MB at E = 48, 42 % of
// its algorithmic bytes), a transposition kernel re-reads and re-writes it channels-last, and the 1x1 convolution reads it a
// third time.  Here a workgroup keeps the 196 values of its 64 pixels in LDS (all four levels, the same 8-lane rows as above,
// bit-identical values), multiplies them by the 196 x 128 weight matrix on the matrix cores -- out[64 px][128] =
// X[64][208] W^T[208][128], 13 k-chunks, one 32 x 32 output tile per wave, the wave's 13 weight fragments resident in
// registers across the workgroup's pixel groups -- adds the bias, applies the ReLU and writes [E,ht,wd,128] f16 channels-last:
// what the second encoder convolution reads.  f32 accumulation like the convolution kernel it replaces; equal to
// corr1(lookup) up to the f16 rounding of the output (tests/test_corr_gpu.py::test_lookup_fused_with_the_correlation_encoder).
// ---------------------------------------------------------------------------------------------
typedef _Float16 lk_f16x8 __attribute__((ext_vector_type(8)));
typedef float lk_f32x16 __attribute__((ext_vector_type(16)));
#define ENC_K 208        // 196 lookup channels padded to 13 chunks of 16
#define ENC_GROUPS 4     // 64-pixel groups per workgroup (the weight fragments are loaded once per workgroup)
#define ENC_PP 216       // halves per pixel row of the LDS tile: 208 channels + 8 (432 B = 27 x 16 B: the 16-byte fragment reads of
                         // 32 consecutive pixel rows fall on different bank groups)

// Round 6 (91 us against a 57-us traffic floor, 0.51 of the HBM peak).  The tile is [pixel][channel] now: an
// A fragment of the matrix-core phase is ONE 16-byte LDS read (round 4 kept [channel][pixel], the plain kernel's output order,
// and gathered a fragment with eight two-byte reads: 104 LDS instructions per lane and group): 92.2 / 92.4 / 96.4 -> 88.6 / 86.3 /
// 89.0 us per 48-edge launch, same box, interleaved, same bits.  Measured with it and NOT kept: the loads of level l + 1 in
// flight while level l is blended, and the next group's coordinates, slot and level-0 loads requested before the matrix-core
// phase (coop_geo / coop_issue / coop_finish exist for that; barriers waiting for LDS only): 128 registers + 32 B of scratch,
// 96.4-98.3 us -- the kernel is not waiting on a chain of round trips (round 4 found the same with all four levels up front).
__global__ __launch_bounds__(512, 4) void corr_lookup_enc_kernel(LookupLevels L, const float* __restrict__ coords, int interleaved,
                                                              const lk_f16x8* __restrict__ wfrag, const float* __restrict__ bias,
                                                              _Float16* __restrict__ out, int E, int HW1,
                                                              const int* __restrict__ slot) {
  __shared__ __attribute__((aligned(16))) uint16_t ob[64 * ENC_PP];   // 27 KB: [64 pixels][208 channels + pad]
  const int t = threadIdx.x, pl = t >> 3;
  const int lane = t & 63, wave = t >> 6, j = lane & 31, h = lane >> 5;
  const int mt = wave & 1, nt = wave >> 1;             // the wave's output tile: pixels 32 mt.., channels 32 nt..
  const long total_px = (long)E * HW1;
  for (int e = t; e < 64 * (ENC_PP - 196); e += 512) ob[(e / (ENC_PP - 196)) * ENC_PP + 196 + e % (ENC_PP - 196)] = 0;   // pad channels
  lk_f16x8 Bf[13];
#pragma unroll
  for (int c = 0; c < 13; c++) Bf[c] = wfrag[(nt * 13 + c) * 64 + lane];
  const float bj = bias[32 * nt + j];
  // the lane's pixel of a group: coordinates + volume index
  struct Px { float cx, cy; long vidx; bool live; };
  auto pixel_of = [&](int g) {
    Px P;
    const long idx = ((long)blockIdx.x * ENC_GROUPS + g) * 64 + pl;
    P.live = idx < total_px;
    const long idc = P.live ? idx : total_px - 1;
    const int n = (int)(idc / HW1);
    const int p = (int)(idc - (long)n * HW1);
    if (interleaved) {
      const float2 c = *reinterpret_cast<const float2*>(coords + 2 * idc);
      P.cx = c.x;
      P.cy = c.y;
    } else {
      P.cx = coords[((long)n * 2 + 0) * HW1 + p];
      P.cy = coords[((long)n * 2 + 1) * HW1 + p];
    }
    // slot-addressed pools: edge n reads volume index slot[n] (the edge list is reordered / shrunk without moving volumes)
    P.vidx = slot ? (long)slot[n] * HW1 + p : idc;
    return P;
  };
  CoopRaw raw;
  for (int g = 0; g < ENC_GROUPS; g++) {
    const long idx0 = ((long)blockIdx.x * ENC_GROUPS + g) * 64;
    if (idx0 >= total_px) break;                       // (uniform)
    const Px cur = pixel_of(g);
    lds_barrier();                                     // the previous group's fragments have been read (and the pad zeroed); its
                                                       // output stores stay in flight (LDS-only barrier)
#pragma unroll 1
    for (int lvl = 0; lvl < 4; lvl++) {
      const CoopGeo G = coop_geo(L, lvl, cur.cx, cur.cy, cur.live, cur.vidx, t);
      coop_issue(G, raw);
      coop_finish<1, ENC_PP>(G, raw, t, ob + lvl * 49);
    }
    lds_barrier();
    lk_f32x16 acc = (lk_f32x16)0.0f;
    const uint16_t* xa = ob + (32 * mt + j) * ENC_PP + 8 * h;        // A: row = pixel 32 mt + j, k = channel 16 c + 8 h + q
#pragma unroll
    for (int c = 0; c < 13; c++) {
      const lk_f16x8 a = *reinterpret_cast<const lk_f16x8*>(xa + 16 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bf[c], acc, 0, 0, 0);
    }
    // acc[r]: pixel 32 mt + 4 h + (r & 3) + 8 (r >> 2) of the group, channel 32 nt + j
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const long px = idx0 + 32 * mt + 4 * h + (r & 3) + 8 * (r >> 2);
      if (px < total_px) out[px * 128 + 32 * nt + j] = (_Float16)fmaxf(acc[r] + bj, 0.0f);
    }
  }
}

// Generic single-level kernel (any radius, f16 or f32): one lane per pixel, scalar taps.
// Same arithmetic order as the reference; used by the drop-in op when radius != 3 or dtype f32.
template <typename T>
__global__ __launch_bounds__(256) void corr_index_generic_kernel(const T* __restrict__ vol,
                                                                 const float* __restrict__ coords,
                                                                 T* __restrict__ out, int B, int HW1, int h2, int w2,
                                                                 int r) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW1) return;
  const int n = (int)(idx / HW1);
  const int p = (int)(idx - (long)n * HW1);
  const float x0 = coords[((long)n * 2 + 0) * HW1 + p];
  const float y0 = coords[((long)n * 2 + 1) * HW1 + p];
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const float dx = sane ? x0 - fx0 : 0.0f, dy = sane ? y0 - fy0 : 0.0f;
  const int xb = sane ? (int)fx0 - r : -100000;
  const int yb = sane ? (int)fy0 - r : -100000;
  const int rd = 2 * r + 1;
  float p00 = (1.0f - dx) * (1.0f - dy), p01 = (1.0f - dx) * dy, p10 = dx * (1.0f - dy), p11 = dx * dy;
  asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));  // no mul+cvt fusion (see lookup_r3_f16)
  const T w00 = (T)p00, w01 = (T)p01, w10 = (T)p10, w11 = (T)p11;
  const T* v = vol + idx * (long)h2 * w2;
  T* o = out + (long)n * rd * rd * HW1 + p;
  auto tap = [&](int i, int j) -> T {
    const int x1 = xb + i, y1 = yb + j;
    return (x1 >= 0 && x1 < w2 && y1 >= 0 && y1 < h2) ? v[(long)y1 * w2 + x1] : (T)0;
  };
  for (int a = 0; a < rd; a++)
    for (int b = 0; b < rd; b++) {
      T acc;
      if constexpr (sizeof(T) == 2) {
        acc = tap(a, b) * w00;
        acc = acc + tap(a, b + 1) * w01;
        acc = acc + tap(a + 1, b) * w10;
        acc = acc + tap(a + 1, b + 1) * w11;
      } else {
        // float: `corr += s*w` contracts to an FMA under nvcc's default -fmad=true
        acc = __builtin_fmaf(tap(a, b), w00, 0.0f);
        acc = __builtin_fmaf(tap(a, b + 1), w01, acc);
        acc = __builtin_fmaf(tap(a + 1, b), w10, acc);
        acc = __builtin_fmaf(tap(a + 1, b + 1), w11, acc);
      }
      o[(long)(a * rd + b) * HW1] = acc;
    }
}

// corr_index_backward_kernel<float>: one lane per pixel, each lane owns its private slice of
// volume_grad, so the `+=` needs no atomics (same as the reference).
__global__ __launch_bounds__(256) void corr_index_backward_kernel(const float* __restrict__ coords,
                                                                  const float* __restrict__ cg,
                                                                  float* __restrict__ vg, int B, int HW1, int h2,
                                                                  int w2, int r) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW1) return;
  const int n = (int)(idx / HW1);
  const int p = (int)(idx - (long)n * HW1);
  const float x0 = coords[((long)n * 2 + 0) * HW1 + p];
  const float y0 = coords[((long)n * 2 + 1) * HW1 + p];
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const float dx = x0 - fx0, dy = y0 - fy0;
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const int xb = sane ? (int)fx0 - r : -100000;
  const int yb = sane ? (int)fy0 - r : -100000;
  const int rd = 2 * r + 1;
  const float* g = cg + (long)n * rd * rd * HW1 + p;
  float* v = vg + idx * (long)h2 * w2;
  for (int i = 0; i < rd + 1; i++)
    for (int j = 0; j < rd + 1; j++) {
      const int x1 = xb + i, y1 = yb + j;
      if (x1 >= 0 && x1 < w2 && y1 >= 0 && y1 < h2) {
        float acc = 0.0f;
        if (i > 0 && j > 0) acc += g[(long)((i - 1) * rd + (j - 1)) * HW1] * (dx * dy);
        if (i > 0 && j < rd) acc += g[(long)((i - 1) * rd + j) * HW1] * (dx * (1.0f - dy));
        if (i < rd && j > 0) acc += g[(long)(i * rd + (j - 1)) * HW1] * ((1.0f - dx) * dy);
        if (i < rd && j < rd) acc += g[(long)(i * rd + j) * HW1] * ((1.0f - dx) * (1.0f - dy));
        v[(long)y1 * w2 + x1] += acc;
      }
    }
}

extern "C" int ns_corr_lookup_pyramid(const void* const* pyr_host, int num_levels, const float* coords,
                                      int coords_interleaved, void* out, int E, int h1, int w1, int tiled,
                                      void* stream) {
  return ns_corr_lookup_pyramid_slots(pyr_host, num_levels, coords, coords_interleaved, out, E, h1, w1, tiled, nullptr, E,
                                      stream);
}

extern "C" int ns_corr_lookup_pyramid_slots(const void* const* pyr_host, int num_levels, const float* coords,
                                            int coords_interleaved, void* out, int E, int h1, int w1, int tiled,
                                            const int* slot, int capacity, void* stream) {
  if (E == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(slot == nullptr || capacity >= 1, "ns_corr_lookup_pyramid_slots: capacity of the volume pool must be given");
  NS_REQUIRE(pyr_host && coords && out, "ns_corr_lookup_pyramid: null pointer");
  NS_REQUIRE(num_levels >= 1 && num_levels <= 4, "ns_corr_lookup_pyramid: num_levels=%d not in 1..4", num_levels);
  NS_REQUIRE(E >= 0 && h1 > 0 && w1 > 0, "ns_corr_lookup_pyramid: bad shape E=%d h1=%d w1=%d", E, h1, w1);
  if (E == 0) return NS_OK;
  LookupLevels L;
  L.num_levels = num_levels;
  const long HW1 = (long)h1 * w1;
  for (int l = 0; l < 4; l++) {
    const int ll = l < num_levels ? l : num_levels - 1;
    L.vol[l] = (const _Float16*)pyr_host[ll];
    NS_REQUIRE(L.vol[l] != nullptr, "ns_corr_lookup_pyramid: pyr[%d] is null", ll);
    L.h2[l] = h1 >> ll;
    L.w2[l] = w1 >> ll;
    L.scale[l] = 1.0f / (float)(1 << ll);
    L.ntx[l] = (tiled && ll < 2) ? (L.w2[l] + 7) / 8 : 0;
    L.slice_elems[l] = L.ntx[l] ? (long)((L.h2[l] + 7) / 8) * L.ntx[l] * 64 : (long)L.h2[l] * L.w2[l];
    L.total_elems[l] = (long)(slot ? capacity : E) * HW1 * L.slice_elems[l];
    NS_REQUIRE(L.h2[l] > 0 && L.w2[l] > 0, "ns_corr_lookup_pyramid: level %d is empty", ll);
  }
#ifdef NS_TEST_VARIANTS
  static const bool one_lane = ns_variant_env("NS_LOOKUP_ONE_LANE") != nullptr;  // comparison switch: one lane per pixel
  if (one_lane && slot == nullptr) {
    dim3 grid1(ns_cdiv((long)E * HW1, 256), num_levels);
    hipLaunchKernelGGL(corr_lookup_pyramid_kernel, grid1, dim3(256), 0, (hipStream_t)stream, L, coords,
                       coords_interleaved, (_Float16*)out, E, (int)HW1);
    NS_CHECK_LAUNCH("corr_lookup_pyramid_kernel");
    return NS_OK;
  }
#endif
  dim3 grid(ns_cdiv((long)E * HW1, 64), num_levels);
  hipLaunchKernelGGL(corr_lookup_coop_kernel, grid, dim3(512), 0, (hipStream_t)stream, L, coords, coords_interleaved,
                     (_Float16*)out, E, (int)HW1, slot);
  NS_CHECK_LAUNCH("corr_lookup_coop_kernel");
  return NS_OK;
}

extern "C" int ns_corr_lookup_encode_slots(const void* const* pyr_host, const float* coords, int coords_interleaved,
                                           const void* wfrag, const float* bias, void* out, int E, int h1, int w1, int tiled,
                                           const int* slot, int capacity, void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(slot == nullptr || capacity >= 1, "ns_corr_lookup_encode_slots: capacity of the volume pool must be given");
  NS_REQUIRE(pyr_host && coords && out && wfrag && bias, "ns_corr_lookup_encode_slots: null pointer");
  NS_REQUIRE(E >= 0 && h1 > 0 && w1 > 0, "ns_corr_lookup_encode_slots: bad shape E=%d h1=%d w1=%d", E, h1, w1);
  NS_REQUIRE(((uintptr_t)wfrag % 16) == 0, "ns_corr_lookup_encode_slots: the weight fragments must be 16-byte aligned");
  LookupLevels L;
  L.num_levels = 4;
  const long HW1 = (long)h1 * w1;
  for (int l = 0; l < 4; l++) {
    L.vol[l] = (const _Float16*)pyr_host[l];
    NS_REQUIRE(L.vol[l] != nullptr, "ns_corr_lookup_encode_slots: pyr[%d] is null", l);
    L.h2[l] = h1 >> l;
    L.w2[l] = w1 >> l;
    L.scale[l] = 1.0f / (float)(1 << l);
    L.ntx[l] = (tiled && l < 2) ? (L.w2[l] + 7) / 8 : 0;
    L.slice_elems[l] = L.ntx[l] ? (long)((L.h2[l] + 7) / 8) * L.ntx[l] * 64 : (long)L.h2[l] * L.w2[l];
    L.total_elems[l] = (long)(slot ? capacity : E) * HW1 * L.slice_elems[l];
    NS_REQUIRE(L.h2[l] > 0 && L.w2[l] > 0, "ns_corr_lookup_encode_slots: level %d is empty (four levels are required)", l);
  }
  hipLaunchKernelGGL(corr_lookup_enc_kernel, dim3(ns_cdiv((long)E * HW1, 64 * ENC_GROUPS)), dim3(512), 0, (hipStream_t)stream, L,
                     coords, coords_interleaved, (const lk_f16x8*)wfrag, bias, (_Float16*)out, E, (int)HW1, slot);
  NS_CHECK_LAUNCH("corr_lookup_enc_kernel");
  return NS_OK;
}

extern "C" int ns_corr_index_forward(const void* volume, const float* coords, void* corr, int dtype, int B, int h1,
                                     int w1, int h2, int w2, int radius, void* stream) {
  if (B == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(volume && coords && corr, "ns_corr_index_forward: null pointer");
  NS_REQUIRE(dtype == NS_F16 || dtype == NS_F32, "ns_corr_index_forward: dtype %d unsupported", dtype);
  NS_REQUIRE(B >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && radius >= 0,
             "ns_corr_index_forward: bad shape B=%d h1=%d w1=%d h2=%d w2=%d r=%d", B, h1, w1, h2, w2, radius);
  if (B == 0) return NS_OK;
  const long HW1 = (long)h1 * w1;
  if (dtype == NS_F16 && radius == 3) {
    LookupLevels L;
    L.num_levels = 1;
    for (int l = 0; l < 4; l++) {
      L.vol[l] = (const _Float16*)volume;
      L.h2[l] = h2;
      L.w2[l] = w2;
      L.scale[l] = 1.0f;
      L.ntx[l] = 0;
      L.slice_elems[l] = (long)h2 * w2;
      L.total_elems[l] = (long)B * HW1 * h2 * w2;
    }
    dim3 grid(ns_cdiv((long)B * HW1, 256), 1);
    hipLaunchKernelGGL(corr_lookup_pyramid_kernel, grid, dim3(256), 0, (hipStream_t)stream, L, coords, 0,
                       (_Float16*)corr, B, (int)HW1);
    NS_CHECK_LAUNCH("corr_lookup_pyramid_kernel");
    return NS_OK;
  }
  dim3 grid(ns_cdiv((long)B * HW1, 256));
  if (dtype == NS_F16)
    hipLaunchKernelGGL(corr_index_generic_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)volume, coords, (_Float16*)corr, B, (int)HW1, h2, w2, radius);
  else
    hipLaunchKernelGGL(corr_index_generic_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)volume, coords, (float*)corr, B, (int)HW1, h2, w2, radius);
  NS_CHECK_LAUNCH("corr_index_generic_kernel");
  return NS_OK;
}

extern "C" int ns_corr_index_backward(const float* coords, const float* corr_grad, float* volume_grad, int B, int h1,
                                      int w1, int h2, int w2, int radius, void* stream) {
  if (B == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(coords && corr_grad && volume_grad, "ns_corr_index_backward: null pointer");
  NS_REQUIRE(B >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && radius >= 0, "ns_corr_index_backward: bad shape");
  if (B == 0) return NS_OK;
  dim3 grid(ns_cdiv((long)B * h1 * w1, 256));
  hipLaunchKernelGGL(corr_index_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, coords, corr_grad,
                     volume_grad, B, h1 * w1, h2, w2, radius);
  NS_CHECK_LAUNCH("corr_index_backward_kernel");
  return NS_OK;
}
