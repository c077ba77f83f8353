// altcorr.hip -- on-the-fly correlation lookup (no HW^2 volume) for gfx950.
//
// Replaces altcorr_forward_kernel (src/altcorr_kernel.cu:28-149; caller networks/modules/corr.py:92-140).
// The reference runs 32-thread blocks (half a CDNA wavefront), stages 32x32 tiles through shared
// memory with one barrier per tap and does 4 global read-modify-writes per tap and channel slab.
//
// Here one wave64 owns one (edge, pixel) at a time:
//   * channels-last features make a window row (8 taps x C floats) ONE contiguous 4 KiB run, so
//     the 8x8 window is fetched as 32 fully coalesced 1 KiB wave loads (lane = 4 channels of a tap);
//   * each lane forms 32 four-channel partial dot products; a 31-step reduce-scatter butterfly
//     inside each 32-lane half (instead of 32 x 5 full reductions) leaves exactly one finished tap
//     per lane;
//   * the bilinear blend of the 4 neighbouring taps is three lane permutes; 49 lanes store.
// No LDS, no barriers, no atomics, no zero-fill.  Arithmetic is f32 like the reference's float
// dispatch (corr.py:121); summation order differs from the reference's 32-channel slabs, so the
// parity tolerance is relative 1e-5 of max|corr| (tests/test_altcorr.py).
#include <cstdlib>

#include "common.h"

#define ALT_PIX_PER_WAVE 8

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// One (edge, pixel, level) by one wave: f1 = 128-channel-slab feature of the pixel, f2 = feature map
// of the target frame at this level.  `out` points at channel 0 of this pixel, channel stride cs.
__device__ __forceinline__ void altcorr_pixel(const float* __restrict__ f1, const float* __restrict__ f2b, int H2,
                                              int W2, int C, float x2, float y2, float* __restrict__ out, long cs,
                                              int lane) {
  const int half = lane >> 5;   // which of the two interleaved tap columns
  const int cl = lane & 31;     // channel quad inside a 128-channel slab
  // output lane layout == finished-tap layout: y = cl>>2, x = half + 2*(cl&3)
  const int oy = cl >> 2, ox = half + 2 * (cl & 3);
  // lanes holding taps (y, x+1), (y+1, x), (y+1, x+1)
  const int src_x1 = (half == 0) ? lane + 32 : lane - 32 + 1;
  const int src_y1 = lane + 4;
  const int src_xy = src_x1 + 4;

  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const float fx0 = floorf(x2), fy0 = floorf(y2);
  const float dx = sane ? x2 - fx0 : 0.0f, dy = sane ? y2 - fy0 : 0.0f;
  const int xb = sane ? (int)fx0 - 3 : -100000;
  const int yb = sane ? (int)fy0 - 3 : -100000;

  float p[32];
#pragma unroll
  for (int v = 0; v < 32; v++) p[v] = 0.0f;

  for (int c0 = 0; c0 < C; c0 += 128) {
    const int ch = c0 + 4 * cl;
    const bool chok = ch < C;  // C % 4 == 0 is required by the entry points
    const float4 a = chok ? *reinterpret_cast<const float4*>(f1 + ch) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int iy = 0; iy < 8; iy++) {
      const int h2 = yb + iy;
      const bool rowok = chok && h2 >= 0 && h2 < H2;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int w2 = xb + half + 2 * k;
        if (rowok && w2 >= 0 && w2 < W2) {
          const float4 f = *reinterpret_cast<const float4*>(f2b + ((long)h2 * W2 + w2) * C + ch);
          p[iy * 4 + k] += dot4(a, f);
        }
      }
    }
  }

  // reduce-scatter over the 32 lanes of this half: after the step on lane bit s, a lane keeps the
  // values whose index has bit s equal to its own; after 5 steps lane cl holds the total of value cl.
#pragma unroll
  for (int s = 4; s >= 0; s--) {
    const int cnt = 1 << s;
    const bool up = (cl >> s) & 1;
#pragma unroll
    for (int i = 0; i < cnt; i++) {
      const float lo = p[i], hi = p[i + cnt];
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      p[i] = keep + __shfl_xor(send, cnt, 64);
    }
  }
  const float s00 = p[0];  // tap (oy, ox)
  const float s01 = __shfl(s00, src_x1, 64);
  const float s10 = __shfl(s00, src_y1, 64);
  const float s11 = __shfl(s00, src_xy, 64);
  if (oy < 7 && ox < 7) {
    // reference weights (altcorr_kernel.cu:112-115): se, sw, ne, nw as seen from the output cell
    const float val = s00 * ((1.0f - dy) * (1.0f - dx)) + s01 * ((1.0f - dy) * dx) + s10 * (dy * (1.0f - dx)) +
                      s11 * (dy * dx);
    out[(long)(oy + 7 * ox) * cs] = val;  // channel = iy + 7*ix (:102-105)
  }
}

// drop-in op: pre-gathered per-edge feature maps, one level, N coordinate sets
__global__ __launch_bounds__(256, 4) void altcorr_forward_kernel(const float* __restrict__ fmap1,
                                                              const float* __restrict__ fmap2,
                                                              const float* __restrict__ coords,
                                                              float* __restrict__ corr, int B, int H1, int W1,
                                                              int H2, int W2, int C, int N) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long HW1 = (long)H1 * W1;
  const long ntask = (long)B * N * HW1;
  const long first = ((long)blockIdx.x * 4 + wave) * ALT_PIX_PER_WAVE;
  for (int t = 0; t < ALT_PIX_PER_WAVE; t++) {
    const long task = first + t;
    if (task >= ntask) return;  // wave-uniform
    const long pix = task % HW1;
    const long bn = task / HW1;
    const int b = (int)(bn / N);
    const float2 c = *reinterpret_cast<const float2*>(coords + task * 2);
    altcorr_pixel(fmap1 + ((long)b * HW1 + pix) * C, fmap2 + (long)b * H2 * W2 * C, H2, W2, C, c.x, c.y,
                  corr + bn * 49 * HW1 + pix, HW1, lane);
  }
}

// Fused AltCorrBlock.__call__ (networks/modules/corr.py:107-126): all pyramid levels in one launch,
// feature maps addressed by frame index (no per-edge gather copies, no .float() copies, no cat).
struct AltPyramid {
  const float* fmap[4];  // level l: [nframes, H>>l, W>>l, C] channels-last f32 (already / 4)
  int num_levels;
};

__global__ __launch_bounds__(256, 4) void altcorr_pyramid_kernel(AltPyramid P, const int64_t* __restrict__ ii,
                                                              const int64_t* __restrict__ jj,
                                                              const float* __restrict__ coords,
                                                              float* __restrict__ out, int E, int H1, int W1, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lvl = blockIdx.y;
  const long HW1 = (long)H1 * W1;
  const long ntask = (long)E * HW1;
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const float scale = 1.0f / (float)(1 << lvl);
  const long first = ((long)blockIdx.x * 4 + wave) * ALT_PIX_PER_WAVE;
  for (int t = 0; t < ALT_PIX_PER_WAVE; t++) {
    const long task = first + t;
    if (task >= ntask) return;
    const long pix = task % HW1;
    const int e = (int)(task / HW1);
    const long fi = ii[e], fj = jj[e];
    const float2 c = *reinterpret_cast<const float2*>(coords + task * 2);
    altcorr_pixel(P.fmap[0] + (fi * HW1 + pix) * C, P.fmap[lvl] + fj * (long)H2 * W2 * C, H2, W2, C, c.x * scale,
                  c.y * scale, out + ((long)e * P.num_levels * 49 + lvl * 49) * HW1 + pix, HW1, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// Tiled variant of the fused pyramid kernel.  The wave-per-pixel kernel above reads a private 8x8x128-channel window
// (32 KiB) per (edge, pixel, level) through L1/L2: 30 GB per 48-edge call, i.e. it runs at the L2 bandwidth (~13 TB/s,
// 2.3 ms).  Neighbouring pixels of a smooth flow field look at nearly the same target region, so here a workgroup owns
// an 8x8 SOURCE-PIXEL TILE of one (edge, level): it stages the union of the tile's windows (bounding box, clipped to the
// image) through LDS one 16-channel slab at a time, and thread (pixel p, quarter q) accumulates the 16 raw taps of window
// rows 2q, 2q+1 of its pixel across the slabs.  The raw
// 8x8 taps then meet in LDS for the bilinear blend.  Tiles whose bounding box does not fit (wild flow) fall back to the
// wave-per-pixel routine inside the same launch.  L2 traffic drops ~10x; the kernel becomes LDS-read bound.
// ---------------------------------------------------------------------------------------------
#define AT_MAXR 640   // staged region, pixels (25 x 25; x 20 floats pitch = 50 KiB)
#define AT_SLAB 16    // channels staged at a time
#define AT_PITCH 20   // floats per staged pixel and slab (16-byte aligned rows)
#define AT_TAPP 65    // pitch of the raw-tap table

__global__ __launch_bounds__(256) void altcorr_tile_kernel(AltPyramid P, const int64_t* __restrict__ ii,
                                                           const int64_t* __restrict__ jj,
                                                           const float* __restrict__ coords, float* __restrict__ out,
                                                           int E, int H1, int W1, int C) {
  __shared__ __attribute__((aligned(16))) float region[AT_MAXR * AT_PITCH];
  __shared__ int bbox[4];
  float* taps = region;  // the raw-tap table reuses the staging buffer after the last slab (64 x 65 floats)
  const int tid = threadIdx.x, p = tid >> 2, q = tid & 3;
  const int lvl = blockIdx.y, e = blockIdx.z;
  const int ntx = (W1 + 7) >> 3;
  const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
  const long HW1 = (long)H1 * W1;
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const float scale = 1.0f / (float)(1 << lvl);
  const long fi = ii[e], fj = jj[e];
  const float* __restrict__ f2 = P.fmap[lvl] + fj * (long)H2 * W2 * C;
  float* __restrict__ obase = out + ((long)e * P.num_levels * 49 + lvl * 49) * HW1;
  const int py = 8 * ty + (p >> 3), px = 8 * tx + (p & 7);
  const bool inimg = py < H1 && px < W1;
  const long pix = inimg ? (long)py * W1 + px : 0;
  float x2 = 0.0f, y2 = 0.0f;
  if (inimg) {
    const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + pix) * 2);
    x2 = c.x * scale;
    y2 = c.y * scale;
  }
  const bool sane = inimg && (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const float fx0 = floorf(x2), fy0 = floorf(y2);
  const float dx = sane ? x2 - fx0 : 0.0f, dy = sane ? y2 - fy0 : 0.0f;
  const int xb = sane ? (int)fx0 - 3 : -100000;
  const int yb = sane ? (int)fy0 - 3 : -100000;
  // bounding box of the windows that touch the image at all
  if (tid < 4) bbox[tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;
  __syncthreads();
  const bool touches = sane && xb > -8 && xb < W2 && yb > -8 && yb < H2;
  if (q == 0 && touches) {
    atomicMin(&bbox[0], max(xb, 0));
    atomicMin(&bbox[1], max(yb, 0));
    atomicMax(&bbox[2], min(xb + 8, W2));
    atomicMax(&bbox[3], min(yb + 8, H2));
  }
  __syncthreads();
  // (an untouched box keeps its +/-INT_MAX sentinels, whose difference wraps to +2: test the sentinel, not the difference)
  const bool empty = bbox[0] == 0x7fffffff || bbox[2] == -0x7fffffff;
  const int x0 = bbox[0], y0 = bbox[1], RW = empty ? 0 : bbox[2] - bbox[0], RH = empty ? 0 : bbox[3] - bbox[1];
  if (RW <= 0 || RH <= 0) {  // nothing of this tile looks into the image: zeros
    if (inimg && q == 0)
      for (int ch = 0; ch < 49; ch++) obase[(long)ch * HW1 + pix] = 0.0f;
    return;
  }
  if ((long)RW * RH > AT_MAXR || (C % AT_SLAB) != 0) {  // workgroup-uniform: the union does not fit -> wave per pixel
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < 16; k++) {
      const int pp = wave * 16 + k;
      const int qy = 8 * ty + (pp >> 3), qx = 8 * tx + (pp & 7);
      if (qy >= H1 || qx >= W1) continue;  // wave-uniform
      const long qpix = (long)qy * W1 + qx;
      const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + qpix) * 2);
      altcorr_pixel(P.fmap[0] + (fi * HW1 + qpix) * C, f2, H2, W2, C, c.x * scale, c.y * scale, obase + qpix, HW1, lane);
    }
    return;
  }
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; t++) acc[t] = 0.0f;
  const float* __restrict__ f1 = P.fmap[0] + (fi * HW1 + pix) * C;
  const int R4 = RW * RH * (AT_SLAB / 4);
  // LDS offsets of this thread's 16 taps (window rows 2q, 2q+1), -1 when the tap is outside the image; computed once.
  // (A double-buffered variant with the next slab's loads in flight was not faster: the kernel is bound by the LDS
  // reads and the FMAs of the dot products, not by the staging latency.)
  int toff[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int h2 = yb + 2 * q + (t >> 3), w2 = xb + (t & 7);
    const bool ok = h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2;  // inside the image => inside the box
    toff[t] = ok ? ((h2 - y0) * RW + (w2 - x0)) * AT_PITCH : -1;
  }
  for (int c0 = 0; c0 < C; c0 += AT_SLAB) {
    __syncthreads();  // the previous slab has been consumed
    for (int idx = tid; idx < R4; idx += 256) {
      const int r = idx >> 2, c4 = idx & 3;
      const int ry = r / RW, rx = r - ry * RW;
      *reinterpret_cast<float4*>(region + r * AT_PITCH + 4 * c4) =
          *reinterpret_cast<const float4*>(f2 + ((long)(y0 + ry) * W2 + (x0 + rx)) * C + c0 + 4 * c4);
    }
    float4 a[AT_SLAB / 4];
#pragma unroll
    for (int k = 0; k < AT_SLAB / 4; k++) a[k] = sane ? *reinterpret_cast<const float4*>(f1 + c0 + 4 * k) : make_float4(0, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const float* __restrict__ rr = region + max(toff[t], 0);  // branch-free: a masked tap reads entry 0 and adds 0
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < AT_SLAB / 4; k++) s += dot4(a[k], *reinterpret_cast<const float4*>(rr + 4 * k));
      acc[t] += toff[t] >= 0 ? s : 0.0f;
    }
  }
  __syncthreads();  // all slabs consumed: the staging buffer becomes the raw-tap table
#pragma unroll
  for (int t = 0; t < 16; t++) taps[p * AT_TAPP + 16 * q + t] = acc[t];  // raw tap (row 2q + t/8, column t%8)
  __syncthreads();
  if (!inimg) return;
  const float w00 = (1.0f - dy) * (1.0f - dx), w01 = (1.0f - dy) * dx, w10 = dy * (1.0f - dx), w11 = dy * dx;
#pragma unroll
  for (int r2 = 0; r2 < 2; r2++) {
    const int oy = 2 * q + r2;
    if (oy >= 7) continue;
    const float* T0 = taps + p * AT_TAPP + oy * 8;
#pragma unroll
    for (int ox = 0; ox < 7; ox++) {
      // reference weights (altcorr_kernel.cu:112-115), channel = iy + 7*ix (:102-105)
      const float val = T0[ox] * w00 + T0[ox + 1] * w01 + T0[8 + ox] * w10 + T0[8 + ox + 1] * w11;
      obase[(long)(oy + 7 * ox) * HW1 + pix] = val;
    }
  }
}

extern "C" int ns_altcorr_pyramid(const float* const* fmaps_host, int num_levels, const int64_t* ii,
                                  const int64_t* jj, const float* coords, float* out, int E, int H1, int W1, int C,
                                  void* stream) {
  if (E == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(fmaps_host && ii && jj && coords && out, "ns_altcorr_pyramid: null pointer");
  NS_REQUIRE(num_levels >= 1 && num_levels <= 4, "ns_altcorr_pyramid: num_levels=%d not in 1..4", num_levels);
  NS_REQUIRE(E >= 0 && H1 > 0 && W1 > 0 && C > 0 && C % 4 == 0, "ns_altcorr_pyramid: bad shape");
  NS_REQUIRE((H1 >> (num_levels - 1)) > 0 && (W1 >> (num_levels - 1)) > 0, "ns_altcorr_pyramid: level is empty");
  if (E == 0) return NS_OK;
  AltPyramid P;
  P.num_levels = num_levels;
  for (int l = 0; l < 4; l++) {
    P.fmap[l] = fmaps_host[l < num_levels ? l : num_levels - 1];
    NS_REQUIRE(P.fmap[l] != nullptr, "ns_altcorr_pyramid: fmaps[%d] is null", l);
  }
  static const bool per_pixel = ns_variant_env("NS_ALTCORR_PER_PIXEL") != nullptr;  // comparison switch: wave-per-pixel kernel
  if (per_pixel || E > 65535) {
    const long ntask = (long)E * H1 * W1;
    dim3 grid(ns_cdiv(ntask, 4 * ALT_PIX_PER_WAVE), num_levels);
    hipLaunchKernelGGL(altcorr_pyramid_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, out, E, H1,
                       W1, C);
    NS_CHECK_LAUNCH("altcorr_pyramid_kernel");
  } else {
    dim3 grid(((H1 + 7) / 8) * ((W1 + 7) / 8), num_levels, E);
    hipLaunchKernelGGL(altcorr_tile_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, out, E, H1, W1,
                       C);
    NS_CHECK_LAUNCH("altcorr_tile_kernel");
  }
  return NS_OK;
}

// ---------------------------------------------------------------------------------------------
// MFMA variant for HALF-precision feature pyramids (round 3).  The reference's AltCorrBlock holds its pyramid in the dtype of
// the features it is given -- half, in RaftVisualFrontend (visual_frontend.py:209, corr.py:96-105: `/ 4.0` and avg_pool2d
// stay in half; only the call casts to float, :121) -- so every operand of the dot products is an f16 value, products of f16
// values are exact in f32, and v_mfma_f32_32x32x16_f16 computes the same sums as the f32 FMA kernels above up to summation order.
//
// The work IS GEMM-shaped: a workgroup owns an 8 x 8 source-pixel tile of one (edge, level); D = A B^T with A = the tile's 64
// feature vectors (64 x 128) and B = the feature vectors of the union of the tile's windows (R x 128, R ~ 15 x 15 for a smooth
// flow): each source pixel needs 64 of its row's R dot products, the rest is the price of a dense product -- at ~4x the
// flops the matrix cores still finish in a fraction of the time of the vector FMAs (the tile kernel above: 22 TFLOP/s of f32
// FMAs, LDS-read bound).  No operand staging: an A or B fragment is 8 consecutive channels of one pixel = one 16-byte load
// from the channels-last map (L1 / L2 serve the reuse); the A fragments of a wave's 32 pixels stay in registers for all of its
// column tiles.  The accumulators are scattered straight into the 64 x 64 raw-tap table in LDS (a column = region pixel
// belongs to the window of row m iff its offset from the window origin is in [0, 8)^2), the bilinear blend reads that table.
// Tiles whose union exceeds AM_MAXR pixels (wild flow) take the wave-per-pixel routine.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4v_t __attribute__((ext_vector_type(4)));
typedef float f16acc_t __attribute__((ext_vector_type(16)));
#define AM_MAXR 1024
#define AM_C 128

struct AltPyramidH {
  const _Float16* fmap[4];  // level l: [nframes, H>>l, W>>l, 128] channels-last f16 (already / 4, pooled in half)
  int num_levels;
};

// wave-per-pixel fallback on f16 maps (same scheme as altcorr_pixel)
__device__ __forceinline__ void altcorr_pixel_h(const _Float16* __restrict__ f1, const _Float16* __restrict__ f2b, int H2, int W2,
                                                float x2, float y2, float* __restrict__ out, long cs, int lane) {
  const int half = lane >> 5, cl = lane & 31;
  const int oy = cl >> 2, ox = half + 2 * (cl & 3);
  const int src_x1 = (half == 0) ? lane + 32 : lane - 32 + 1;
  const int src_y1 = lane + 4;
  const int src_xy = src_x1 + 4;
  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const float fx0 = floorf(x2), fy0 = floorf(y2);
  const float dx = sane ? x2 - fx0 : 0.0f, dy = sane ? y2 - fy0 : 0.0f;
  const int xb = sane ? (int)fx0 - 3 : -100000;
  const int yb = sane ? (int)fy0 - 3 : -100000;
  float p[32];
#pragma unroll
  for (int v = 0; v < 32; v++) p[v] = 0.0f;
  const h4v_t ah = *reinterpret_cast<const h4v_t*>(f1 + 4 * cl);
  const float4 a = make_float4((float)ah[0], (float)ah[1], (float)ah[2], (float)ah[3]);
#pragma unroll
  for (int iy = 0; iy < 8; iy++) {
    const int h2 = yb + iy;
    const bool rowok = h2 >= 0 && h2 < H2;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int w2 = xb + half + 2 * k;
      if (rowok && w2 >= 0 && w2 < W2) {
        const h4v_t fh = *reinterpret_cast<const h4v_t*>(f2b + ((long)h2 * W2 + w2) * AM_C + 4 * cl);
        p[iy * 4 + k] += dot4(a, make_float4((float)fh[0], (float)fh[1], (float)fh[2], (float)fh[3]));
      }
    }
  }
#pragma unroll
  for (int s = 4; s >= 0; s--) {
    const int cnt = 1 << s;
    const bool up = (cl >> s) & 1;
#pragma unroll
    for (int i = 0; i < cnt; i++) {
      const float lo = p[i], hi = p[i + cnt];
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      p[i] = keep + __shfl_xor(send, cnt, 64);
    }
  }
  const float s00 = p[0];
  const float s01 = __shfl(s00, src_x1, 64);
  const float s10 = __shfl(s00, src_y1, 64);
  const float s11 = __shfl(s00, src_xy, 64);
  if (oy < 7 && ox < 7)
    out[(long)(oy + 7 * ox) * cs] = s00 * ((1.0f - dy) * (1.0f - dx)) + s01 * ((1.0f - dy) * dx) + s10 * (dy * (1.0f - dx)) +
                                    s11 * (dy * dx);
}

#ifdef NS_TEST_VARIANTS   // comparison kernel: libnerfslam_hip_variants.so only (common.h)
__global__ __launch_bounds__(256) void altcorr_tile_mfma_kernel(AltPyramidH P, const int64_t* __restrict__ ii,
                                                                const int64_t* __restrict__ jj,
                                                                const float* __restrict__ coords, float* __restrict__ out,
                                                                int E, int H1, int W1, int xcd_order) {
  __shared__ float taps[64 * AT_TAPP];
  __shared__ int bbox[4], sxb[64], syb[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lvl = blockIdx.y, e = blockIdx.z;
  const int ntx = (W1 + 7) >> 3;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by linear id, and an 8-pixel-wide tile writes 32-byte
  // pieces of the 128-byte lines of its 49 output planes -- horizontally adjacent tiles complete each other's lines only if
  // they meet in the SAME L2.  Tile t of the launch therefore goes to workgroup (t % per) * 8 + t / per, per = tiles / 8:
  // every XCD owns a contiguous run of tiles.
  int tile = blockIdx.x;
  if (xcd_order && (gridDim.x & 7) == 0) {
    const int per = gridDim.x >> 3;
    tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const long HW1 = (long)H1 * W1;
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const float scale = 1.0f / (float)(1 << lvl);
  const long fi = ii[e], fj = jj[e];
  const _Float16* __restrict__ f1 = P.fmap[0] + fi * HW1 * AM_C;
  const _Float16* __restrict__ f2 = P.fmap[lvl] + fj * (long)H2 * W2 * AM_C;
  float* __restrict__ obase = out + ((long)e * P.num_levels * 49 + lvl * 49) * HW1;
  // ---- per source pixel (threads 0..63): window origin, blend weights; bounding box of the windows that touch the image ----
  bool inimg = false;
  long pix = 0;
  if (tid < 4) bbox[tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;
  for (int t = tid; t < 64 * AT_TAPP; t += 256) taps[t] = 0.0f;
  __syncthreads();
  if (tid < 64) {
    const int py = 8 * ty + (tid >> 3), px = 8 * tx + (tid & 7);
    inimg = py < H1 && px < W1;
    pix = inimg ? (long)py * W1 + px : 0;
    float x2 = 0.0f, y2 = 0.0f;
    if (inimg) {
      const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + pix) * 2);
      x2 = c.x * scale;
      y2 = c.y * scale;
    }
    const bool sane = inimg && (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
    const float fx0 = floorf(x2), fy0 = floorf(y2);
    const int xb = sane ? (int)fx0 - 3 : -100000, yb = sane ? (int)fy0 - 3 : -100000;
    sxb[tid] = xb;
    syb[tid] = yb;
    if (sane && xb > -8 && xb < W2 && yb > -8 && yb < H2) {
      atomicMin(&bbox[0], max(xb, 0));
      atomicMin(&bbox[1], max(yb, 0));
      atomicMax(&bbox[2], min(xb + 8, W2));
      atomicMax(&bbox[3], min(yb + 8, H2));
    }
  }
  __syncthreads();
  const bool empty = bbox[0] == 0x7fffffff || bbox[2] == -0x7fffffff;
  const int x0 = bbox[0], y0 = bbox[1], RW = empty ? 0 : bbox[2] - bbox[0], RH = empty ? 0 : bbox[3] - bbox[1];
  if (RW <= 0 || RH <= 0) {  // nothing of this tile looks into the image: zeros
    if (tid < 64 && inimg)
      for (int ch = 0; ch < 49; ch++) obase[(long)ch * HW1 + pix] = 0.0f;
    return;
  }
  const int R = RW * RH;
  if (R > AM_MAXR) {         // workgroup-uniform: wild flow -> wave per pixel
    for (int k = 0; k < 16; k++) {
      const int pp = wave * 16 + k;
      const int qy = 8 * ty + (pp >> 3), qx = 8 * tx + (pp & 7);
      if (qy >= H1 || qx >= W1) continue;  // wave-uniform
      const long qpix = (long)qy * W1 + qx;
      const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + qpix) * 2);
      altcorr_pixel_h(f1 + qpix * AM_C, f2, H2, W2, c.x * scale, c.y * scale, obase + qpix, HW1, lane);
    }
    return;
  }
  // ---- D = A B^T on the matrix cores: wave -> row tile (wave & 1), column tiles (wave >> 1), +2, ... ----
  const int j = lane & 31, kg = lane >> 5;
  const int mt = wave & 1;
  h8_t afrag[AM_C / 16];
  {
    const int m = 32 * mt + j;                                 // source pixel of the tile this lane supplies as A row
    const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
    const bool ok = py < H1 && px < W1;
    const _Float16* __restrict__ src = f1 + ((long)py * W1 + px) * AM_C + 8 * kg;
#pragma unroll
    for (int cc = 0; cc < AM_C / 16; cc++) afrag[cc] = ok ? *reinterpret_cast<const h8_t*>(src + 16 * cc) : (h8_t)(_Float16)0;
  }
  // window origins of the 16 accumulator rows of this lane (rows 4 kg + (q & 3) + 8 (q >> 2) of row tile mt), in registers
  int wxb[16], wyb[16];
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
    wxb[q] = sxb[m];
    wyb[q] = syb[m];
  }
  const int nnt = (R + 31) >> 5;
  // (software-pipelining the fragment loads of the next column tile behind the matrix cores was SLOWER, 1.07 -> 1.45 ms per
  // 48-edge launch: 32 more VGPRs cost more occupancy than the overlap won.  Counters, profiles/r03_altcorr_pmc.json: the
  // matrix cores are ~4 % busy, HBM traffic is 0.97 GB per launch = the output written once + the maps read once; what the
  // kernel waits for is the vector-memory pipeline -- every fragment is a 16-byte piece per lane, 32 different lines per load.)
  for (int nt = wave >> 1; nt < nnt; nt += 2) {
    const int r = 32 * nt + j;                                 // region pixel this lane supplies as B column
    const bool rok = r < R;
    const int ry = rok ? r / RW : 0, rx = rok ? r - ry * RW : 0;
    const _Float16* __restrict__ src = f2 + ((long)(y0 + ry) * W2 + (x0 + rx)) * AM_C + 8 * kg;
    h8_t bfrag[AM_C / 16];
#pragma unroll
    for (int cc = 0; cc < AM_C / 16; cc++) bfrag[cc] = rok ? *reinterpret_cast<const h8_t*>(src + 16 * cc) : (h8_t)(_Float16)0;
    f16acc_t acc = (f16acc_t)0.0f;
#pragma unroll
    for (int cc = 0; cc < AM_C / 16; cc++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[cc], bfrag[cc], acc, 0, 0, 0);
    // this lane holds column j (its own region pixel) of rows 4 kg + (q & 3) + 8 (q >> 2)
    if (rok) {
      const int gx = x0 + rx, gy = y0 + ry;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
        const int tx8 = gx - wxb[q], ty8 = gy - wyb[q];
        if ((unsigned)tx8 < 8u && (unsigned)ty8 < 8u) taps[m * AT_TAPP + ty8 * 8 + tx8] = acc[q];
      }
    }
  }
  __syncthreads();
  // ---- bilinear blend: thread (pixel p, quarter q4) writes output rows 2 q4, 2 q4 + 1 ----
  const int p = tid >> 2, q4 = tid & 3;
  const int py = 8 * ty + (p >> 3), px = 8 * tx + (p & 7);
  if (py >= H1 || px >= W1) return;
  const long opix = (long)py * W1 + px;
  const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + opix) * 2);
  const float x2 = c.x * scale, y2 = c.y * scale;
  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const float ddx = sane ? x2 - floorf(x2) : 0.0f, ddy = sane ? y2 - floorf(y2) : 0.0f;
  const float w00 = (1.0f - ddy) * (1.0f - ddx), w01 = (1.0f - ddy) * ddx, w10 = ddy * (1.0f - ddx), w11 = ddy * ddx;
#pragma unroll
  for (int r2 = 0; r2 < 2; r2++) {
    const int oy = 2 * q4 + r2;
    if (oy >= 7) continue;
    const float* T0 = taps + p * AT_TAPP + oy * 8;
#pragma unroll
    for (int ox = 0; ox < 7; ox++)
      obase[(long)(oy + 7 * ox) * HW1 + opix] = T0[ox] * w00 + T0[ox + 1] * w01 + T0[8 + ox] * w10 + T0[8 + ox + 1] * w11;
  }
}
#endif  // NS_TEST_VARIANTS

// ---------------------------------------------------------------------------------------------
// The same product with its operands STAGED through LDS and its load round trips cut to two (round 4).  What the kernel above
// waits for is neither HBM nor the matrix cores (1.08-1.37 ms per 48-edge launch at 160 x 90, 0.1 of the HBM roof):
//   * a fragment load is 16 bytes per lane from 32 different pixels -- 32 cache lines per instruction for 1 KB -- and the two
//     row-tile waves fetch every column tile twice;
//   * a workgroup is a CHAIN of dependent round trips at 12 waves per CU: source vectors, flow, one per column tile, flow again
//     for the blend, and every __syncthreads() in between drains the vector-memory counter.  Stage-by-stage early exits:
//     prologue alone 0.56 ms, + blend and stores 0.71, + the column tiles 1.10 -- each round trip costs ~3 us under this load.
// Here (1) the workgroup copies whole feature vectors: 16 lanes x 16 bytes = one pixel's 256 bytes, a wave instruction = 4
// pixels = 8 full lines, every byte of the region fetched ONCE per workgroup; the fragments are 16-byte LDS reads at a
// 272-byte pixel pitch (17 sixteen-byte slots: conflict-free).  (2) Source vectors and flow leave together; the first TWO
// 128-pixel chunks of the region (a smooth flow's whole 15 x 15 .. 16 x 16 region) leave together into two register sets; the
// blend weights wait in LDS.  (3) The barriers wait for LDS only (s_waitcnt lgkmcnt(0) + s_barrier), so loads stay in flight
// across them.  Same sums in the same order as the kernel above (k = channels 0..127 in chunks of 16): bit-identical output.
// Measured: 1.05 -> 0.49 ms per 48-edge launch on the global BA's own flow (0.24 of the HBM roof in algorithmic bytes; 4.6 GB
// through the L1s at ~10 TB/s), 1.37 -> 0.71 ms on 3 px of independent noise per pixel (regions of 500-900 pixels).
// ---------------------------------------------------------------------------------------------
#define AS_PITCH 136   // halves per staged pixel (272 bytes)
#define AS_CHUNK 128   // region pixels per chunk
typedef short s16x2_t __attribute__((ext_vector_type(2)));

// (lds_barrier(): common.h)

__global__ __launch_bounds__(256) void altcorr_tile_mfma_lds_kernel(AltPyramidH P, const int64_t* __restrict__ ii,
                                                                    const int64_t* __restrict__ jj,
                                                                    const float* __restrict__ coords, float* __restrict__ out,
                                                                    int E, int H1, int W1, int xcd_order) {
  __shared__ float taps[64 * AT_TAPP];
  __shared__ __attribute__((aligned(16))) _Float16 stage[AS_CHUNK * AS_PITCH];
  __shared__ int bbox[4], sxy[64];
  __shared__ float2 sfr[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lvl = blockIdx.y, e = blockIdx.z;
  const int ntx = (W1 + 7) >> 3;
  int tile = blockIdx.x;
  if (xcd_order && (gridDim.x & 7) == 0) {     // (every XCD owns a contiguous run of tiles: see altcorr_tile_mfma_kernel)
    const int per = gridDim.x >> 3;
    tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const long HW1 = (long)H1 * W1;
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const float scale = 1.0f / (float)(1 << lvl);
  const long fi = ii[e], fj = jj[e];
  const _Float16* __restrict__ f1 = P.fmap[0] + fi * HW1 * AM_C;
  const _Float16* __restrict__ f2 = P.fmap[lvl] + fj * (long)H2 * W2 * AM_C;
  float* __restrict__ obase = out + ((long)e * P.num_levels * 49 + lvl * 49) * HW1;
  // ---- round trip 1: the flow of the tile's pixels and their feature vectors ----
  const int sp = tid >> 4, piece = tid & 15;
  bool inimg = false;
  long pix = 0;
  float2 cf = make_float2(0.0f, 0.0f);
  if (tid < 64) {
    const int py = 8 * ty + (tid >> 3), px = 8 * tx + (tid & 7);
    inimg = py < H1 && px < W1;
    pix = inimg ? (long)py * W1 + px : 0;
    if (inimg) cf = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + pix) * 2);
  }
  h8_t pa[AS_CHUNK / 16], pb[AS_CHUNK / 16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = sp + 16 * i;
    const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
    // (pixels of the tile beyond the image edge: the address is clamped, their rows of the product are never stored)
    pa[i] = *reinterpret_cast<const h8_t*>(f1 + ((long)min(py, H1 - 1) * W1 + min(px, W1 - 1)) * AM_C + 8 * piece);
  }
  for (int t = tid; t < 64 * AT_TAPP; t += 256) taps[t] = 0.0f;
  lds_barrier();
  if (tid < 64) {
    const float x2 = cf.x * scale, y2 = cf.y * scale;
    const bool sane = inimg && (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
    const float fx0 = floorf(x2), fy0 = floorf(y2);
    const int xb = sane ? (int)fx0 - 3 : -100000, yb = sane ? (int)fy0 - 3 : -100000;
    sfr[tid] = make_float2(sane ? x2 - fx0 : 0.0f, sane ? y2 - fy0 : 0.0f);      // the blend's weights
    // window origin as two 16-bit halves (a window that touches the image has its origin in (-8, 32767); anything else is
    // parked at -20000, where no region pixel is within 8 of it)
    const bool touches = sane && xb > -8 && xb < W2 && yb > -8 && yb < H2;
    sxy[tid] = touches ? ((xb & 0xffff) | (yb << 16)) : (int)0xb1e0b1e0u;
    // bounding box of the windows that touch the image: butterfly over the wave (these 64 threads are wave 0)
    int bx0 = touches ? max(xb, 0) : 0x7fffffff, by0 = touches ? max(yb, 0) : 0x7fffffff;
    int bx1 = touches ? min(xb + 8, W2) : -0x7fffffff, by1 = touches ? min(yb + 8, H2) : -0x7fffffff;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      bx0 = min(bx0, __shfl_xor(bx0, d, 64));
      by0 = min(by0, __shfl_xor(by0, d, 64));
      bx1 = max(bx1, __shfl_xor(bx1, d, 64));
      by1 = max(by1, __shfl_xor(by1, d, 64));
    }
    if (tid == 0) {
      bbox[0] = bx0;
      bbox[1] = by0;
      bbox[2] = bx1;
      bbox[3] = by1;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) *reinterpret_cast<h8_t*>(&stage[(sp + 16 * i) * AS_PITCH + 8 * piece]) = pa[i];
  lds_barrier();
  const bool empty = bbox[0] == 0x7fffffff || bbox[2] == -0x7fffffff;
  const int x0 = bbox[0], y0 = bbox[1], RW = empty ? 0 : bbox[2] - bbox[0], RH = empty ? 0 : bbox[3] - bbox[1];
  if (RW <= 0 || RH <= 0) {  // nothing of this tile looks into the image: zeros
    if (tid < 64 && inimg)
      for (int ch = 0; ch < 49; ch++) obase[(long)ch * HW1 + pix] = 0.0f;
    return;
  }
  const int R = RW * RH;
  if (R > AM_MAXR) {         // workgroup-uniform: wild flow -> wave per pixel
    for (int k = 0; k < 16; k++) {
      const int pp = wave * 16 + k;
      const int qy = 8 * ty + (pp >> 3), qx = 8 * tx + (pp & 7);
      if (qy >= H1 || qx >= W1) continue;  // wave-uniform
      const long qpix = (long)qy * W1 + qx;
      const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + qpix) * 2);
      altcorr_pixel_h(f1 + qpix * AM_C, f2, H2, W2, c.x * scale, c.y * scale, obase + qpix, HW1, lane);
    }
    return;
  }
  // region pixel r -> (row, column) of the bounding box.  (r + 0.5) / RW is at least 0.5 / RW away from an integer and the
  // float product is off by < 2e-7 r / RW: the truncation is the exact quotient for every r < 2^20.
  const float inv_rw = 1.0f / (float)RW;
  auto fetch = [&](h8_t (&pre)[AS_CHUNK / 16], int chunk) {
#pragma unroll
    for (int i = 0; i < AS_CHUNK / 16; i++) {
      const int r = min(AS_CHUNK * chunk + sp + 16 * i, R - 1);      // (columns past the region: clamped, never looked at)
      const int ry = (int)(((float)r + 0.5f) * inv_rw), rx = r - ry * RW;
      pre[i] = *reinterpret_cast<const h8_t*>(f2 + ((long)(y0 + ry) * W2 + (x0 + rx)) * AM_C + 8 * piece);
    }
  };
  // ---- round trip 2: the first two chunks of the region ----
  const int nchunk = (R + AS_CHUNK - 1) / AS_CHUNK;
  fetch(pa, 0);
  if (nchunk > 1) fetch(pb, 1);
  const int j = lane & 31, kg = lane >> 5;
  const int mt = wave & 1, ntw = wave >> 1;
  h8_t afrag[AM_C / 16];
#pragma unroll
  for (int cc = 0; cc < AM_C / 16; cc++)
    afrag[cc] = *reinterpret_cast<const h8_t*>(&stage[(32 * mt + j) * AS_PITCH + 16 * cc + 8 * kg]);
  // window origins of the 16 accumulator rows of this lane (rows 4 kg + (q & 3) + 8 (q >> 2) of row tile mt)
  s16x2_t wxy[16];
#pragma unroll
  for (int q = 0; q < 16; q++) wxy[q] = __builtin_bit_cast(s16x2_t, sxy[32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2)]);
  lds_barrier();                                               // (every wave has its source-pixel fragments: the buffer is free)
  auto products = [&](int c) {                                 // chunk c is in the buffer
#pragma unroll
    for (int h = 0; h < AS_CHUNK / 64; h++) {
      const int col = 64 * h + 32 * ntw;                       // column tile of this wave inside the chunk
      if (AS_CHUNK * c + col >= R) continue;                   // (wave-uniform)
      const int r = AS_CHUNK * c + col + j;                    // region pixel this lane holds as column j of its product
      f16acc_t acc = (f16acc_t)0.0f;
      const _Float16* __restrict__ bsrc = &stage[(col + j) * AS_PITCH + 8 * kg];
#pragma unroll
      for (int cc = 0; cc < AM_C / 16; cc++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[cc], *reinterpret_cast<const h8_t*>(bsrc + 16 * cc), acc, 0, 0, 0);
      if (r < R) {
        const int ry = (int)(((float)r + 0.5f) * inv_rw), rx = r - ry * RW;
        s16x2_t gxy;
        gxy[0] = (short)(x0 + rx);
        gxy[1] = (short)(y0 + ry);
#pragma unroll
        for (int q = 0; q < 16; q++) {
          // both offsets from the window origin at once: inside [0, 8)^2 iff no bit above the low three of either half
          const uint32_t d = __builtin_bit_cast(uint32_t, (s16x2_t)(gxy - wxy[q]));
          if ((d & 0xfff8fff8u) == 0u) {
            const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
            taps[m * AT_TAPP + (d >> 13) + (d & 7u)] = acc[q];      // row offset * 8 + column offset
          }
        }
      }
    }
  };
  for (int c = 0; c < nchunk; c += 2) {
#pragma unroll
    for (int i = 0; i < AS_CHUNK / 16; i++) *reinterpret_cast<h8_t*>(&stage[(sp + 16 * i) * AS_PITCH + 8 * piece]) = pa[i];
    lds_barrier();
    if (c + 2 < nchunk) fetch(pa, c + 2);                      // (uniform) in flight while the matrix cores work
    products(c);
    lds_barrier();                                             // (the buffer is free / the taps are complete)
    if (c + 1 >= nchunk) break;
#pragma unroll
    for (int i = 0; i < AS_CHUNK / 16; i++) *reinterpret_cast<h8_t*>(&stage[(sp + 16 * i) * AS_PITCH + 8 * piece]) = pb[i];
    lds_barrier();
    if (c + 3 < nchunk) fetch(pb, c + 3);
    products(c + 1);
    lds_barrier();
  }
  // ---- bilinear blend: thread (pixel p, quarter q4) writes output rows 2 q4, 2 q4 + 1 ----
  const int p = tid >> 2, q4 = tid & 3;
  const int py = 8 * ty + (p >> 3), px = 8 * tx + (p & 7);
  if (py >= H1 || px >= W1) return;
  const long opix = (long)py * W1 + px;
  const float2 fr = sfr[p];
  const float ddx = fr.x, ddy = fr.y;
  const float w00 = (1.0f - ddy) * (1.0f - ddx), w01 = (1.0f - ddy) * ddx, w10 = ddy * (1.0f - ddx), w11 = ddy * ddx;
#pragma unroll
  for (int r2 = 0; r2 < 2; r2++) {
    const int oy = 2 * q4 + r2;
    if (oy >= 7) continue;
    const float* T0 = taps + p * AT_TAPP + oy * 8;
#pragma unroll
    for (int ox = 0; ox < 7; ox++)
      obase[(long)(oy + 7 * ox) * HW1 + opix] = T0[ox] * w00 + T0[ox + 1] * w01 + T0[8 + ox] * w10 + T0[8 + ox + 1] * w11;
  }
}

// ---------------------------------------------------------------------------------------------
// On-the-fly correlation + correlation encoder in ONE launch (round 4; the config-#5 counterpart of corr_lookup_enc_kernel).
// Reference chain: AltCorrBlock.__call__ (networks/modules/corr.py:107-131, four levels, `.float()` results concatenated) ->
// UpdateModule.corr_encoder[0:2] = Conv2d(196,128,1) + ReLU (networks/droid_net.py:83-87,133, under autocast: half inputs).
// The plain kernel writes 196 f32 planes per edge -- 96 % of its bytes at 160x90 -- which a transposition and a 1x1 convolution
// then read twice.  Here a workgroup owns an 8x8 tile through ALL FOUR levels: per level the same bounding-region product on the
// matrix cores and the same bilinear blend as altcorr_tile_mfma_kernel, the 49 results rounded to half (what autocast hands the
// convolution) into an LDS tile [208 channels][64 pixels]; then X W^T on MFMA (13 k-chunks, each wave two 32x32 output tiles),
// bias, ReLU, [E,H,W,128] f16 channels-last.  The source-pixel fragments (level independent) are loaded once per tile.
// ---------------------------------------------------------------------------------------------
#define AE_K 208
#define AE_PITCH 72
#ifdef NS_TEST_VARIANTS   // comparison kernel: libnerfslam_hip_variants.so only (common.h)
__global__ __launch_bounds__(256) void altcorr_tile_enc_kernel(AltPyramidH P, const int64_t* __restrict__ ii,
                                                               const int64_t* __restrict__ jj, const float* __restrict__ coords,
                                                               const h8_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                               _Float16* __restrict__ out, int E, int H1, int W1) {
  __shared__ float taps[64 * AT_TAPP];
  __shared__ __attribute__((aligned(16))) uint16_t X[AE_K * AE_PITCH];       // 30 KB: [channel][pixel of the tile]
  __shared__ int bbox[4], sxb[64], syb[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = blockIdx.y, tile = blockIdx.x;
  const int ntx = (W1 + 7) >> 3;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const long HW1 = (long)H1 * W1;
  const long fi = ii[e], fj = jj[e];
  const _Float16* __restrict__ f1 = P.fmap[0] + fi * HW1 * AM_C;
  const int j = lane & 31, kg = lane >> 5;
  const int mt = wave & 1;
  for (int t = tid; t < (AE_K - 196) * AE_PITCH; t += 256) X[196 * AE_PITCH + t] = 0;     // pad channels stay zero
  h8_t afrag[AM_C / 16];
  {
    const int m = 32 * mt + j;                                 // source pixel of the tile this lane supplies as A row
    const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
    const bool ok = py < H1 && px < W1;
    const _Float16* __restrict__ src = f1 + ((long)py * W1 + px) * AM_C + 8 * kg;
#pragma unroll
    for (int cc = 0; cc < AM_C / 16; cc++) afrag[cc] = ok ? *reinterpret_cast<const h8_t*>(src + 16 * cc) : (h8_t)(_Float16)0;
  }
#pragma unroll 1
  for (int lvl = 0; lvl < 4; lvl++) {
    const int H2 = H1 >> lvl, W2 = W1 >> lvl;
    const float scale = 1.0f / (float)(1 << lvl);
    const _Float16* __restrict__ f2 = P.fmap[lvl] + fj * (long)H2 * W2 * AM_C;
    uint16_t* __restrict__ Xl = X + lvl * 49 * AE_PITCH;
    __syncthreads();                                           // the previous level's taps / bbox have been consumed
    if (tid < 4) bbox[tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;
    for (int t = tid; t < 64 * AT_TAPP; t += 256) taps[t] = 0.0f;
    __syncthreads();
    if (tid < 64) {
      const int py = 8 * ty + (tid >> 3), px = 8 * tx + (tid & 7);
      const bool inimg = py < H1 && px < W1;
      float x2 = 0.0f, y2 = 0.0f;
      if (inimg) {
        const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + (long)py * W1 + px) * 2);
        x2 = c.x * scale;
        y2 = c.y * scale;
      }
      const bool sane = inimg && (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
      const float fx0 = floorf(x2), fy0 = floorf(y2);
      const int xb = sane ? (int)fx0 - 3 : -100000, yb = sane ? (int)fy0 - 3 : -100000;
      sxb[tid] = xb;
      syb[tid] = yb;
      if (sane && xb > -8 && xb < W2 && yb > -8 && yb < H2) {
        atomicMin(&bbox[0], max(xb, 0));
        atomicMin(&bbox[1], max(yb, 0));
        atomicMax(&bbox[2], min(xb + 8, W2));
        atomicMax(&bbox[3], min(yb + 8, H2));
      }
    }
    __syncthreads();
    const bool empty = bbox[0] == 0x7fffffff || bbox[2] == -0x7fffffff;
    const int x0 = bbox[0], y0 = bbox[1], RW = empty ? 0 : bbox[2] - bbox[0], RH = empty ? 0 : bbox[3] - bbox[1];
    const int R = RW * RH;
    const bool wild = R > AM_MAXR;                             // (workgroup-uniform) wild flow -> wave per pixel, 49 outputs straight
    if (RW > 0 && RH > 0 && wild) {                            //  into the tap table (row p, 49 entries: cs = 1)
      for (int k = 0; k < 16; k++) {
        const int pp = wave * 16 + k;
        const int qy = 8 * ty + (pp >> 3), qx = 8 * tx + (pp & 7);
        if (qy >= H1 || qx >= W1) continue;  // wave-uniform
        const long qpix = (long)qy * W1 + qx;
        const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + qpix) * 2);
        altcorr_pixel_h(f1 + qpix * AM_C, f2, H2, W2, c.x * scale, c.y * scale, taps + pp * AT_TAPP, 1, lane);
      }
    } else if (RW > 0 && RH > 0) {
      int wxb[16], wyb[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
        wxb[q] = sxb[m];
        wyb[q] = syb[m];
      }
      const int nnt = (R + 31) >> 5;
      for (int nt = wave >> 1; nt < nnt; nt += 2) {
        const int r = 32 * nt + j;                             // region pixel this lane supplies as B column
        const bool rok = r < R;
        const int ry = rok ? r / RW : 0, rx = rok ? r - ry * RW : 0;
        const _Float16* __restrict__ src = f2 + ((long)(y0 + ry) * W2 + (x0 + rx)) * AM_C + 8 * kg;
        h8_t bfrag[AM_C / 16];
#pragma unroll
        for (int cc = 0; cc < AM_C / 16; cc++) bfrag[cc] = rok ? *reinterpret_cast<const h8_t*>(src + 16 * cc) : (h8_t)(_Float16)0;
        f16acc_t acc = (f16acc_t)0.0f;
#pragma unroll
        for (int cc = 0; cc < AM_C / 16; cc++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[cc], bfrag[cc], acc, 0, 0, 0);
        if (rok) {
          const int gx = x0 + rx, gy = y0 + ry;
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
            const int tx8 = gx - wxb[q], ty8 = gy - wyb[q];
            if ((unsigned)tx8 < 8u && (unsigned)ty8 < 8u) taps[m * AT_TAPP + ty8 * 8 + tx8] = acc[q];
          }
        }
      }
    }
    __syncthreads();
    // ---- the level's 49 values of every pixel, rounded to half, into the encoder's input tile ----
    {
      const int p = tid >> 2, q4 = tid & 3;
      const int py = 8 * ty + (p >> 3), px = 8 * tx + (p & 7);
      const bool inimg = py < H1 && px < W1;
      if (RW <= 0 || RH <= 0 || !inimg) {                      // nothing looks into the image (or padding pixel): zeros
        for (int ch = q4; ch < 49; ch += 4) Xl[ch * AE_PITCH + p] = 0;
      } else if (wild) {
        for (int ch = q4; ch < 49; ch += 4) Xl[ch * AE_PITCH + p] = __builtin_bit_cast(uint16_t, (_Float16)taps[p * AT_TAPP + ch]);
      } else {
        const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + (long)py * W1 + px) * 2);
        const float x2 = c.x * scale, y2 = c.y * scale;
        const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
        const float ddx = sane ? x2 - floorf(x2) : 0.0f, ddy = sane ? y2 - floorf(y2) : 0.0f;
        const float w00 = (1.0f - ddy) * (1.0f - ddx), w01 = (1.0f - ddy) * ddx, w10 = ddy * (1.0f - ddx), w11 = ddy * ddx;
#pragma unroll
        for (int r2 = 0; r2 < 2; r2++) {
          const int oy = 2 * q4 + r2;
          if (oy >= 7) continue;
          const float* T0 = taps + p * AT_TAPP + oy * 8;
#pragma unroll
          for (int ox = 0; ox < 7; ox++) {
            const float v = T0[ox] * w00 + T0[ox + 1] * w01 + T0[8 + ox] * w10 + T0[8 + ox + 1] * w11;
            Xl[(oy + 7 * ox) * AE_PITCH + p] = __builtin_bit_cast(uint16_t, (_Float16)v);
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- out[64 px][128] = relu(X W^T + b): wave (mt, nt0 = wave >> 1) computes the tiles (mt, nt0) and (mt, nt0 + 2) ----
  const uint16_t* xa = X + 8 * kg * AE_PITCH + 32 * mt + j;    // A: row = pixel 32 mt + j, k = channel 16 c + 8 kg + q
  h8_t a[13];
#pragma unroll
  for (int c = 0; c < 13; c++)
#pragma unroll
    for (int q = 0; q < 8; q++) a[c][q] = __builtin_bit_cast(_Float16, xa[(16 * c + q) * AE_PITCH]);
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int nt = (wave >> 1) + 2 * u;
    f16acc_t acc = (f16acc_t)0.0f;
#pragma unroll
    for (int c = 0; c < 13; c++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], wfrag[(nt * 13 + c) * 64 + lane], acc, 0, 0, 0);
    const float bj = bias[32 * nt + j];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * mt + 4 * kg + (r & 3) + 8 * (r >> 2);
      const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
      if (py < H1 && px < W1) out[((long)e * HW1 + (long)py * W1 + px) * 128 + 32 * nt + j] = (_Float16)fmaxf(acc[r] + bj, 0.0f);
    }
  }
}
#endif  // NS_TEST_VARIANTS

// ---------------------------------------------------------------------------------------------
// The fused kernel with the LDS-staged operands of altcorr_tile_mfma_lds_kernel (round 4).  A workgroup owns an 8x8 tile
// through all four levels and sees its work as ONE stream of 64-pixel chunks -- level 0's region, then level 1's, ... -- with
// the loads of the next THREE chunks always in flight (three register sets; at two workgroups per CU a wave has 256 registers
// to itself): after the first round trip (flow + source vectors; wave l prepares level l: window origins, blend weights,
// bounding box by butterfly) the matrix cores never wait for a whole round trip again.  The tap table is not zero-filled:
// the blend masks taps outside the image by their coordinates (same values: a zero tap contributes a zero term).  Encoder
// input tile as [pixel][216 halves] so that its fragments are 16-byte LDS reads.  65 KB of LDS per workgroup.
// Measured (48 edges, 160 x 90, a rigid scene's flow): 1.36 ms -> 0.63-0.65 ms per launch, the global BA's correlation leg
// 112 -> 51 ms per pass.  Stage by stage: prologue alone 0.04 ms; the stream without products 0.45 (3.9 GB through the L1s at
// ~10 TB/s: three chunks in flight cover a third of the ~2.5 us a load takes at that rate -- the L2 -> L1 fabric, not HBM, is
// the roof of this kernel, and only larger tiles would move fewer bytes); products + 0.2, blends + 0.11, encoder + 0.1.
// ---------------------------------------------------------------------------------------------
#define AE_XP 216      // halves per pixel of the encoder's input tile (208 channels + 8: 27 sixteen-byte slots)
#define AE_CHUNK 64
#define AE_DEPTH 3
__global__ __launch_bounds__(256, 2) void altcorr_tile_enc_lds_kernel(AltPyramidH P, const int64_t* __restrict__ ii,
                                                                      const int64_t* __restrict__ jj,
                                                                      const float* __restrict__ coords,
                                                                      const h8_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                                      _Float16* __restrict__ out, int E, int H1, int W1,
                                                                      int xcd_order) {
  __shared__ float taps[64 * AT_TAPP];
  __shared__ __attribute__((aligned(16))) uint16_t X[64 * AE_XP];                  // 27 KB: [pixel of the tile][channel]
  __shared__ __attribute__((aligned(16))) _Float16 stage[AE_CHUNK * AS_PITCH];     // 17 KB
  __shared__ int sxy[4][64], lvp[4][4];
  __shared__ float2 sfr[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = blockIdx.y;
  int tile = blockIdx.x;
  if (xcd_order && (gridDim.x & 7) == 0) {     // (every XCD owns a contiguous run of tiles: neighbours share their regions in L2)
    const int per = gridDim.x >> 3;
    tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const int ntx = (W1 + 7) >> 3;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const long HW1 = (long)H1 * W1;
  const long fi = ii[e], fj = jj[e];
  const _Float16* __restrict__ f1 = P.fmap[0] + fi * HW1 * AM_C;
  const int j = lane & 31, kg = lane >> 5;
  const int mt = wave & 1, ntw = wave >> 1;
  const int sp = tid >> 4, piece = tid & 15;
  // ---- round trip 1: flow of the tile's pixels (every wave: lane = pixel) and their feature vectors ----
  const int ppy = 8 * ty + (lane >> 3), ppx = 8 * tx + (lane & 7);
  const bool inimg = ppy < H1 && ppx < W1;
  const float2 cf = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + (long)min(ppy, H1 - 1) * W1 + min(ppx, W1 - 1)) * 2);
  h8_t pre[AE_DEPTH][AE_CHUNK / 16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = sp + 16 * i;
    const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
    pre[0][i] = *reinterpret_cast<const h8_t*>(f1 + ((long)min(py, H1 - 1) * W1 + min(px, W1 - 1)) * AM_C + 8 * piece);
  }
  for (int t = tid; t < 64 * (AE_XP - 196); t += 256) X[(t / (AE_XP - 196)) * AE_XP + 196 + t % (AE_XP - 196)] = 0;   // pad channels
  {
    // wave l prepares level l
    const int l = wave;
    const int H2 = H1 >> l, W2 = W1 >> l;
    const float scale = 1.0f / (float)(1 << l);
    const float x2 = cf.x * scale, y2 = cf.y * scale;
    const bool sane = inimg && (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
    const float fx0 = floorf(x2), fy0 = floorf(y2);
    const int xb = sane ? (int)fx0 - 3 : -100000, yb = sane ? (int)fy0 - 3 : -100000;
    sfr[l][lane] = make_float2(sane ? x2 - fx0 : 0.0f, sane ? y2 - fy0 : 0.0f);
    // window origin as two 16-bit halves; a window that does not touch the image is parked at -20000 (no tap of it is inside)
    const bool touches = sane && xb > -8 && xb < W2 && yb > -8 && yb < H2;
    sxy[l][lane] = touches ? ((xb & 0xffff) | (yb << 16)) : (int)0xb1e0b1e0u;
    int bx0 = touches ? max(xb, 0) : 0x7fffffff, by0 = touches ? max(yb, 0) : 0x7fffffff;
    int bx1 = touches ? min(xb + 8, W2) : -0x7fffffff, by1 = touches ? min(yb + 8, H2) : -0x7fffffff;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      bx0 = min(bx0, __shfl_xor(bx0, d, 64));
      by0 = min(by0, __shfl_xor(by0, d, 64));
      bx1 = max(bx1, __shfl_xor(bx1, d, 64));
      by1 = max(by1, __shfl_xor(by1, d, 64));
    }
    if (lane == 0) {
      const bool none = bx0 == 0x7fffffff || bx1 <= bx0 || by1 <= by0;
      lvp[l][0] = none ? 0 : bx0;
      lvp[l][1] = none ? 0 : by0;
      lvp[l][2] = none ? 0 : bx1 - bx0;                        // RW
      lvp[l][3] = none ? 0 : (bx1 - bx0) * (by1 - by0);       // R (0: nothing of this tile looks into the image)
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) *reinterpret_cast<h8_t*>(&stage[(sp + 16 * i) * AS_PITCH + 8 * piece]) = pre[0][i];
  lds_barrier();
  h8_t afrag[AM_C / 16];
#pragma unroll
  for (int cc = 0; cc < AM_C / 16; cc++)
    afrag[cc] = *reinterpret_cast<const h8_t*>(&stage[(32 * mt + j) * AS_PITCH + 16 * cc + 8 * kg]);
  // chunks per level: 0 for an empty level and for a wild one (R > AM_MAXR: wave per pixel, below)
  int nch[4], nelem = 0;
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const int R = lvp[l][3];
    nch[l] = (R > 0 && R <= AM_MAXR) ? (R + AE_CHUNK - 1) / AE_CHUNK : 0;
    nelem += nch[l];
  }
  auto locate = [&](int k, int& l, int& c) {                   // element k of the stream -> (level, chunk)
    l = 0;
    c = k;
    if (c >= nch[0]) { c -= nch[0]; l = 1;
      if (c >= nch[1]) { c -= nch[1]; l = 2;
        if (c >= nch[2]) { c -= nch[2]; l = 3; } } }
  };
  auto fetch = [&](h8_t (&pr)[AE_CHUNK / 16], int k) {
    int l, c;
    locate(k, l, c);
    const int H2 = H1 >> l, W2 = W1 >> l;
    const _Float16* __restrict__ fm = l == 0 ? P.fmap[0] : l == 1 ? P.fmap[1] : l == 2 ? P.fmap[2] : P.fmap[3];
    const _Float16* __restrict__ f2 = fm + fj * (long)H2 * W2 * AM_C;
    const int x0 = lvp[l][0], y0 = lvp[l][1], RW = lvp[l][2], R = lvp[l][3];
    const float inv_rw = 1.0f / (float)RW;
#pragma unroll
    for (int i = 0; i < AE_CHUNK / 16; i++) {
      const int r = min(AE_CHUNK * c + sp + 16 * i, R - 1);          // (columns past the region: clamped, never looked at)
      const int ry = (int)(((float)r + 0.5f) * inv_rw), rx = r - ry * RW;
      pr[i] = *reinterpret_cast<const h8_t*>(f2 + ((long)(y0 + ry) * W2 + (x0 + rx)) * AM_C + 8 * piece);
    }
  };
  lds_barrier();                                               // (every wave has its source-pixel fragments: the buffer is free)
  if (nelem > 0) fetch(pre[0], 0);
  if (nelem > 1) fetch(pre[1], 1);
  if (nelem > 2) fetch(pre[2], 2);
  // ---- the blend of one level: thread (pixel p, quarter q4) -> rows 2 q4, 2 q4 + 1 of the 7 x 7 outputs, rounded to half ----
  auto blend = [&](int l, bool wild) {
    const int p = tid >> 2, q4 = tid & 3;
    const int py = 8 * ty + (p >> 3), px = 8 * tx + (p & 7);
    uint16_t* __restrict__ Xp = X + p * AE_XP + 49 * l;
    if (py >= H1 || px >= W1) {                                // padding pixel of the tile
      for (int ch = q4; ch < 49; ch += 4) Xp[ch] = 0;
    } else if (wild) {                                         // the wave-per-pixel routine left the 49 finished values
      for (int ch = q4; ch < 49; ch += 4) Xp[ch] = __builtin_bit_cast(uint16_t, (_Float16)taps[p * AT_TAPP + ch]);
    } else {
      const int H2 = H1 >> l, W2 = W1 >> l;
      const float2 fr = sfr[l][p];
      const int w = sxy[l][p];
      const int xb = (int)(short)(w & 0xffff), yb = w >> 16;
      const float ddx = fr.x, ddy = fr.y;
      const float w00 = (1.0f - ddy) * (1.0f - ddx), w01 = (1.0f - ddy) * ddx, w10 = ddy * (1.0f - ddx), w11 = ddy * ddx;
      float T[3][8];                                           // window rows 2 q4 .. 2 q4 + 2, taps outside the image as zeros
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int wy = 2 * q4 + r;
        const bool rowok = wy < 8 && (unsigned)(yb + wy) < (unsigned)H2;
#pragma unroll
        for (int x = 0; x < 8; x++) {
          const float v = taps[p * AT_TAPP + min(wy, 7) * 8 + x];
          T[r][x] = (rowok && (unsigned)(xb + x) < (unsigned)W2) ? v : 0.0f;
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; r2++) {
        const int oy = 2 * q4 + r2;
        if (oy >= 7) continue;
#pragma unroll
        for (int ox = 0; ox < 7; ox++) {
          const float v = T[r2][ox] * w00 + T[r2][ox + 1] * w01 + T[r2 + 1][ox] * w10 + T[r2 + 1][ox + 1] * w11;
          Xp[oy + 7 * ox] = __builtin_bit_cast(uint16_t, (_Float16)v);
        }
      }
    }
  };
  // ---- levels without chunks: empty (the blend's masks give zeros) or wild (wave per pixel into the tap table) ----
#pragma unroll 1
  for (int l = 0; l < 4; l++) {
    if (nch[l] > 0) continue;                                  // (uniform)
    const bool wild = lvp[l][3] > AM_MAXR;
    if (wild) {
      const int H2 = H1 >> l, W2 = W1 >> l;
      const float scale = 1.0f / (float)(1 << l);
      const _Float16* __restrict__ fm = l == 0 ? P.fmap[0] : l == 1 ? P.fmap[1] : l == 2 ? P.fmap[2] : P.fmap[3];
      const _Float16* __restrict__ f2 = fm + fj * (long)H2 * W2 * AM_C;
      for (int k = 0; k < 16; k++) {
        const int pp = wave * 16 + k;
        const int qy = 8 * ty + (pp >> 3), qx = 8 * tx + (pp & 7);
        if (qy >= H1 || qx >= W1) continue;  // wave-uniform
        const long qpix = (long)qy * W1 + qx;
        const float2 c = *reinterpret_cast<const float2*>(coords + ((long)e * HW1 + qpix) * 2);
        altcorr_pixel_h(f1 + qpix * AM_C, f2, H2, W2, c.x * scale, c.y * scale, taps + pp * AT_TAPP, 1, lane);
      }
      __syncthreads();
    }
    blend(l, wild);
    __syncthreads();
  }
  // ---- the stream of chunks ----
  s16x2_t wxy[16];
  auto step = [&](h8_t (&pr)[AE_CHUNK / 16], int k) {
    int l, c;
    locate(k, l, c);
#pragma unroll
    for (int i = 0; i < AE_CHUNK / 16; i++) *reinterpret_cast<h8_t*>(&stage[(sp + 16 * i) * AS_PITCH + 8 * piece]) = pr[i];
    if (c == 0) {                                              // (uniform) a new level: window origins of this lane's 16 rows
#pragma unroll
      for (int q = 0; q < 16; q++) wxy[q] = __builtin_bit_cast(s16x2_t, sxy[l][32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2)]);
    }
    lds_barrier();
    if (k + AE_DEPTH < nelem) fetch(pr, k + AE_DEPTH);         // (uniform) in flight for the next three steps
    const int x0 = lvp[l][0], y0 = lvp[l][1], RW = lvp[l][2], R = lvp[l][3];
    const int col = 32 * ntw;                                  // column tile of this wave inside the chunk
    if (AE_CHUNK * c + col < R) {                              // (wave-uniform)
      const int r = AE_CHUNK * c + col + j;                    // region pixel this lane holds as column j of its product
      f16acc_t acc = (f16acc_t)0.0f;
      const _Float16* __restrict__ bsrc = &stage[(col + j) * AS_PITCH + 8 * kg];
#pragma unroll
      for (int cc = 0; cc < AM_C / 16; cc++)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[cc], *reinterpret_cast<const h8_t*>(bsrc + 16 * cc), acc, 0, 0, 0);
      if (r < R) {
        const float inv_rw = 1.0f / (float)RW;
        const int ry = (int)(((float)r + 0.5f) * inv_rw), rx = r - ry * RW;
        s16x2_t gxy;
        gxy[0] = (short)(x0 + rx);
        gxy[1] = (short)(y0 + ry);
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const uint32_t d = __builtin_bit_cast(uint32_t, (s16x2_t)(gxy - wxy[q]));
          if ((d & 0xfff8fff8u) == 0u) {                       // both offsets from the window origin inside [0, 8)
            const int m = 32 * mt + 4 * kg + (q & 3) + 8 * (q >> 2);
            taps[m * AT_TAPP + (d >> 13) + (d & 7u)] = acc[q];
          }
        }
      }
    }
    lds_barrier();                                             // (the buffer is free / this chunk's taps are in the table)
    if (c == nch[l] - 1) {                                     // (uniform) the level is complete
      blend(l, false);
      lds_barrier();
    }
  };
  for (int k = 0; k < nelem; k += AE_DEPTH) {
    step(pre[0], k);
    if (k + 1 < nelem) step(pre[1], k + 1);
    if (k + 2 < nelem) step(pre[2], k + 2);
  }
  lds_barrier();
  // ---- out[64 px][128] = relu(X W^T + b): wave (mt, nt0 = wave >> 1) computes the tiles (mt, nt0) and (mt, nt0 + 2) ----
  const uint16_t* xa = X + (32 * mt + j) * AE_XP + 8 * kg;     // A: row = pixel 32 mt + j, k = channel 16 c + 8 kg + q
  h8_t a[13];
#pragma unroll
  for (int c = 0; c < 13; c++) a[c] = *reinterpret_cast<const h8_t*>(xa + 16 * c);
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int nt = ntw + 2 * u;
    f16acc_t acc = (f16acc_t)0.0f;
#pragma unroll
    for (int c = 0; c < 13; c++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], wfrag[(nt * 13 + c) * 64 + lane], acc, 0, 0, 0);
    const float bj = bias[32 * nt + j];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * mt + 4 * kg + (r & 3) + 8 * (r >> 2);
      const int py = 8 * ty + (m >> 3), px = 8 * tx + (m & 7);
      if (py < H1 && px < W1) out[((long)e * HW1 + (long)py * W1 + px) * 128 + 32 * nt + j] = (_Float16)fmaxf(acc[r] + bj, 0.0f);
    }
  }
}

extern "C" int ns_altcorr_pyramid_encode_f16(const void* const* fmaps_host, const int64_t* ii, const int64_t* jj, const float* coords,
                                             const void* wfrag, const float* bias, void* out, int E, int H1, int W1, int C,
                                             void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(fmaps_host && ii && jj && coords && wfrag && bias && out, "ns_altcorr_pyramid_encode_f16: null pointer");
  NS_REQUIRE(E >= 0 && H1 > 0 && W1 > 0 && (H1 >> 3) > 0 && (W1 >> 3) > 0, "ns_altcorr_pyramid_encode_f16: bad shape (four levels)");
  NS_REQUIRE(((uintptr_t)wfrag % 16) == 0, "ns_altcorr_pyramid_encode_f16: the weight fragments must be 16-byte aligned");
  if (C != AM_C || E > 65535) {
    ns_set_error("ns_altcorr_pyramid_encode_f16: built for 128 channels and at most 65535 edges per call (C=%d, E=%d)", C, E);
    return NS_ENOSUP;
  }
  AltPyramidH P;
  P.num_levels = 4;
  for (int l = 0; l < 4; l++) {
    P.fmap[l] = (const _Float16*)fmaps_host[l];
    NS_REQUIRE(P.fmap[l] != nullptr && ((uintptr_t)P.fmap[l] % 16) == 0, "ns_altcorr_pyramid_encode_f16: fmaps[%d] null or not 16-byte aligned", l);
  }
  dim3 grid(((H1 + 7) / 8) * ((W1 + 7) / 8), E);
  static const bool direct = ns_variant_env("NS_ALTCORR_DIRECT") != nullptr;   // A/B switch: fragments straight from global memory
  static const bool no_xcd = ns_variant_env("NS_ALTCORR_NO_XCD") != nullptr;   // A/B switch: linear tile order
#ifdef NS_TEST_VARIANTS
  if (direct) {
    hipLaunchKernelGGL(altcorr_tile_enc_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, (const h8_t*)wfrag, bias,
                       (_Float16*)out, E, H1, W1);
    NS_CHECK_LAUNCH("altcorr_tile_enc_kernel");
    return NS_OK;
  }
#endif
  (void)direct;
  hipLaunchKernelGGL(altcorr_tile_enc_lds_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, (const h8_t*)wfrag,
                     bias, (_Float16*)out, E, H1, W1, no_xcd ? 0 : 1);
  NS_CHECK_LAUNCH("altcorr_tile_enc_lds_kernel");
  return NS_OK;
}

extern "C" int ns_altcorr_pyramid_f16(const void* const* fmaps_host, int num_levels, const int64_t* ii, const int64_t* jj,
                                      const float* coords, float* out, int E, int H1, int W1, int C, void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(fmaps_host && ii && jj && coords && out, "ns_altcorr_pyramid_f16: null pointer");
  NS_REQUIRE(num_levels >= 1 && num_levels <= 4, "ns_altcorr_pyramid_f16: num_levels=%d not in 1..4", num_levels);
  NS_REQUIRE(E >= 0 && H1 > 0 && W1 > 0, "ns_altcorr_pyramid_f16: bad shape");
  NS_REQUIRE((H1 >> (num_levels - 1)) > 0 && (W1 >> (num_levels - 1)) > 0, "ns_altcorr_pyramid_f16: level is empty");
  if (C != AM_C || E > 65535) {
    ns_set_error("ns_altcorr_pyramid_f16: built for 128 channels and at most 65535 edges per call (C=%d, E=%d): use the f32 entry", C, E);
    return NS_ENOSUP;
  }
  AltPyramidH P;
  P.num_levels = num_levels;
  for (int l = 0; l < 4; l++) {
    P.fmap[l] = (const _Float16*)fmaps_host[l < num_levels ? l : num_levels - 1];
    NS_REQUIRE(P.fmap[l] != nullptr && ((uintptr_t)P.fmap[l] % 16) == 0, "ns_altcorr_pyramid_f16: fmaps[%d] null or not 16-byte aligned", l);
  }
  dim3 grid(((H1 + 7) / 8) * ((W1 + 7) / 8), num_levels, E);
  static const bool no_xcd = ns_variant_env("NS_ALTCORR_NO_XCD") != nullptr;   // A/B switch: linear tile order
  static const bool direct = ns_variant_env("NS_ALTCORR_DIRECT") != nullptr;   // A/B switch: fragments straight from global memory
#ifdef NS_TEST_VARIANTS
  if (direct) {
    hipLaunchKernelGGL(altcorr_tile_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, out, E, H1, W1,
                       no_xcd ? 0 : 1);
    NS_CHECK_LAUNCH("altcorr_tile_mfma_kernel");
    return NS_OK;
  }
#endif
  (void)direct;
  hipLaunchKernelGGL(altcorr_tile_mfma_lds_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, ii, jj, coords, out, E, H1, W1,
                     no_xcd ? 0 : 1);
  NS_CHECK_LAUNCH("altcorr_tile_mfma_lds_kernel");
  return NS_OK;
}

extern "C" int ns_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B,
                                  int H1, int W1, int H2, int W2, int C, int N, int radius, void* stream) {
  NS_REQUIRE(fmap1 && fmap2 && coords && corr, "ns_altcorr_forward: null pointer");
  NS_REQUIRE(B >= 0 && N >= 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0, "ns_altcorr_forward: bad shape");
  if (radius != 3) {
    ns_set_error("ns_altcorr_forward: only radius 3 is built (the reference never uses another, corr.py:93)");
    return NS_ENOSUP;
  }
  NS_REQUIRE(C % 4 == 0, "ns_altcorr_forward: C=%d must be a multiple of 4", C);
  const long ntask = (long)B * N * H1 * W1;
  if (ntask == 0) return NS_OK;
  const int blocks = ns_cdiv(ntask, 4 * ALT_PIX_PER_WAVE);
  hipLaunchKernelGGL(altcorr_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, coords,
                     corr, B, H1, W1, H2, W2, C, N);
  NS_CHECK_LAUNCH("altcorr_forward_kernel");
  return NS_OK;
}

// ---------------------------------------------------------------------------------------------
// altcorr_backward (src/droid.cpp:315-327 -> altcorr_kernel.cu:150-288): gradients of the on-the-fly correlation with
// respect to both feature maps.  Dead in the reference's live path (inference only, examples/slam_demo.py:198) -- kept so
// that every op of the module answers.  One wave per source pixel (b, h1, w1), lanes over channels: per coordinate set the
// 49 output gradients are folded back onto the 8x8 raw taps with the transposed bilinear weights (:232-248),
//   fmap1_grad[b,h1,w1,:]  += G(tap) * fmap2[b, tap, :]      (one writer: plain store at the end)
//   fmap2_grad[b, tap, :]  += G(tap) * fmap1[b,h1,w1,:]      (float atomics, like the reference's :266-267)
// ---------------------------------------------------------------------------------------------
#define ALTB_MAXC 8   // channels per lane: C <= 512

__global__ __launch_bounds__(256) void altcorr_backward_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                               const float* __restrict__ coords,
                                                               const float* __restrict__ corr_grad,
                                                               float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad,
                                                               int B, int H1, int W1, int H2, int W2, int C, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long HW1 = (long)H1 * W1;
  const long task = (long)blockIdx.x * 4 + wave;     // (b, pixel)
  if (task >= (long)B * HW1) return;                 // wave-uniform
  const int b = (int)(task / HW1);
  const long pix = task - (long)b * HW1;
  const float* __restrict__ f1 = fmap1 + task * C;
  float a[ALTB_MAXC], ga[ALTB_MAXC];
#pragma unroll
  for (int k = 0; k < ALTB_MAXC; k++) {
    const int c = lane + 64 * k;
    a[k] = c < C ? f1[c] : 0.0f;
    ga[k] = 0.0f;
  }
  for (int n = 0; n < N; n++) {
    const float2 xy = *reinterpret_cast<const float2*>(coords + (((long)b * N + n) * HW1 + pix) * 2);
    if (!(fabsf(xy.x) < 1.0e6f) || !(fabsf(xy.y) < 1.0e6f)) continue;   // the forward pass writes zeros there
    const float fx0 = floorf(xy.x), fy0 = floorf(xy.y);
    const float dx = xy.x - fx0, dy = xy.y - fy0;
    const int xb = (int)fx0 - 3, yb = (int)fy0 - 3;
    // channel iy + 7 ix of the output gradient on lane iy + 7 ix
    const float gl = lane < 49 ? corr_grad[(((long)b * N + n) * 49 + lane) * HW1 + pix] : 0.0f;
    for (int iy = 0; iy < 8; iy++) {
      for (int ix = 0; ix < 8; ix++) {
        float G = 0.0f;   // (altcorr_kernel.cu:237-248: nw, ne, sw, se of the raw tap)
        const float g_nw = __shfl(gl, max((iy - 1) + 7 * (ix - 1), 0), 64), g_ne = __shfl(gl, max((iy - 1) + 7 * min(ix, 6), 0), 64);
        const float g_sw = __shfl(gl, max(min(iy, 6) + 7 * (ix - 1), 0), 64), g_se = __shfl(gl, min(iy, 6) + 7 * min(ix, 6), 64);
        if (iy > 0 && ix > 0) G += g_nw * dy * dx;
        if (iy > 0 && ix < 7) G += g_ne * dy * (1.0f - dx);
        if (iy < 7 && ix > 0) G += g_sw * (1.0f - dy) * dx;
        if (iy < 7 && ix < 7) G += g_se * (1.0f - dy) * (1.0f - dx);
        const int h2 = yb + iy, w2 = xb + ix;
        if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2 || G == 0.0f) continue;   // wave-uniform
        const long o2 = (((long)b * H2 + h2) * W2 + w2) * C;
#pragma unroll
        for (int k = 0; k < ALTB_MAXC; k++) {
          const int c = lane + 64 * k;
          if (c < C) {
            ga[k] = fmaf(G, fmap2[o2 + c], ga[k]);
            atomicAdd(&fmap2_grad[o2 + c], G * a[k]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ALTB_MAXC; k++) {
    const int c = lane + 64 * k;
    if (c < C) fmap1_grad[task * C + c] = ga[k];
  }
}

extern "C" int ns_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                                   float* fmap1_grad, float* fmap2_grad, int B, int H1, int W1, int H2, int W2, int C, int N,
                                   int radius, void* stream) {
  NS_REQUIRE(fmap1 && fmap2 && coords && corr_grad && fmap1_grad && fmap2_grad, "ns_altcorr_backward: null pointer");
  NS_REQUIRE(B >= 0 && N >= 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0 && C <= 64 * ALTB_MAXC,
             "ns_altcorr_backward: bad shape (C = %d, at most %d channels)", C, 64 * ALTB_MAXC);
  if (radius != 3) {
    ns_set_error("ns_altcorr_backward: only radius 3 is built (the reference never uses another, corr.py:93)");
    return NS_ENOSUP;
  }
  const long tasks = (long)B * H1 * W1;
  if (tasks == 0) return NS_OK;
  hipLaunchKernelGGL(altcorr_backward_kernel, dim3(ns_cdiv(tasks, 4)), dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, coords,
                     corr_grad, fmap1_grad, fmap2_grad, B, H1, W1, H2, W2, C, N);
  NS_CHECK_LAUNCH("altcorr_backward_kernel");
  return NS_OK;
}
