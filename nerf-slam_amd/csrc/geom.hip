// geom.hip -- frame distance, projection maps, back-projection, depth filter (gfx950).
//
//   K3 frame_distance_kernel   src/droid_kernels.cu:630-769   (live: keyframe selection, proximity edges)
//   K2 projmap_kernel          :539-628                        (dead in the live path, kept for API parity)
//   K5 iproj_kernel            :896-967                        (dead)
//   K4 depth_filter_kernel     :773-892                        (dead)
#include "common.h"
#include "se3.h"

// One workgroup per frame pair.  The three sums are reduced in a FIXED order (lane-strided
// partials, xor-butterfly inside each wave, waves 0..3 added in order) so the result is
// bit-reproducible run to run: add_proximity_factors argsorts these distances
// (visual_frontend.py:750) and the factor-graph indices must not depend on scheduling.
__global__ __launch_bounds__(256) void frame_distance_kernel(const float* __restrict__ poses,
                                                             const float* __restrict__ disps,
                                                             const float* __restrict__ intr,
                                                             const int64_t* __restrict__ ii,
                                                             const int64_t* __restrict__ jj, float* __restrict__ dist,
                                                             int HW, int wd, float beta) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int ix = (int)ii[b], jx = (int)jj[b];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float tij[3], qij[4];
  se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij,
               qij);
  float accum = 0.0f, valid = 0.0f, total = 0.0f;
  const float omb = 1.0f - beta;
  const float* __restrict__ disp = disps + (long)ix * HW;
  for (int k = tid; k < HW; k += 256) {
    const int i = k / wd, j = k - i * wd;
    const float u = (float)j, v = (float)i;
    float Xi[4], Xj[4];
    Xi[0] = (u - cx) / fx;
    Xi[1] = (v - cy) / fy;
    Xi[2] = 1.0f;
    Xi[3] = disp[k];
    se3::act_se3(tij, qij, Xi, Xj);
    float du = fx * (Xj[0] / Xj[2]) + cx - u;
    float dv = fy * (Xj[1] / Xj[2]) + cy - v;
    float d = sqrtf(du * du + dv * dv);
    total += beta;
    if (Xj[2] > NS_MIN_DEPTH) {
      accum += beta * d;
      valid += beta;
    }
    // translation-only flow (:730-748)
    const float X0 = Xi[0] + Xi[3] * tij[0];
    const float X1 = Xi[1] + Xi[3] * tij[1];
    const float X2 = Xi[2] + Xi[3] * tij[2];
    du = fx * (X0 / X2) + cx - u;
    dv = fy * (X1 / X2) + cy - v;
    d = sqrtf(du * du + dv * dv);
    total += omb;
    if (X2 > NS_MIN_DEPTH) {
      accum += omb * d;
      valid += omb;
    }
  }
  __shared__ float red[3][4];
  const int wave = tid >> 6, lane = tid & 63;
  accum = wave_sum(accum);
  valid = wave_sum(valid);
  total = wave_sum(total);
  if (lane == 0) {
    red[0][wave] = accum;
    red[1][wave] = valid;
    red[2][wave] = total;
  }
  __syncthreads();
  if (tid == 0) {
    const float A = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const float V = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const float Tt = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    dist[b] = ((double)V / ((double)Tt + 1e-8) < 0.75) ? 1000.0f : A / V;  // (:767)
  }
}

__global__ __launch_bounds__(256) void projmap_kernel(const float* __restrict__ poses,
                                                      const float* __restrict__ disps,
                                                      const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                                      const int64_t* __restrict__ jj, float* __restrict__ coords,
                                                      float* __restrict__ valid, int HW, int wd) {
  const int b = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k >= HW) return;
  const int ix = (int)ii[b], jx = (int)jj[b];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float tij[3], qij[4];
  se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij,
               qij);
  const int i = k / wd, j = k - i * wd;
  const float u = (float)j, v = (float)i;
  float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, disps[(long)ix * HW + k]}, Xj[4];
  se3::act_se3(tij, qij, Xi, Xj);
  float c0 = u, c1 = v;
  if (Xj[2] > 0.01f) {
    c0 = fx * (Xj[0] / Xj[2]) + cx;
    c1 = fy * (Xj[1] / Xj[2]) + cy;
  }
  float* c = coords + ((long)b * HW + k) * 3;
  c[0] = c0;
  c[1] = c1;
  valid[(long)b * HW + k] = (Xj[2] > NS_MIN_DEPTH) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void iproj_kernel(const float* __restrict__ poses, const float* __restrict__ disps,
                                                    const float* __restrict__ intr, float* __restrict__ points,
                                                    int HW, int wd) {
  const int b = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k >= HW) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int i = k / wd, j = k - i * wd;
  float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.0f, disps[(long)b * HW + k]}, Xj[4];
  se3::act_se3(poses + (long)b * 7, poses + (long)b * 7 + 3, Xi, Xj);
  float* p = points + ((long)b * HW + k) * 3;
  p[0] = Xj[0] / Xj[3];
  p[1] = Xj[1] / Xj[3];
  p[2] = Xj[2] / Xj[3];
}

// grid (num, 6 neighbours, pixel chunks); every (block_id, pixel) is touched by at most six
// workgroups (one per neighbour), so the count uses float atomics exactly like the reference.
__global__ __launch_bounds__(256) void depth_filter_kernel(const float* __restrict__ poses,
                                                           const float* __restrict__ disps,
                                                           const float* __restrict__ intr,
                                                           const int64_t* __restrict__ inds,
                                                           const float* __restrict__ thresh,
                                                           float* __restrict__ counter, int nframes, int ht, int wd) {
  const int b = blockIdx.x;
  const int nb = blockIdx.y;
  const int HW = ht * wd;
  const int k = blockIdx.z * 256 + threadIdx.x;
  const int ix = (int)inds[b];
  const int jx = (nb < 3) ? ix - nb - 1 : ix + nb;
  if (jx < 0 || jx >= nframes || k >= HW) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float t = thresh[b];
  float tij[3], qij[4];
  se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij,
               qij);
  const int i = k / wd, j = k - i * wd;
  float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.0f, disps[(long)ix * HW + k]}, Xj[4];
  se3::act_se3(tij, qij, Xi, Xj);
  const float uj = fx * (Xj[0] / Xj[2]) + cx;
  const float vj = fy * (Xj[1] / Xj[2]) + cy;
  const float dj = Xj[3] / Xj[2];
  const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
  if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
    const float* dm = disps + (long)jx * HW;
    const double idj = 1.0 / (double)dj;
    const double d00 = dm[(v0 + 0) * wd + u0 + 0], d01 = dm[(v0 + 0) * wd + u0 + 1];
    const double d10 = dm[(v0 + 1) * wd + u0 + 0], d11 = dm[(v0 + 1) * wd + u0 + 1];
    if (fabs(idj - 1.0 / d00) < t || fabs(idj - 1.0 / d01) < t || fabs(idj - 1.0 / d10) < t ||
        fabs(idj - 1.0 / d11) < t)
      atomicAdd(&counter[(long)b * HW + k], 1.0f);
  }
}

// Reprojection ii -> jj as the frontend's update() needs it every iteration
// (RaftVisualFrontend.reproject, visual_frontend.py:909-918 -> networks/geom/projective_ops.py:98-145 without
// the Jacobians, which update() computes and throws away): one launch instead of ~15 torch /
// lietorch launches.  Conventions of the torch path: depth below 0.5*MIN_DEPTH is replaced by 1 before the
// division (projective_ops.py:45), MIN_DEPTH = 0.2 for the validity mask (:8, :117), stereo pairs (ii == jj)
// use the fixed baseline (:100,110).  coords [E,ht,wd,2] (the layout the lookup kernel reads), valid [E,ht,wd].
__global__ __launch_bounds__(256) void reproject_kernel(const float* __restrict__ poses,
                                                        const float* __restrict__ disps,
                                                        const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                                        const int64_t* __restrict__ jj, float* __restrict__ coords,
                                                        float* __restrict__ valid, int HW, int wd) {
  const int e = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k >= HW) return;
  const int ix = (int)ii[e], jx = (int)jj[e];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float tij[3], qij[4];
  if (ix == jx) {
    tij[0] = -0.1f;
    tij[1] = tij[2] = 0.0f;
    qij[0] = qij[1] = qij[2] = 0.0f;
    qij[3] = 1.0f;
  } else {
    se3::rel_se3(poses + (long)ix * 7, poses + (long)ix * 7 + 3, poses + (long)jx * 7, poses + (long)jx * 7 + 3, tij,
                 qij);
  }
  const int i = k / wd, j = k - i * wd;
  float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.0f, disps[(long)ix * HW + k]}, Xj[4];
  se3::act_se3(tij, qij, Xi, Xj);
  const float Z = (Xj[2] < 0.1f) ? 1.0f : Xj[2];
  const float d = 1.0f / Z;
  float2 c;
  c.x = fx * (Xj[0] * d) + cx;
  c.y = fy * (Xj[1] * d) + cy;
  *reinterpret_cast<float2*>(coords + ((long)e * HW + k) * 2) = c;
  if (valid) valid[(long)e * HW + k] = (Xj[2] > 0.2f) ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Convex 8x upsampling of a per-keyframe map with the update operator's mask (utils/flow_viz.py:166-183, used at
// visual_frontend.py:445-446,513-514): out[8y+sy, 8x+sx] = sum_k softmax_k(mask[k, sy, sx, y, x])^pow * data[y+dy_k, x+dx_k]
// over the 3x3 neighbourhood, neighbours outside the image excluded from the softmax (the reference sets their logits
// to -inf).  One lane per (coarse pixel, output row): every mask plane is read with consecutive lanes on consecutive x, and
// the lane's 8 outputs leave as two 16-byte pieces of a contiguous row (a wave writes 2 KB contiguous).
// ---------------------------------------------------------------------------------------------
// With kx != nullptr the maps are gathered from / scattered to whole keyframe buffers (frame kx[f] of data_* [buffer,ht,wd]
// and out_* [buffer,8ht,8wd]); data_b / out_b (optional) is a second map upsampled with the SAME weights in the same pass
// (the frontend upsamples inverse depth and depth covariance with one mask, visual_frontend.py:445-446), so the 576-channel
// mask -- 98 % of the bytes -- is read and soft-maxed once instead of twice.
template <typename MT, bool PAIR>
__global__ __launch_bounds__(256) void cvx_upsample_kernel(const float* __restrict__ data_a, const float* __restrict__ data_b,
                                                           const int64_t* __restrict__ kx, const MT* __restrict__ mask,
                                                           float* __restrict__ out_a, float* __restrict__ out_b, int n,
                                                           int ht, int wd, float pw) {
  // one lane per (coarse pixel, output row sy): 8x the lanes of a lane-per-pixel mapping.  With ~10 keyframes of 4800
  // pixels a lane per pixel is < 1 wave per SIMD, each walking 576 dependent strided loads (540 us measured).
  // Lanes run over the LINEAR pixel index (rows of 80 pixels would leave a 64-wide x tiling 62 % full and every
  // 160-byte mask row straddling two cache lines).
  const long HW = (long)ht * wd;
  const int p = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int f = blockIdx.z;
  if (p >= HW) return;
  const int y = p / wd, x = p - y * wd;
  const long fs = kx ? (long)kx[f] : (long)f;
  const float* da = data_a + fs * HW;
  const float* db = PAIR ? data_b + fs * HW : nullptr;
  const MT* m = mask + (long)f * 576 * HW + (long)y * wd + x;
  float na[9], nb[9];
  bool ok[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    ok[k] = yy >= 0 && yy < ht && xx >= 0 && xx < wd;
    na[k] = ok[k] ? da[(long)yy * wd + xx] : 0.0f;
    nb[k] = (PAIR && ok[k]) ? db[(long)yy * wd + xx] : 0.0f;
  }
  float lg[8][9];
#pragma unroll
  for (int sx = 0; sx < 8; sx++)
#pragma unroll
    for (int k = 0; k < 9; k++) lg[sx][k] = ok[k] ? (float)m[(long)(k * 64 + sy * 8 + sx) * HW] : -INFINITY;
  float ra[8], rb[8];
#pragma unroll
  for (int sx = 0; sx < 8; sx++) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; k++) mx = fmaxf(mx, lg[sx][k]);
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      lg[sx][k] = ok[k] ? __expf(lg[sx][k] - mx) : 0.0f;
      den += lg[sx][k];
    }
    const float inv = 1.0f / den;
    float acca = 0.0f, accb = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      float w = lg[sx][k] * inv;
      if (pw != 1.0f) w = __powf(w, pw);
      acca = fmaf(w, na[k], acca);
      if (PAIR) accb = fmaf(w, nb[k], accb);
    }
    ra[sx] = acca;
    rb[sx] = accb;
  }
  const long o = fs * HW * 64 + ((long)(8 * y + sy) * (8 * wd) + 8 * x);
  float4* dst = reinterpret_cast<float4*>(out_a + o);
  dst[0] = make_float4(ra[0], ra[1], ra[2], ra[3]);
  dst[1] = make_float4(ra[4], ra[5], ra[6], ra[7]);
  if (PAIR) {
    float4* dsb = reinterpret_cast<float4*>(out_b + o);
    dsb[0] = make_float4(rb[0], rb[1], rb[2], rb[3]);
    dsb[1] = make_float4(rb[4], rb[5], rb[6], rb[7]);
  }
}

// f16 masks, even wd: one lane per (PAIR of x-adjacent coarse pixels, output row sy, half of the 8 sub-columns).  The logits
// of the two pixels are one aligned dword, so a wave instruction fetches 256 contiguous bytes of a mask plane instead of
// 128: measured 22 us (L2/MALL-warm mask) / 51 us (cold) for 10 keyframes of 80x60 against 58 / 91 us with one pixel per
// lane (fetching the pixel's own dword and selecting the half in registers did not help: it is the bytes per wave
// instruction, not the 2-byte load type).
template <bool PAIR>
__global__ __launch_bounds__(256) void cvx_upsample_h2_kernel(const float* __restrict__ data_a, const float* __restrict__ data_b,
                                                              const int64_t* __restrict__ kx,
                                                              const _Float16* __restrict__ mask, float* __restrict__ out_a,
                                                              float* __restrict__ out_b, int n, int ht, int wd, float pw) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  const long HW = (long)ht * wd;
  const int pp = blockIdx.x * 64 + (threadIdx.x & 63);  // pixel pair: pixels 2pp, 2pp+1 (same row: wd is even)
  const int sub = blockIdx.y * 4 + (threadIdx.x >> 6);   // 0..15
  const int sy = sub >> 1, sx0 = (sub & 1) * 4;
  const int f = blockIdx.z;
  if (2L * pp >= HW) return;
  const int y = (2 * pp) / wd, x = 2 * pp - y * wd;
  const long fs = kx ? (long)kx[f] : (long)f;
  const float* da = data_a + fs * HW;
  const float* db = PAIR ? data_b + fs * HW : nullptr;
  // 3 x 4 window of the data around the pair
  float wa[3][4], wb[3][4];
  bool okr[3], okc[4];
#pragma unroll
  for (int r = 0; r < 3; r++) okr[r] = (y + r - 1) >= 0 && (y + r - 1) < ht;
#pragma unroll
  for (int c = 0; c < 4; c++) okc[c] = (x + c - 1) >= 0 && (x + c - 1) < wd;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const bool o = okr[r] && okc[c];
      wa[r][c] = o ? da[(long)(y + r - 1) * wd + (x + c - 1)] : 0.0f;
      wb[r][c] = (PAIR && o) ? db[(long)(y + r - 1) * wd + (x + c - 1)] : 0.0f;
    }
  const half2_t* m2 = reinterpret_cast<const half2_t*>(mask + (long)f * 576 * HW) + pp;
  half2_t lg[4][9];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int k = 0; k < 9; k++) lg[s][k] = m2[((long)(k * 64 + sy * 8 + sx0 + s) * HW) >> 1];
  float ra[2][4], rb[2][4];
#pragma unroll
  for (int q = 0; q < 2; q++)   // pixel of the pair
#pragma unroll
    for (int s = 0; s < 4; s++) {
      float e[9], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const bool o = okr[k / 3] && okc[k % 3 + q];
        e[k] = o ? (float)lg[s][k][q] : -INFINITY;
        mx = fmaxf(mx, e[k]);
      }
      float den = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const bool o = okr[k / 3] && okc[k % 3 + q];
        e[k] = o ? __expf(e[k] - mx) : 0.0f;
        den += e[k];
      }
      const float inv = 1.0f / den;
      float acca = 0.0f, accb = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        float w = e[k] * inv;
        if (pw != 1.0f) w = __powf(w, pw);
        acca = fmaf(w, wa[k / 3][k % 3 + q], acca);
        if (PAIR) accb = fmaf(w, wb[k / 3][k % 3 + q], accb);
      }
      ra[q][s] = acca;
      rb[q][s] = accb;
    }
  const long o = fs * HW * 64 + ((long)(8 * y + sy) * (8 * wd) + 8 * x + sx0);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    *reinterpret_cast<float4*>(out_a + o + 8 * q) = make_float4(ra[q][0], ra[q][1], ra[q][2], ra[q][3]);
    if (PAIR) *reinterpret_cast<float4*>(out_b + o + 8 * q) = make_float4(rb[q][0], rb[q][1], rb[q][2], rb[q][3]);
  }
}

// Channels-last mask [n, ht, wd, 576] f16 (what the update operator's 1x1 convolution writes): one lane per (pixel, output
// row sy), the 8 lanes of a pixel side by side -- for each neighbour k they fetch the 8 logits of row sy as one 16-byte
// piece, 128 contiguous bytes per pixel and k -- and a wave's 8 pixels are x-adjacent, so every output row segment it
// stores is 256 contiguous bytes.
template <bool PAIR>
__global__ __launch_bounds__(256) void cvx_upsample_cl_kernel(const float* __restrict__ data_a, const float* __restrict__ data_b,
                                                              const int64_t* __restrict__ kx,
                                                              const _Float16* __restrict__ mask, float* __restrict__ out_a,
                                                              float* __restrict__ out_b, int n, int ht, int wd, float pw) {
  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  const long HW = (long)ht * wd;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  const long p = t >> 3;
  const int sy = (int)(t & 7);
  if (p >= HW) return;
  const int y = (int)(p / wd), x = (int)(p - (long)y * wd);
  const long fs = kx ? (long)kx[f] : (long)f;
  const float* da = data_a + fs * HW;
  const float* db = PAIR ? data_b + fs * HW : nullptr;
  const _Float16* m = mask + ((long)f * HW + p) * 576 + sy * 8;
  float na[9], nb[9];
  bool ok[9];
  half8_t lg[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    ok[k] = yy >= 0 && yy < ht && xx >= 0 && xx < wd;
    na[k] = ok[k] ? da[(long)yy * wd + xx] : 0.0f;
    nb[k] = (PAIR && ok[k]) ? db[(long)yy * wd + xx] : 0.0f;
    lg[k] = *reinterpret_cast<const half8_t*>(m + k * 64);
  }
  float ra[8], rb[8];
#pragma unroll
  for (int sx = 0; sx < 8; sx++) {
    float e[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      e[k] = ok[k] ? (float)lg[k][sx] : -INFINITY;
      mx = fmaxf(mx, e[k]);
    }
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      e[k] = ok[k] ? __expf(e[k] - mx) : 0.0f;
      den += e[k];
    }
    const float inv = 1.0f / den;
    float acca = 0.0f, accb = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      float w = e[k] * inv;
      if (pw != 1.0f) w = __powf(w, pw);
      acca = fmaf(w, na[k], acca);
      if (PAIR) accb = fmaf(w, nb[k], accb);
    }
    ra[sx] = acca;
    rb[sx] = accb;
  }
  const long o = fs * HW * 64 + ((long)(8 * y + sy) * (8 * wd) + 8 * x);
  float4* dst = reinterpret_cast<float4*>(out_a + o);
  dst[0] = make_float4(ra[0], ra[1], ra[2], ra[3]);
  dst[1] = make_float4(ra[4], ra[5], ra[6], ra[7]);
  if (PAIR) {
    float4* dsb = reinterpret_cast<float4*>(out_b + o);
    dsb[0] = make_float4(rb[0], rb[1], rb[2], rb[3]);
    dsb[1] = make_float4(rb[4], rb[5], rb[6], rb[7]);
  }
}

extern "C" int ns_cvx_upsample_keyframes_nhwc(const float* data_a, const float* data_b, const int64_t* kx, const void* mask,
                                              float* out_a, float* out_b, int n, int ht, int wd, float pow_, void* stream) {
  if (n == 0) return NS_OK;
  NS_REQUIRE(data_a && mask && out_a && kx, "ns_cvx_upsample_keyframes_nhwc: null pointer");
  NS_REQUIRE((data_b == nullptr) == (out_b == nullptr), "ns_cvx_upsample_keyframes_nhwc: data_b and out_b go together");
  NS_REQUIRE(n > 0 && n <= 65535 && ht > 0 && wd > 0, "ns_cvx_upsample_keyframes_nhwc: bad shape");
  NS_REQUIRE(((uintptr_t)mask % 16) == 0, "ns_cvx_upsample_keyframes_nhwc: mask must be 16-byte aligned");
  dim3 grid(ns_cdiv((long)ht * wd * 8, 256), n);
  if (data_b)
    hipLaunchKernelGGL((cvx_upsample_cl_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                       (const _Float16*)mask, out_a, out_b, n, ht, wd, pow_);
  else
    hipLaunchKernelGGL((cvx_upsample_cl_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                       (const _Float16*)mask, out_a, out_b, n, ht, wd, pow_);
  NS_CHECK_LAUNCH("cvx_upsample_cl_kernel");
  return NS_OK;
}

template <typename MT>
static void cvx_launch(const float* data_a, const float* data_b, const int64_t* kx, const void* mask, float* out_a, float* out_b,
                       int n, int ht, int wd, float pow_, void* stream) {
  if (sizeof(MT) == 2 && wd % 2 == 0 && ns_variant_env("NS_CVX_ONE_PIXEL") == nullptr) {
    dim3 g2(ns_cdiv((long)ht * wd / 2, 64), 4, n);
    if (data_b)
      hipLaunchKernelGGL((cvx_upsample_h2_kernel<true>), g2, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                         (const _Float16*)mask, out_a, out_b, n, ht, wd, pow_);
    else
      hipLaunchKernelGGL((cvx_upsample_h2_kernel<false>), g2, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                         (const _Float16*)mask, out_a, out_b, n, ht, wd, pow_);
    return;
  }
  dim3 grid(ns_cdiv((long)ht * wd, 64), 2, n);
  if (data_b)
    hipLaunchKernelGGL((cvx_upsample_kernel<MT, true>), grid, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                       (const MT*)mask, out_a, out_b, n, ht, wd, pow_);
  else
    hipLaunchKernelGGL((cvx_upsample_kernel<MT, false>), grid, dim3(256), 0, (hipStream_t)stream, data_a, data_b, kx,
                       (const MT*)mask, out_a, out_b, n, ht, wd, pow_);
}

extern "C" int ns_cvx_upsample(const float* data, const void* mask, int mask_dtype, float* out, int n, int ht, int wd,
                               float pow_, void* stream) {
  if (n == 0) return NS_OK;
  NS_REQUIRE(data && mask && out, "ns_cvx_upsample: null pointer");
  NS_REQUIRE(n > 0 && ht > 0 && wd > 0, "ns_cvx_upsample: bad shape");
  NS_REQUIRE(mask_dtype == NS_F16 || mask_dtype == NS_F32, "ns_cvx_upsample: mask dtype %d unsupported", mask_dtype);
  if (mask_dtype == NS_F16)
    cvx_launch<_Float16>(data, nullptr, nullptr, mask, out, nullptr, n, ht, wd, pow_, stream);
  else
    cvx_launch<float>(data, nullptr, nullptr, mask, out, nullptr, n, ht, wd, pow_, stream);
  NS_CHECK_LAUNCH("cvx_upsample_kernel");
  return NS_OK;
}

extern "C" int ns_cvx_upsample_keyframes(const float* data_a, const float* data_b, const int64_t* kx, const void* mask,
                                         int mask_dtype, float* out_a, float* out_b, int n, int ht, int wd, float pow_,
                                         void* stream) {
  if (n == 0) return NS_OK;
  NS_REQUIRE(data_a && mask && out_a && kx, "ns_cvx_upsample_keyframes: null pointer");
  NS_REQUIRE((data_b == nullptr) == (out_b == nullptr), "ns_cvx_upsample_keyframes: data_b and out_b go together");
  NS_REQUIRE(n > 0 && ht > 0 && wd > 0, "ns_cvx_upsample_keyframes: bad shape");
  NS_REQUIRE(mask_dtype == NS_F16 || mask_dtype == NS_F32, "ns_cvx_upsample_keyframes: mask dtype %d unsupported", mask_dtype);
  if (mask_dtype == NS_F16)
    cvx_launch<_Float16>(data_a, data_b, kx, mask, out_a, out_b, n, ht, wd, pow_, stream);
  else
    cvx_launch<float>(data_a, data_b, kx, mask, out_a, out_b, n, ht, wd, pow_, stream);
  NS_CHECK_LAUNCH("cvx_upsample_kernel");
  return NS_OK;
}

// motion features of the update operator (visual_frontend.py:379-386): cat(coords1 - coords0, target - coords1) clamped to
// +-64, channel-first [E,4,ht,wd], from the frontend's interleaved [E,ht,wd,2] tensors -- one launch instead of
// sub, sub, cat, permute, clamp, contiguous.
__global__ __launch_bounds__(256) void motion_features_kernel(const float* __restrict__ coords1,
                                                              const float* __restrict__ target, float* __restrict__ out,
                                                              int E, int HW, int wd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)E * HW) return;
  const int e = (int)(i / HW), p = (int)(i - (long)e * HW);
  const float2 c = *reinterpret_cast<const float2*>(coords1 + 2 * i);
  const float2 t = *reinterpret_cast<const float2*>(target + 2 * i);
  const float gx = (float)(p % wd), gy = (float)(p / wd);
  float* o = out + (long)e * 4 * HW + p;
  o[0] = fminf(fmaxf(c.x - gx, -64.0f), 64.0f);
  o[HW] = fminf(fmaxf(c.y - gy, -64.0f), 64.0f);
  o[2L * HW] = fminf(fmaxf(t.x - c.x, -64.0f), 64.0f);
  o[3L * HW] = fminf(fmaxf(t.y - c.y, -64.0f), 64.0f);
}

extern "C" int ns_motion_features(const float* coords1, const float* target, float* out, int E, int ht, int wd,
                                  void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(coords1 && target && out, "ns_motion_features: null pointer");
  NS_REQUIRE(E > 0 && ht > 0 && wd > 0, "ns_motion_features: bad shape");
  const long n = (long)E * ht * wd;
  hipLaunchKernelGGL(motion_features_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coords1, target, out,
                     E, ht * wd, wd);
  NS_CHECK_LAUNCH("motion_features_kernel");
  return NS_OK;
}

extern "C" int ns_reproject(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                            const int64_t* jj, float* coords, float* valid, int num, int ht, int wd, void* stream) {
  if (num <= 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(poses && disps && intrinsics && coords, "ns_reproject: null pointer");
  NS_REQUIRE(ii && jj, "ns_reproject: null index");
  hipLaunchKernelGGL(reproject_kernel, dim3(num, ns_cdiv(ht * wd, 256)), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, ii, jj, coords, valid, ht * wd, wd);
  NS_CHECK_LAUNCH("reproject_kernel");
  return NS_OK;
}

extern "C" int ns_frame_distance(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                                 const int64_t* jj, float* dist, int num, int ht, int wd, float beta, void* stream) {
  if (num == 0) return NS_OK;  // an empty set is a no-op whatever the pointers are
  NS_REQUIRE(poses && disps && intrinsics && dist, "ns_frame_distance: null pointer");
  NS_REQUIRE(num >= 0 && ht > 0 && wd > 0, "ns_frame_distance: bad shape");
  if (num == 0) return NS_OK;
  NS_REQUIRE(ii && jj, "ns_frame_distance: null index");
  hipLaunchKernelGGL(frame_distance_kernel, dim3(num), dim3(256), 0, (hipStream_t)stream, poses, disps, intrinsics, ii,
                     jj, dist, ht * wd, wd, beta);
  NS_CHECK_LAUNCH("frame_distance_kernel");
  return NS_OK;
}

extern "C" int ns_projmap(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                          const int64_t* jj, float* coords, float* valid, int num, int ht, int wd, void* stream) {
  NS_REQUIRE(poses && disps && intrinsics && coords && valid, "ns_projmap: null pointer");
  if (num <= 0) return NS_OK;
  NS_REQUIRE(ii && jj, "ns_projmap: null index");
  hipLaunchKernelGGL(projmap_kernel, dim3(num, ns_cdiv(ht * wd, 256)), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, ii, jj, coords, valid, ht * wd, wd);
  NS_CHECK_LAUNCH("projmap_kernel");
  return NS_OK;
}

extern "C" int ns_iproj(const float* poses, const float* disps, const float* intrinsics, float* points, int nm, int ht,
                        int wd, void* stream) {
  NS_REQUIRE(poses && disps && intrinsics && points, "ns_iproj: null pointer");
  if (nm <= 0) return NS_OK;
  hipLaunchKernelGGL(iproj_kernel, dim3(nm, ns_cdiv(ht * wd, 256)), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, points, ht * wd, wd);
  NS_CHECK_LAUNCH("iproj_kernel");
  return NS_OK;
}

extern "C" int ns_depth_filter(const float* poses, const float* disps, const float* intrinsics, const int64_t* inds,
                               const float* thresh, float* counter, int num, int nframes, int ht, int wd,
                               void* stream) {
  NS_REQUIRE(poses && disps && intrinsics && counter, "ns_depth_filter: null pointer");
  if (num <= 0) return NS_OK;
  NS_REQUIRE(inds && thresh, "ns_depth_filter: null index");
  hipLaunchKernelGGL(depth_filter_kernel, dim3(num, 6, ns_cdiv(ht * wd, 256)), dim3(256), 0, (hipStream_t)stream, poses,
                     disps, intrinsics, inds, thresh, counter, nframes, ht, wd);
  NS_CHECK_LAUNCH("depth_filter_kernel");
  return NS_OK;
}
