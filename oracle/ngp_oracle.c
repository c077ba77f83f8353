/*
 * ngp_oracle.c -- CPU ORACLE of the mapping hot path (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED.  The arithmetic of NeRF-SLAM's mapping backend lives in an un-vendored fork,
 * ToniRV/instant-ngp @ feature/nerf_slam (commit unknown; /root/reference/.gitmodules:7-10), reached
 * only through ~40 `pyngp` call sites (/root/reference/fusion/nerf_fusion.py:57-101, 285-303,
 * 388-424).  There are no sources, tests or golden vectors for it in /root/reference, so this file
 * restates the PUBLISHED algorithm (Mueller et al. 2022, "Instant Neural Graphics Primitives";
 * tiny-cuda-nn's multiresolution hash encoding and fully-fused MLP; instant-ngp's occupancy-grid
 * ray marcher and NeRF loss) with the configuration written down in DESIGN.md 7, and the HIP
 * kernels are tested against THIS restatement.
 *
 * Conventions: positions in the unit cube [0,1]^3; features / weights / activations f16 with f32
 * accumulation; hash (x*1) ^ (y*2654435761) ^ (z*805459861) mod T.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint16_t orc_f2h(float x);
float orc_h2f(uint16_t h);
#define H2F orc_h2f
#define F2H orc_f2h

/* ------------------------------------------------------------------------------------------ */
/* grid layout                                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int n_levels, n_features, log2_hashmap, base_res;
  float per_level_scale;
} orc_grid_cfg;

/* scale_l = base * b^l - 1; res_l = ceil(scale_l) + 1; entries_l = min(ceil8(res^3), T) */
void orc_ngp_grid_layout(const orc_grid_cfg* c, float* scale, int* res, uint32_t* offset /* n_levels+1 */) {
  uint32_t off = 0;
  const uint32_t T = 1u << c->log2_hashmap;
  for (int l = 0; l < c->n_levels; l++) {
    scale[l] = exp2f(l * log2f(c->per_level_scale)) * (float)c->base_res - 1.0f;
    res[l] = (int)ceilf(scale[l]) + 1;
    uint64_t dense = (uint64_t)res[l] * res[l] * res[l];
    dense = (dense + 7) / 8 * 8;
    uint32_t n = dense > T ? T : (uint32_t)dense;
    offset[l] = off;
    off += n;
  }
  offset[c->n_levels] = off;
}

static uint32_t grid_index(uint32_t hashmap_size, uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t stride = 1, index = 0;
  const uint32_t p[3] = {x, y, z};
  for (int d = 0; d < 3 && stride <= hashmap_size; d++) {
    index += p[d] * stride;
    stride *= res;
  }
  if (hashmap_size < stride) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
  return index % hashmap_size;
}

/* positions [N,3] f32 in [0,1]; params f16 [total*F]; out [N, L*F] f16 (row-major per sample). */
void orc_ngp_encode_fwd(const orc_grid_cfg* c, const float* pos, const uint16_t* params, uint16_t* out, long N) {
  float scale[32];
  int res[32];
  uint32_t off[33];
  orc_ngp_grid_layout(c, scale, res, off);
  const int F = c->n_features, L = c->n_levels;
  for (long i = 0; i < N; i++)
    for (int l = 0; l < L; l++) {
      const uint32_t hs = off[l + 1] - off[l];
      float w[3];
      uint32_t g[3];
      for (int d = 0; d < 3; d++) {
        float p = fmaf(scale[l], pos[i * 3 + d], 0.5f);
        float fl = floorf(p);
        g[d] = (uint32_t)(int)fl;
        w[d] = p - fl;
      }
      float acc[8] = {0};
      for (int corner = 0; corner < 8; corner++) {
        float wt = 1.0f;
        uint32_t q[3];
        for (int d = 0; d < 3; d++) {
          if (corner & (1 << d)) { wt *= w[d]; q[d] = g[d] + 1; }
          else { wt *= 1.0f - w[d]; q[d] = g[d]; }
        }
        const uint32_t idx = grid_index(hs, (uint32_t)res[l], q[0], q[1], q[2]);
        for (int f = 0; f < F; f++) acc[f] = fmaf(wt, H2F(params[((long)off[l] + idx) * F + f]), acc[f]);
      }
      for (int f = 0; f < F; f++) out[i * (long)(L * F) + l * F + f] = F2H(acc[f]);
    }
}

/* grad_params f32 [total*F] += w_corner * dL/dout  (dLdout [N, L*F] f16). */
void orc_ngp_encode_bwd(const orc_grid_cfg* c, const float* pos, const uint16_t* dLdout, float* grad, long N) {
  float scale[32];
  int res[32];
  uint32_t off[33];
  orc_ngp_grid_layout(c, scale, res, off);
  const int F = c->n_features, L = c->n_levels;
  for (long i = 0; i < N; i++)
    for (int l = 0; l < L; l++) {
      const uint32_t hs = off[l + 1] - off[l];
      float w[3];
      uint32_t g[3];
      for (int d = 0; d < 3; d++) {
        float p = fmaf(scale[l], pos[i * 3 + d], 0.5f);
        float fl = floorf(p);
        g[d] = (uint32_t)(int)fl;
        w[d] = p - fl;
      }
      for (int corner = 0; corner < 8; corner++) {
        float wt = 1.0f;
        uint32_t q[3];
        for (int d = 0; d < 3; d++) {
          if (corner & (1 << d)) { wt *= w[d]; q[d] = g[d] + 1; }
          else { wt *= 1.0f - w[d]; q[d] = g[d]; }
        }
        const uint32_t idx = grid_index(hs, (uint32_t)res[l], q[0], q[1], q[2]);
        for (int f = 0; f < F; f++)
          grad[((long)off[l] + idx) * F + f] += wt * H2F(dLdout[i * (long)(L * F) + l * F + f]);
      }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* spherical harmonics degree 4 (16 coefficients) of a unit direction                         */
/* ------------------------------------------------------------------------------------------ */
void orc_ngp_sh16(const float* d, float* o) {
  const float x = d[0], y = d[1], z = d[2];
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

/* ------------------------------------------------------------------------------------------ */
/* fully-fused MLP pair (density net 32->64->16, colour net 32->64->64->16), f16 weights and   */
/* activations, f32 accumulation, ReLU hidden, no output activation.                          */
/* W row-major [out][in] f16.  Per sample: feat[32] f16, dir[3] f32 -> dens_out[16], rgb_out[16] (f16). */
/* Also returns the hidden activations needed by the backward pass.                            */
/* ------------------------------------------------------------------------------------------ */
static void dense(const uint16_t* W, int nout, int nin, const uint16_t* x, int relu, uint16_t* y) {
  for (int o = 0; o < nout; o++) {
    float acc = 0.0f;
    for (int k = 0; k < nin; k++) acc = fmaf(H2F(W[(long)o * nin + k]), H2F(x[k]), acc);
    if (relu && acc < 0.0f) acc = 0.0f;
    y[o] = F2H(acc);
  }
}

typedef struct {
  const uint16_t *W1, *W2, *W3, *W4, *W5; /* [64,32] [16,64] [64,32] [64,64] [16,64] */
} orc_mlp;

void orc_ngp_mlp_fwd(const orc_mlp* m, const uint16_t* feat, const float* dirs, long N, uint16_t* h1 /*[N,64]*/,
                     uint16_t* dens /*[N,16]*/, uint16_t* cin /*[N,32]*/, uint16_t* h3, uint16_t* h4 /*[N,64]*/,
                     uint16_t* rgb /*[N,16]*/) {
  for (long i = 0; i < N; i++) {
    dense(m->W1, 64, 32, feat + i * 32, 1, h1 + i * 64);
    dense(m->W2, 16, 64, h1 + i * 64, 0, dens + i * 16);
    float sh[16];
    orc_ngp_sh16(dirs + i * 3, sh);
    for (int k = 0; k < 16; k++) {
      cin[i * 32 + k] = dens[i * 16 + k];
      cin[i * 32 + 16 + k] = F2H(sh[k]);
    }
    dense(m->W3, 64, 32, cin + i * 32, 1, h3 + i * 64);
    dense(m->W4, 64, 64, h3 + i * 64, 1, h4 + i * 64);
    dense(m->W5, 16, 64, h4 + i * 64, 0, rgb + i * 16);
  }
}

/* Backward.  Inputs: the forward's activations, dL/drgb_out [N,16] f16 (only [0..2] non-zero in
 * practice) and dL/ddens_out [N,16] f16 (only [0]).  Outputs: dL/dfeat [N,32] f16 and f32 weight
 * gradients (accumulated, caller zeroes).  Activation gradients are rounded to f16 between layers
 * (as a fully-fused f16 MLP does); weight gradients are summed in double here.                     */
static void dense_bwd(const uint16_t* W, int nout, int nin, const uint16_t* x, const uint16_t* y, int relu,
                      const uint16_t* dy_in, uint16_t* dx, double* dW) {
  float dy[64];
  for (int o = 0; o < nout; o++) dy[o] = (relu && !(H2F(y[o]) > 0.0f)) ? 0.0f : H2F(dy_in[o]);
  for (int o = 0; o < nout; o++)
    for (int k = 0; k < nin; k++) dW[(long)o * nin + k] += (double)dy[o] * (double)H2F(x[k]);
  if (dx)
    for (int k = 0; k < nin; k++) {
      float acc = 0.0f;
      for (int o = 0; o < nout; o++) acc = fmaf(H2F(W[(long)o * nin + k]), H2F(F2H(dy[o])), acc);
      dx[k] = F2H(acc);
    }
}

void orc_ngp_mlp_bwd(const orc_mlp* m, const uint16_t* feat, long N, const uint16_t* h1, const uint16_t* dens,
                     const uint16_t* cin, const uint16_t* h3, const uint16_t* h4, const uint16_t* rgb,
                     const uint16_t* dLdrgb, const uint16_t* dLddens, uint16_t* dLdfeat, double* dW1, double* dW2,
                     double* dW3, double* dW4, double* dW5) {
  (void)rgb;
  for (long i = 0; i < N; i++) {
    uint16_t d4[64], d3[64], dc[32], dd[16], d1[64];
    dense_bwd(m->W5, 16, 64, h4 + i * 64, rgb + i * 16, 0, dLdrgb + i * 16, d4, dW5);
    dense_bwd(m->W4, 64, 64, h3 + i * 64, h4 + i * 64, 1, d4, d3, dW4);
    dense_bwd(m->W3, 64, 32, cin + i * 32, h3 + i * 64, 1, d3, dc, dW3);
    for (int k = 0; k < 16; k++) dd[k] = F2H(H2F(dc[k]) + H2F(dLddens[i * 16 + k]));
    dense_bwd(m->W2, 16, 64, h1 + i * 64, dens + i * 16, 0, dd, d1, dW2);
    dense_bwd(m->W1, 64, 32, feat + i * 32, h1 + i * 64, 1, d1, dLdfeat + i * 32, dW1);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* volume rendering of one training batch + loss + gradients w.r.t. the network outputs.       */
/* Samples of ray r are [ray_start[r], ray_start[r]+ray_n[r]).  Per sample: rgb_raw[16] and      */
/* dens_raw[16] f16 (network outputs), dt.  rgb = sigmoid(raw), sigma = exp(dens_raw[0]),        */
/* alpha = 1 - exp(-sigma*dt), T_k = prod (1-alpha).  Loss per ray (instant-ngp NeRF loss with   */
/* the NeRF-SLAM fork's depth term, nerf_fusion.py:100-101,285-289):                             */
/*   L = sum_c (rgb_c - gt_c)^2 / 3  +  lambda * (depth - gt_depth)^2 / gt_depth_cov   (if gt_depth > 0) */
/* averaged over the rays of the batch (loss_scale folded in by the caller).                     */
/* ------------------------------------------------------------------------------------------ */
void orc_ngp_composite_loss(const uint16_t* rgb_raw, const uint16_t* dens_raw, const float* dt, const float* tmid,
                            const int32_t* ray_start, const int32_t* ray_n, int n_rays, const float* gt_rgb,
                            const float* gt_depth, const float* gt_depth_cov, float depth_lambda, float loss_scale,
                            float* out_rgb, float* out_depth, float* loss, uint16_t* dLdrgb, uint16_t* dLddens) {
  double total = 0.0;
  for (int r = 0; r < n_rays; r++) {
    const int s0 = ray_start[r], n = ray_n[r];
    float T = 1.0f, C[3] = {0, 0, 0}, D = 0.0f;
    for (int k = 0; k < n; k++) {
      const long s = s0 + k;
      const float sigma = expf(H2F(dens_raw[s * 16]));
      const float alpha = 1.0f - expf(-sigma * dt[s]);
      const float wgt = alpha * T;
      for (int c = 0; c < 3; c++) C[c] += wgt / (1.0f + expf(-H2F(rgb_raw[s * 16 + c])));
      D += wgt * tmid[s];
      T *= 1.0f - alpha;
    }
    float dC[3], dD = 0.0f, l = 0.0f;
    for (int c = 0; c < 3; c++) {
      const float diff = C[c] - gt_rgb[r * 3 + c];
      l += diff * diff / 3.0f;
      dC[c] = 2.0f * diff / 3.0f;
    }
    if (gt_depth[r] > 0.0f && depth_lambda > 0.0f) {
      const float diff = D - gt_depth[r];
      l += depth_lambda * diff * diff / gt_depth_cov[r];
      dD = depth_lambda * 2.0f * diff / gt_depth_cov[r];
    }
    if (n >= 0) total += l; /* n < 0: ray refused by the marcher (batch full), not part of the batch */
    out_rgb[r * 3 + 0] = C[0]; out_rgb[r * 3 + 1] = C[1]; out_rgb[r * 3 + 2] = C[2];
    out_depth[r] = D;
    /* backward: d/d(rgb_k) = wgt_k * dC ; d/d(sigma_k) = dt_k * (T_k*(1-alpha_k) * (dC.rgb_k + dD*t_k) - suffix_k) */
    T = 1.0f;
    float C2[3] = {0, 0, 0}, D2 = 0.0f;
    const float sc = loss_scale / (float)n_rays;
    for (int k = 0; k < n; k++) {
      const long s = s0 + k;
      const float sigma = expf(H2F(dens_raw[s * 16]));
      const float alpha = 1.0f - expf(-sigma * dt[s]);
      const float wgt = alpha * T;
      float rgbv[3];
      for (int c = 0; c < 3; c++) rgbv[c] = 1.0f / (1.0f + expf(-H2F(rgb_raw[s * 16 + c])));
      for (int c = 0; c < 3; c++) C2[c] += wgt * rgbv[c];
      D2 += wgt * tmid[s];
      const float Tn = T * (1.0f - alpha);
      /* suffix = sum_{j>k} wgt_j * val_j = (total - prefix_k) */
      float g = 0.0f;
      for (int c = 0; c < 3; c++) g += dC[c] * (Tn * rgbv[c] - (C[c] - C2[c]));
      g += dD * (Tn * tmid[s] - (D - D2));
      const float dsigma = dt[s] * g;
      for (int c = 0; c < 16; c++) {
        float v = 0.0f;
        if (c < 3) v = sc * wgt * dC[c] * rgbv[c] * (1.0f - rgbv[c]);
        dLdrgb[s * 16 + c] = F2H(v);
        dLddens[s * 16 + c] = F2H(c == 0 ? sc * dsigma * sigma : 0.0f);
      }
      T = Tn;
    }
  }
  *loss = (float)(total / (double)n_rays);
}

/* ------------------------------------------------------------------------------------------ */
/* Adam (tiny-cuda-nn's: beta1 .9, beta2 .99, eps 1e-15, optional L2 on the weights), f32 master */
/* params, f16 working copy refreshed.                                                          */
/* ------------------------------------------------------------------------------------------ */
void orc_ngp_adam(float* master, uint16_t* half_params, const float* grad, float* m1, float* m2, long n, int step,
                  float lr, float beta1, float beta2, float eps, float l2, float grad_scale) {
  const float c1 = 1.0f - powf(beta1, (float)step), c2 = 1.0f - powf(beta2, (float)step);
  for (long i = 0; i < n; i++) {
    float g = grad[i] / grad_scale;
    if (g == 0.0f && l2 == 0.0f) { half_params[i] = F2H(master[i]); continue; } /* untouched hash entries keep their moments */
    g += l2 * master[i];
    m1[i] = beta1 * m1[i] + (1.0f - beta1) * g;
    m2[i] = beta2 * m2[i] + (1.0f - beta2) * g * g;
    master[i] -= lr * (m1[i] / c1) / (sqrtf(m2[i] / c2) + eps);
    half_params[i] = F2H(master[i]);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* occupancy-grid ray marching (one cascade-less grid of G^3 bits over the unit cube scaled by  */
/* `aabb_scale` cascades: cascade m covers [0.5 - 2^(m-1), 0.5 + 2^(m-1)]^3).                   */
/*   dt(t) = clamp(t*cone, min_step, max_step);  a sample at x = o + t d is emitted when the bit */
/*   of its cell in cascade mip(x, dt) is set; otherwise t advances by dt.                       */
/* ------------------------------------------------------------------------------------------ */
static int mip_of(const float* x, float dt, int G, int ncasc) {
  float m = fmaxf(fabsf(x[0] - 0.5f), fmaxf(fabsf(x[1] - 0.5f), fabsf(x[2] - 0.5f)));
  int mip = 0;
  while (mip < ncasc - 1 && m >= 0.5f * (float)(1 << mip)) mip++;
  /* coarser cascade when the step is larger than a cell of the current one */
  while (mip < ncasc - 1 && dt * (float)G > (float)(1 << mip)) mip++;
  return mip;
}

int orc_ngp_occupied(const uint8_t* bits, const float* x, float dt, int G, int ncasc) {
  const int mip = mip_of(x, dt, G, ncasc);
  const float s = 1.0f / (float)(1 << mip);
  int c[3];
  for (int d = 0; d < 3; d++) {
    const float p = (x[d] - 0.5f) * s + 0.5f;
    c[d] = (int)floorf(p * (float)G);
    if (c[d] < 0 || c[d] >= G) return 0;
  }
  const long idx = ((long)mip * G + c[2]) * G * G + (long)c[1] * G + c[0];
  return (bits[idx >> 3] >> (idx & 7)) & 1;
}

/* rays: origin[3], dir[3] (unit); returns the number of samples written for the ray (<= max_n).  */
int orc_ngp_march_ray(const uint8_t* bits, int G, int ncasc, const float* o, const float* d, float cone, float min_step,
                      float max_step, float t0, float t1, int max_n, float* pos, float* dts, float* ts) {
  int n = 0;
  float t = t0;
  while (t < t1 && n < max_n) {
    float dt = t * cone;
    dt = dt < min_step ? min_step : (dt > max_step ? max_step : dt);
    const float x[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    if (orc_ngp_occupied(bits, x, dt, G, ncasc)) {
      pos[n * 3 + 0] = x[0]; pos[n * 3 + 1] = x[1]; pos[n * 3 + 2] = x[2];
      dts[n] = dt;
      ts[n] = t;
      n++;
    }
    t += dt;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* training-ray sampling (restates csrc/ngp.hip:ngp_sample_rays_kernel; the pixel choice is this project's */
/* own: instant-ngp draws its training pixels from its own RNG stream)                                     */
/* ------------------------------------------------------------------------------------------ */
static uint32_t orc_pcg(uint32_t v) {
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

void orc_ngp_sample_rays(const float* images, const float* depths, const float* covs, const float* c2w, int n, int H,
                         int W, float fx, float fy, float cx, float cy, float box_lo, float box_hi, float near,
                         uint32_t seed, int R, float* rays_o, float* rays_d, float* t_range, float* gt_rgb,
                         float* gt_depth, float* gt_cov, int32_t* picks) {
  for (int r = 0; r < R; r++) {
    const uint32_t base = seed + (uint32_t)r * 3u;
    const int img = (int)(orc_pcg(base) % (uint32_t)n), u = (int)(orc_pcg(base + 1u) % (uint32_t)W),
              v = (int)(orc_pcg(base + 2u) % (uint32_t)H);
    picks[r * 3] = img; picks[r * 3 + 1] = u; picks[r * 3 + 2] = v;
    const float* M = c2w + (long)img * 12;
    const float dcx = ((float)u + 0.5f - cx) / fx, dcy = ((float)v + 0.5f - cy) / fy;
    float d[3], o[3];
    for (int k = 0; k < 3; k++) { d[k] = M[k * 4] * dcx + M[k * 4 + 1] * dcy + M[k * 4 + 2]; o[k] = M[k * 4 + 3]; }
    const float inv_n = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float tmin = -INFINITY, tmax = INFINITY;
    for (int k = 0; k < 3; k++) {
      d[k] *= inv_n;
      const float inv = 1.0f / (fabsf(d[k]) < 1e-9f ? 1e-9f : d[k]);
      const float t0 = (box_lo - o[k]) * inv, t1 = (box_hi - o[k]) * inv;
      tmin = fmaxf(tmin, fminf(t0, t1));
      tmax = fminf(tmax, fmaxf(t0, t1));
    }
    tmin = fmaxf(tmin, near);
    for (int k = 0; k < 3; k++) { rays_o[r * 3 + k] = o[k]; rays_d[r * 3 + k] = d[k]; }
    t_range[r * 2] = tmin; t_range[r * 2 + 1] = fmaxf(tmax, tmin);
    const long pix = ((long)img * H + v) * W + u;
    for (int k = 0; k < 3; k++) gt_rgb[r * 3 + k] = images[pix * 4 + k];
    gt_depth[r] = depths[pix];
    gt_cov[r] = fmaxf(covs[pix], 1e-6f);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* camera-pose refinement (restates csrc/ngp.hip: ngp_encode_bwd_input_kernel, ngp_camera_grad_kernel,      */
/* ngp_camera_step_kernel).  dLdout [N, L*F] f16 row-major here.                                           */
/* ------------------------------------------------------------------------------------------ */
void orc_ngp_encode_bwd_input(const orc_grid_cfg* c, const float* pos, const uint16_t* params, const uint16_t* dLdout,
                              float* dLdpos, long N) {
  float scale[32];
  int res[32];
  uint32_t off[33];
  orc_ngp_grid_layout(c, scale, res, off);
  const int L = c->n_levels;
  for (long i = 0; i < N; i++) {
    float g3[3] = {0, 0, 0};
    for (int l = 0; l < L; l++) {
      const float d0 = H2F(dLdout[i * (long)(L * 2) + l * 2]), d1 = H2F(dLdout[i * (long)(L * 2) + l * 2 + 1]);
      if (d0 == 0.0f && d1 == 0.0f) continue;
      const uint32_t hs = off[l + 1] - off[l];
      float w[3];
      uint32_t g[3];
      for (int d = 0; d < 3; d++) {
        float p = fmaf(scale[l], pos[i * 3 + d], 0.5f);
        float fl = floorf(p);
        g[d] = (uint32_t)(int)fl;
        w[d] = p - fl;
      }
      float s[8];
      for (int corner = 0; corner < 8; corner++) {
        const uint32_t idx = grid_index(hs, (uint32_t)res[l], g[0] + (corner & 1), g[1] + ((corner >> 1) & 1), g[2] + (corner >> 2));
        s[corner] = d0 * H2F(params[((long)off[l] + idx) * 2]) + d1 * H2F(params[((long)off[l] + idx) * 2 + 1]);
      }
      const float wx0 = 1.0f - w[0], wx1 = w[0], wy0 = 1.0f - w[1], wy1 = w[1], wz0 = 1.0f - w[2], wz1 = w[2];
      g3[0] += scale[l] * (wy0 * wz0 * (s[1] - s[0]) + wy1 * wz0 * (s[3] - s[2]) + wy0 * wz1 * (s[5] - s[4]) + wy1 * wz1 * (s[7] - s[6]));
      g3[1] += scale[l] * (wx0 * wz0 * (s[2] - s[0]) + wx1 * wz0 * (s[3] - s[1]) + wx0 * wz1 * (s[6] - s[4]) + wx1 * wz1 * (s[7] - s[5]));
      g3[2] += scale[l] * (wx0 * wy0 * (s[4] - s[0]) + wx1 * wy0 * (s[5] - s[1]) + wx0 * wy1 * (s[6] - s[2]) + wx1 * wy1 * (s[7] - s[3]));
    }
    dLdpos[i * 3] = g3[0]; dLdpos[i * 3 + 1] = g3[1]; dLdpos[i * 3 + 2] = g3[2];
  }
}

void orc_ngp_camera_gradient(const float* dLdpos, const float* tmid, const float* rays_d, const int32_t* ray_start,
                             const int32_t* ray_n, const int32_t* ray_img, float pos_inv, double* cam_grad, int R) {
  for (int r = 0; r < R; r++) {
    double o[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    for (int k = 0; k < ray_n[r]; k++) {
      const long s = (long)ray_start[r] + k;
      for (int a = 0; a < 3; a++) {
        const double gpa = (double)(dLdpos[s * 3 + a] * pos_inv);
        o[a] += gpa;
        d[a] += (double)tmid[s] * gpa;
      }
    }
    if (ray_n[r] <= 0) continue;
    const double x = rays_d[r * 3], y = rays_d[r * 3 + 1], z = rays_d[r * 3 + 2];
    double* gp = cam_grad + (long)ray_img[r] * 6;
    gp[0] += o[0]; gp[1] += o[1]; gp[2] += o[2];
    gp[3] += y * d[2] - z * d[1]; gp[4] += z * d[0] - x * d[2]; gp[5] += x * d[1] - y * d[0];
  }
}

void orc_ngp_camera_step(float* c2w, const float* cam_grad, float* m1, float* m2, int n, int step, float lr_pos,
                         float lr_rot, float beta1, float beta2, float eps, float grad_scale) {
  const float c1 = 1.0f - powf(beta1, (float)step), c2 = 1.0f - powf(beta2, (float)step);
  for (int i = 0; i < n; i++) {
    float st[6];
    int any = 0;
    for (int k = 0; k < 6; k++) {
      const float gk = cam_grad[i * 6 + k] * (1.0f / grad_scale);
      st[k] = 0.0f;
      if (gk != 0.0f) {
        any = 1;
        const float a = beta1 * m1[i * 6 + k] + (1.0f - beta1) * gk, b = beta2 * m2[i * 6 + k] + (1.0f - beta2) * gk * gk;
        m1[i * 6 + k] = a; m2[i * 6 + k] = b;
        st[k] = -(k < 3 ? lr_pos : lr_rot) * (a / c1) / (sqrtf(b / c2) + eps);
      }
    }
    if (!any) continue;
    float* M = c2w + (long)i * 12;
    M[3] += st[0]; M[7] += st[1]; M[11] += st[2];
    const float wx = st[3], wy = st[4], wz = st[5];
    const float th2 = wx * wx + wy * wy + wz * wz, th = sqrtf(th2);
    const float A = th < 1e-6f ? 1.0f - th2 / 6.0f : sinf(th) / th;
    const float B = th < 1e-6f ? 0.5f - th2 / 24.0f : (1.0f - cosf(th)) / th2;
    const float E[9] = {1.0f - B * (wy * wy + wz * wz), B * wx * wy - A * wz, B * wx * wz + A * wy,
                        B * wx * wy + A * wz, 1.0f - B * (wx * wx + wz * wz), B * wy * wz - A * wx,
                        B * wx * wz - A * wy, B * wy * wz + A * wx, 1.0f - B * (wx * wx + wy * wy)};
    float Rn[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Rn[a * 3 + b] = E[a * 3] * M[b] + E[a * 3 + 1] * M[4 + b] + E[a * 3 + 2] * M[8 + b];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) M[a * 4 + b] = Rn[a * 3 + b];
  }
}
