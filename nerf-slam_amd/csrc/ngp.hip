// ngp.hip -- instant-ngp style NeRF mapping kernels for gfx950 (MI355X): multiresolution hash-grid
// encoding (forward / backward), occupancy-grid ray marching, volume-rendering loss with the
// NeRF-SLAM depth + uncertainty term, Adam.  The fully-fused MLP lives in ngp_mlp.hip.
//
// Reference boundary: the `pyngp` calls of /root/reference/fusion/nerf_fusion.py:57-101, 285-303,
// 388-424.  The arithmetic behind that boundary is an un-vendored fork of
// NVIDIA instant-ngp (commit unknown): what is implemented here is the published algorithm with
// the configuration of DESIGN.md 7; parity is against oracle/ngp_oracle.c ("parity unpinned").
//
// Roofline: the hash-grid gather/scatter and Adam are HBM-bound (random 4-byte gathers out of a
// 2^19-entry table per fine level; fine levels miss the caches), marching and compositing are
// latency-bound.  Nothing here is GEMM-shaped.
#include <cstdlib>

#include "common.h"
#include <utility>

#include "ngp_adam.h"

struct GridCfg {
  int n_levels, n_features, log2_hashmap, base_res;
  float per_level_scale;
};

struct GridLayout {
  float scale[16];
  int res[16];
  uint32_t offset[17];
};

// Replicated accumulation tables for the coarse levels of the encode backward (see ngp_encode_bwd_kernel):
// level l has rep[l] (power of two) private copies of its table in the workspace at float offset ws_off[l].
struct ReplicaPlan {
  uint32_t rep[16];
  uint64_t ws_off[16];
  uint64_t total_floats;
};
#define NS_ENC_REPLICA_BUDGET (8u << 20)  // bytes of replicas per level
#define NS_ENC_REPLICA_MAX 64u
#define NS_ENC_REPLICA_TABLE_MAX (2u << 20)  // only tables up to this size are replicated

static void replica_plan_host(const GridLayout& g, int n_levels, ReplicaPlan& r) {
  uint64_t off = 0;
  for (int l = 0; l < 16; l++) {
    r.rep[l] = 1;
    r.ws_off[l] = off;
    if (l >= n_levels) continue;
    const uint64_t bytes = (uint64_t)(g.offset[l + 1] - g.offset[l]) * 2 * sizeof(float);
    uint32_t rep = 1;  // big (hashed) tables: contention is low, no replicas
    while (bytes <= NS_ENC_REPLICA_TABLE_MAX && rep * 2 <= NS_ENC_REPLICA_MAX &&
           (uint64_t)rep * 2 * bytes <= NS_ENC_REPLICA_BUDGET)
      rep *= 2;
    r.rep[l] = rep;
    if (rep > 1) off += (uint64_t)rep * (bytes / sizeof(float));
  }
  r.total_floats = off;
}

// identical to oracle/ngp_oracle.c:orc_ngp_grid_layout (and to what tiny-cuda-nn's GridEncoding does)
static int grid_layout_host(const GridCfg& c, GridLayout& g) {
  if (c.n_levels < 1 || c.n_levels > 16 || c.n_features != 2) return NS_ENOSUP;
  uint32_t off = 0;
  const uint32_t T = 1u << c.log2_hashmap;
  for (int l = 0; l < c.n_levels; l++) {
    g.scale[l] = exp2f(l * log2f(c.per_level_scale)) * (float)c.base_res - 1.0f;
    g.res[l] = (int)ceilf(g.scale[l]) + 1;
    uint64_t dense = (uint64_t)g.res[l] * g.res[l] * g.res[l];
    dense = (dense + 7) / 8 * 8;
    const uint32_t n = dense > T ? T : (uint32_t)dense;
    g.offset[l] = off;
    off += n;
  }
  g.offset[c.n_levels] = off;
  for (int l = c.n_levels; l < 16; l++) {
    g.scale[l] = 0;
    g.res[l] = 1;
    g.offset[l + 1] = off;
  }
  return NS_OK;
}

__device__ __forceinline__ uint32_t grid_index(uint32_t hashmap_size, uint32_t res, uint32_t x, uint32_t y,
                                               uint32_t z) {
  // dense (x + y*res + z*res^2) while it fits, spatial hash otherwise
  uint32_t stride = 1, index = 0;
  index += x * stride;
  stride *= res;
  if (stride <= hashmap_size) {
    index += y * stride;
    stride *= res;
    if (stride <= hashmap_size) {
      index += z * stride;
      stride *= res;
    }
  }
  if (hashmap_size < stride) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
  return index % hashmap_size;
}

// Same index as grid_index() at a fraction of its cost (the generic form ends in an integer modulo by a run-time value:
// ~40 VALU instructions per corner, 8 corners x 16 levels per sample -- it, not the gathers, was most of the encode
// kernels' time).  A level is either hashed -- its table size is then 2^log2_hashmap, the modulo is a mask -- or dense,
// where x + y res + z res^2 < 2 * table size (a far-border corner can reach res + res^2 + res^3), so one conditional
// subtraction is the modulo.  `hashed` is uniform per level.
__device__ __forceinline__ uint32_t grid_index_lvl(bool hashed, uint32_t hs, uint32_t res, uint32_t x, uint32_t y, uint32_t z) {
  if (hashed) return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & (hs - 1u);
  // Clamp first: the bound above holds for coordinates <= res only.  A position outside the unit cube (the up-to-7 tail slots
  // of the 8-rounded sample count hold whatever the sample array held before -- after render() that is SCENE coordinates, e.g.
  // -1.2) gives a negative cell coordinate, i.e. a huge unsigned one, and without a modulo an index far outside the table:
  // an intermittent memory fault in the training steps that follow a render() (found by tests/test_ngp_gpu.py in a loop).
  x = min(x, res); y = min(y, res); z = min(z, res);
  const uint32_t idx = x + (y + z * res) * res;
  return idx >= hs ? idx - hs : idx;
}
__device__ __forceinline__ bool grid_level_hashed(uint32_t hs, uint32_t res) {
  return (uint64_t)res * res * res > (uint64_t)hs;
}

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

// Which workgroup gathers from which level's table.  All lanes of a workgroup read ONE level; the 8 XCDs have 8 separate L2s,
// and with the levels walked in order by the whole chip (round 2: grid (blocks, levels)) every XCD pulled every hashed level's
// 2 MB table through its own L2 -- 8 x 23 MB, most of the kernel's 230 MB of fetches.  Workgroup ids are dealt round-robin to
// the XCDs (observed placement: speed only, any placement is correct), so workgroup b belongs to XCD b % 8 and takes item b / 8
// of that XCD's list: first ALL sample blocks of the one hashed level this XCD owns (the 8 finest hashed levels: their tables
// are fetched once, by one L2), then every 8th block of the levels that are shared (coarser hashed levels, whose neighbouring
// samples hit the same lines anyway, and the dense ones, which stay resident).
struct EncFwdSched {
  int own[8];       // XCD -> the level it owns, or -1
  int shared[16];   // the other levels, costliest first
  int n_shared;
  int nblk, nblk8;  // sample blocks of 256; ceil(nblk / 8)
  int seg;          // items of the owned segment: nblk, or 0 when no level is owned
};

__global__ __launch_bounds__(256) void ngp_encode_fwd_kernel(GridLayout g, const float* __restrict__ pos,
                                                             const h2_t* __restrict__ params,
                                                             h2_t* __restrict__ out, long N, int L, int unit_major,
                                                             const int* __restrict__ n_dev, _Float16* __restrict__ jacT,
                                                             EncFwdSched sc) {
  const int xcd = blockIdx.x & 7, item = blockIdx.x >> 3;
  int l, blk;
  if (item < sc.seg) {
    l = sc.own[xcd];
    blk = item;
    if (l < 0) return;
  } else {
    const int k = item - sc.seg;
    l = sc.shared[k / sc.nblk8];
    blk = (k % sc.nblk8) * 8 + xcd;
    if (blk >= sc.nblk) return;
  }
  const long i = (long)blk * 256 + threadIdx.x;
  // N stays the row stride of the unit-major output.  The device count is rounded up to 8 like in the MLP kernels, which
  // read the features of the (up to 7) tail slots: left unwritten they are whatever the allocation held -- NaN bits there
  // turn the weight gradient into NaN even though the tail's upstream gradient is zero (0 x NaN).
  if (i >= N || (n_dev != nullptr && i >= (((long)*n_dev + 7) & ~7L))) return;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const float scale = g.scale[l];
  const uint32_t res = (uint32_t)g.res[l];
  const bool hashed = grid_level_hashed(hs, res);
  float w[3];
  uint32_t c[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const float p = fmaf(scale, pos[i * 3 + d], 0.5f);
    const float fl = floorf(p);
    c[d] = (uint32_t)(int)fl;
    w[d] = p - fl;
  }
  const h2_t* __restrict__ tab = params + g.offset[l];
  h2_t v[8];
  // The gather is bound by the number of L2 requests, not by bytes.  The two x-neighbours of a cell edge are adjacent
  // table entries on every dense level (stride 1 in x) and, on the hashed levels, whenever x is even (x and x + 1 differ
  // in bit 0 only, and x enters the hash with multiplier 1): one 8-byte load then serves both corners.
#pragma unroll
  for (int yz = 0; yz < 4; yz++) {
    const uint32_t i0 = grid_index_lvl(hashed, hs, res, c[0], c[1] + (yz & 1), c[2] + (yz >> 1));
    const uint32_t i1 = grid_index_lvl(hashed, hs, res, c[0] + 1, c[1] + (yz & 1), c[2] + (yz >> 1));
    const uint32_t lo = min(i0, i1);
    // (round 4, measured: non-temporal gathers on the hashed levels -- no L1 allocation for lines of which 4 bytes are used --
    //  59 -> 175 us: neighbouring samples of a ray DO share lines often enough for the L1 to matter.  Also measured: TWO
    //  (block, level) items per workgroup, both positions and then both sets of gathers in flight together -- half the
    //  workgroups, the same two round trips each: 59.7-61.1 -> 61.7 us.  The kernel is not a chain of round trips; it sits on
    //  the rate at which the L1s take scattered 4- / 8-byte requests and fill lines.)
    if (max(i0, i1) - lo == 1u) {
      struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };
      const Pair pr = *reinterpret_cast<const Pair*>(tab + lo);
      v[2 * yz] = __builtin_bit_cast(h2_t, i0 == lo ? pr.a : pr.b);
      v[2 * yz + 1] = __builtin_bit_cast(h2_t, i0 == lo ? pr.b : pr.a);
    } else {
      v[2 * yz] = tab[i0];
      v[2 * yz + 1] = tab[i1];
    }
  }
  float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    float wt = 1.0f;
    wt *= (corner & 1) ? w[0] : 1.0f - w[0];
    wt *= (corner & 2) ? w[1] : 1.0f - w[1];
    wt *= (corner & 4) ? w[2] : 1.0f - w[2];
    a0 = fmaf(wt, (float)v[corner][0], a0);
    a1 = fmaf(wt, (float)v[corner][1], a1);
  }
  if (jacT != nullptr) {
    // d(feature f)/d(position d) / scale, from the corner values already in registers: the pose refinement's input gradient
    // then is a dot product with dL/dfeature (ngp_encode_jac_dot_kernel) instead of a second gather of the table
    // (ngp_encode_bwd_input_kernel: 8 x 16 gathers per sample again, 104 us per step).  The level's scale is applied there
    // (in f32): scale x value differences would leave the f16 range on the fine levels.
    const float wx0 = 1.0f - w[0], wx1 = w[0], wy0 = 1.0f - w[1], wy1 = w[1], wz0 = 1.0f - w[2], wz1 = w[2];
#pragma unroll
    for (int f = 0; f < 2; f++) {
      float s8[8];
#pragma unroll
      for (int corner = 0; corner < 8; corner++) s8[corner] = (float)v[corner][f];
      const float jx = wy0 * wz0 * (s8[1] - s8[0]) + wy1 * wz0 * (s8[3] - s8[2]) + wy0 * wz1 * (s8[5] - s8[4]) + wy1 * wz1 * (s8[7] - s8[6]);
      const float jy = wx0 * wz0 * (s8[2] - s8[0]) + wx1 * wz0 * (s8[3] - s8[1]) + wx0 * wz1 * (s8[6] - s8[4]) + wx1 * wz1 * (s8[7] - s8[5]);
      const float jz = wx0 * wy0 * (s8[4] - s8[0]) + wx1 * wy0 * (s8[5] - s8[1]) + wx0 * wy1 * (s8[6] - s8[2]) + wx1 * wy1 * (s8[7] - s8[3]);
      jacT[(long)(6 * l + 3 * f + 0) * N + i] = (_Float16)jx;
      jacT[(long)(6 * l + 3 * f + 1) * N + i] = (_Float16)jy;
      jacT[(long)(6 * l + 3 * f + 2) * N + i] = (_Float16)jz;
    }
  }
  if (unit_major) {  // [2L][N]: 128 contiguous bytes per wave and feature
    _Float16* o = reinterpret_cast<_Float16*>(out);
    o[(long)(2 * l) * N + i] = (_Float16)a0;
    o[(long)(2 * l + 1) * N + i] = (_Float16)a1;
  } else {
    h2_t o = {(_Float16)a0, (_Float16)a1};
    out[i * L + l] = o;
  }
}

// Packed fixed-point gradient format (fixed_scale > 0): one 64-bit word per table entry holds both features
// as signed Q(fixed_scale) 32-bit fields, word = round(g0 * S) + (round(g1 * S) << 32).  One 64-bit integer
// atomic replaces two float atomics (the scatter is atomic-rate bound on the hashed levels), and integer
// addition commutes: the accumulated gradient is independent of the execution order (bit-reproducible).
__device__ __forceinline__ unsigned long long pack_fixed(float g0, float g1, float S) {
  const long long lo = (long long)__float2int_rn(g0 * S), hi = (long long)__float2int_rn(g1 * S);
  return (unsigned long long)(lo + (hi << 32));
}
__device__ __forceinline__ void unpack_fixed(unsigned long long w, float inv_S, float& g0, float& g1) {
  const int lo = (int)(unsigned)(w & 0xffffffffull);
  const int hi = (int)(((long long)w - (long long)lo) >> 32);
  g0 = (float)lo * inv_S;
  g1 = (float)hi * inv_S;
}

// (adam_apply(): ngp_adam.h -- ONE definition for this file and for the MLP's fused optimiser step in ngp_mlp.hip)

// Backward of the encode: scatter-add of the 8 trilinear corner contributions per (sample, level).
// Samples arrive in ray order, so on the coarse levels long runs of consecutive lanes fall into the SAME
// cell (level 0: every sample of a small scene hits a few dozen table entries); plain atomics then
// serialise on a handful of L2 addresses (measured 1.4 ms per step, 43 % of the training step).  The wave
// therefore run-length-reduces first: lanes with equal cell coordinates form a run (heads from one
// ballot), a segmented scan sums the 16 weighted contributions inside each run and only the run's last
// lane issues the 16 atomics.  Waves with (almost) no sharing skip the scan (wave-uniform decision).
#ifdef NS_TEST_VARIANTS   // comparison kernel: libnerfslam_hip_variants.so only (common.h)
__global__ __launch_bounds__(256) void ngp_encode_bwd_kernel(GridLayout g, const float* __restrict__ pos,
                                                             const h2_t* __restrict__ dLdout,
                                                             float* __restrict__ grad, long N, int L, int level0,
                                                             ReplicaPlan rp, float* __restrict__ ws, int unit_major,
                                                             float fixed_scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int l = blockIdx.y + level0;
  float d0 = 0.0f, d1 = 0.0f;
  if (i < N) {
    if (unit_major) {
      const _Float16* dp = reinterpret_cast<const _Float16*>(dLdout);
      d0 = (float)dp[(long)(2 * l) * N + i];
      d1 = (float)dp[(long)(2 * l + 1) * N + i];
    } else {
      const h2_t d = dLdout[i * L + l];
      d0 = (float)d[0];
      d1 = (float)d[1];
    }
  }
  const bool valid = d0 != 0.0f || d1 != 0.0f;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const float scale = g.scale[l];
  const uint32_t res = (uint32_t)g.res[l];
  float w[3] = {0.0f, 0.0f, 0.0f};
  uint32_t c[3] = {0u, 0u, 0u};
  if (valid) {
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
      const float p = fmaf(scale, pos[i * 3 + dd], 0.5f);
      const float fl = floorf(p);
      c[dd] = (uint32_t)(int)fl;
      w[dd] = p - fl;
    }
  }
  float v[16];
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    float wt = 1.0f;
    wt *= (corner & 1) ? w[0] : 1.0f - w[0];
    wt *= (corner & 2) ? w[1] : 1.0f - w[1];
    wt *= (corner & 4) ? w[2] : 1.0f - w[2];
    v[corner * 2] = wt * d0;
    v[corner * 2 + 1] = wt * d1;
  }
  // runs of equal cells
  const uint32_t p0 = __shfl_up(c[0], 1), p1 = __shfl_up(c[1], 1), p2 = __shfl_up(c[2], 1);
  const int pv = __shfl_up((int)valid, 1);
  const bool head = lane == 0 || !valid || !pv || p0 != c[0] || p1 != c[1] || p2 != c[2];
  const uint64_t hm = __ballot(head);
  bool issue = valid;
  if (__popcll(hm) <= 40) {  // wave-uniform: enough sharing to pay for the scan
    const int start = 63 - __clzll(hm & (~0ull >> (63 - lane)));
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const bool take = lane - d >= start;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const float u = __shfl_up(v[q], d);
        if (take) v[q] += u;
      }
    }
    issue = valid && (lane == 63 || ((hm >> (lane + 1)) & 1ull));
  }
  if (!issue) return;
  // coarse levels: all samples of a scene fall on a few hundred entries and device-scope atomics on one address
  // serialise at the memory side (measured: 0.4 ms for level 0 alone); spread them over rep[l] private tables
  const bool replica = ws != nullptr && rp.rep[l] > 1;
  float* __restrict__ tab = replica ? ws + rp.ws_off[l] + (uint64_t)(blockIdx.x & (rp.rep[l] - 1)) * hs * 2
                                    : grad + (long)g.offset[l] * 2;
  if (!replica && fixed_scale > 0.0f) {
    unsigned long long* __restrict__ tab64 = reinterpret_cast<unsigned long long*>(tab);
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const uint32_t idx = grid_index(hs, res, c[0] + (corner & 1), c[1] + ((corner >> 1) & 1), c[2] + (corner >> 2));
      atomicAdd(&tab64[idx], pack_fixed(v[corner * 2], v[corner * 2 + 1], fixed_scale));
    }
    return;
  }
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const uint32_t idx = grid_index(hs, res, c[0] + (corner & 1), c[1] + ((corner >> 1) & 1), c[2] + (corner >> 2));
    atomicAdd(&tab[(long)idx * 2 + 0], v[corner * 2]);
    atomicAdd(&tab[(long)idx * 2 + 1], v[corner * 2 + 1]);
  }
}
#endif  // NS_TEST_VARIANTS

// ---------------------------------------------------------------------------------------------
// Encode backward WITHOUT global atomics (round 2; what ns_ngp_encode_backward launches).
//
// Why: on this 8-XCD part a global atomic is executed at the memory side (the per-XCD L2s are not coherent: an atomic
// drops the line from L2 and travels the fabric), so the scatter above is bound by the fabric's atomic rate -- 0.35-0.75 ms
// per 2^18 samples and scene dependent (hot cells serialise).  Here every table entry is OWNED by exactly one workgroup:
// a task = (level, slice of <= 16384 consecutive entries = 128 KB of LDS); the workgroup scans ALL samples of the level,
// recomputes the 8 corner indices (a dozen integer ops each), and accumulates the corners that fall into its slice with
// LDS atomics (ds_add_u64 on the packed fixed-point word / ds_add_f32 pairs); at the end the slice leaves with plain
// coalesced read-modify-writes of its non-zero entries.  32 slices per hashed level -> the index arithmetic is done
// 32x redundantly (~0.1 us of VALU per sample and level), which is cheaper than one fabric atomic per corner; the samples
// (12 B position + 4 B gradient per level) are re-read from L2.  Time is independent of how the samples cluster.
// Dense (coarse) levels have few slices but every sample hits them 8 times, so their tasks are split over NS_ENC_PARTS
// sample ranges, each with a private LDS slice, merged by (few) global atomics on the non-zero entries; the wave-level
// run-length reduction of the atomic kernel is kept for them (long runs of lanes in the same cell).
// Task order: block b serves virtual task (b & 7) * (grid / 8) + (b >> 3): the blocks of one XCD (b % 8, observed
// placement -- speed only, any placement is correct) work on the same one or two levels, whose gradient rows stay in
// that XCD's L2.
// ---------------------------------------------------------------------------------------------
#define NS_ENC_SLICE 16384
#define NS_ENC_PARTS 4
#define NS_ENC_PARTS_BINNED 15   // dense levels of several slices
#define NS_ENC_PARTS_COARSE 64   // single-slice (coarsest) dense levels: see the dense branch of the kernel
struct EncBwdPlan {
  int first[17];   // first virtual task of the k-th level in task order; first[n_levels] = number of tasks
  int level[16];   // k -> level
  int slices[16];  // per LEVEL
  int parts[16];   // per LEVEL
  long plane_base[16];  // per LEVEL: first entry of the level's partial planes (parts[l] planes of its table size each)
};

static bool level_is_hashed(const GridLayout& g, int l) {
  return (uint64_t)g.res[l] * g.res[l] * g.res[l] > (uint64_t)(g.offset[l + 1] - g.offset[l]);
}

// dense_only: the hashed levels are handled by the binned path below; the dense levels then get more (smaller) parts
static int enc_bwd_plan_host(const GridLayout& g, int n_levels, EncBwdPlan& p, bool dense_only, int parts_coarse = NS_ENC_PARTS_COARSE,
                             int parts_multi = NS_ENC_PARTS_BINNED, int max_dense_level = 16,      // dense levels >= it: not planned
                             int slice_entries = NS_ENC_SLICE) {
  int k = 0, t = 0;
  for (int l = 0; l < 16; l++) p.slices[l] = p.parts[l] = 0;
  for (int pass = dense_only ? 1 : 0; pass < 2; pass++)  // hashed (1 part) levels first, then the dense ones, finest first
    for (int li = 0; li < n_levels; li++) {                // (their tasks are the longest: they should start first)
      const int l = pass == 0 ? li : n_levels - 1 - li;
      const uint32_t hs = g.offset[l + 1] - g.offset[l];
      const bool hashed = level_is_hashed(g, l);
      if (hashed != (pass == 0)) continue;
      if (!hashed && l >= max_dense_level) continue;
      p.slices[l] = (int)((hs + slice_entries - 1) / slice_entries);
      p.parts[l] = hashed ? 1 : (!dense_only ? NS_ENC_PARTS : (p.slices[l] == 1 ? parts_coarse : parts_multi));
      p.level[k] = l;
      p.first[k] = t;
      t += p.slices[l] * p.parts[l];
      k++;
    }
  for (; k <= 16; k++) {
    p.first[k] = t;
    if (k < 16) p.level[k] = 0;
  }
  long base = 0;
  for (int l = 0; l < 16; l++) {
    p.plane_base[l] = base;
    if (l < n_levels && !level_is_hashed(g, l) && l < max_dense_level) base += (long)p.parts[l] * (long)(g.offset[l + 1] - g.offset[l]);
  }
  return t;
}

template <bool FIXED>
__global__ __launch_bounds__(1024) void ngp_encode_bwd_lds_kernel(GridLayout g, EncBwdPlan plan, const float* __restrict__ pos,
                                                                  const h2_t* __restrict__ dLdout, float* __restrict__ grad,
                                                                  long N, int L, int n_levels, int unit_major,
                                                                  float fixed_scale, unsigned long long* __restrict__ partial,
                                                                  const int* __restrict__ n_dev) {
  __shared__ unsigned long long tab[NS_ENC_SLICE];  // packed fixed-point words, or float2 bit patterns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = gridDim.x >> 3;
  const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (v >= plan.first[16]) return;   // first[k] = number of tasks for every k past the planned levels
  int k = 0;
  while (v >= plan.first[k + 1]) k++;
  const int l = plan.level[k];
  const int local = v - plan.first[k];
  const int nparts = plan.parts[l];
  const int slice = local / nparts, part = local - slice * nparts;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const uint32_t lo = (uint32_t)slice * NS_ENC_SLICE;
  const uint32_t cnt = min((uint32_t)NS_ENC_SLICE, hs - lo);
  const float scale = g.scale[l];
  const uint32_t res = (uint32_t)g.res[l];
  const bool hashed = (uint64_t)res * res * res > hs;
  for (uint32_t e = tid; e < cnt; e += 1024) tab[e] = 0ull;
  __syncthreads();
  // sample range of this part, in units of 64-sample wave chunks
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;   // N stays the row stride of a unit-major gradient
  const long chunks = (nvalid + 63) >> 6;
  const long c_lo = chunks * part / nparts, c_hi = chunks * (part + 1) / nparts;
  const _Float16* dpu = reinterpret_cast<const _Float16*>(dLdout);
  float* tabf = reinterpret_cast<float*>(tab);
  // software prefetch: the loads of the NEXT chunk are in flight while this one is accumulated (a wave otherwise waits a
  // full L2 round trip per 64 samples: four waves per SIMD do not hide it)
  _Float16 nd0 = (_Float16)0, nd1 = (_Float16)0;
  float npos[3] = {0.0f, 0.0f, 0.0f};
  auto fetch = [&](long chn) {
    const long in = (chn << 6) + lane;
    nd0 = nd1 = (_Float16)0;
    if (chn < c_hi && in < nvalid) {
      if (unit_major) {
        nd0 = dpu[(long)(2 * l) * N + in];
        nd1 = dpu[(long)(2 * l + 1) * N + in];
      } else {
        const h2_t d = dLdout[in * L + l];
        nd0 = d[0];
        nd1 = d[1];
      }
      npos[0] = pos[in * 3];
      npos[1] = pos[in * 3 + 1];
      npos[2] = pos[in * 3 + 2];
    }
  };
  fetch(c_lo + wave);
  for (long ch = c_lo + wave; ch < c_hi; ch += 16) {
    const float d0 = (float)nd0, d1 = (float)nd1;
    const float cpos[3] = {npos[0], npos[1], npos[2]};
    fetch(ch + 16);
    const bool valid = d0 != 0.0f || d1 != 0.0f;
    if (__ballot(valid) == 0ull) continue;
    float w[3] = {0.0f, 0.0f, 0.0f};
    uint32_t c[3] = {0u, 0u, 0u};
    if (valid) {
#pragma unroll
      for (int dd = 0; dd < 3; dd++) {
        const float p = fmaf(scale, cpos[dd], 0.5f);
        const float fl = floorf(p);
        c[dd] = hashed ? (uint32_t)(int)fl : min((uint32_t)(int)fl, res - 1u);   // (dense: see grid_index_lvl)
        w[dd] = p - fl;
      }
    }
    if (hashed) {
      // idx = (x ^ y P1 ^ z P2) & (hs - 1): the 8 corners share the six partial products
      const uint32_t hx[2] = {c[0], c[0] + 1u};
      const uint32_t hy0 = c[1] * 2654435761u, hz0 = c[2] * 805459861u;
      const uint32_t hy[2] = {hy0, hy0 + 2654435761u};
      const uint32_t hz[2] = {hz0, hz0 + 805459861u};
      const float wx[2] = {1.0f - w[0], w[0]}, wy[2] = {1.0f - w[1], w[1]}, wz[2] = {1.0f - w[2], w[2]};
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        const uint32_t idx = (hx[corner & 1] ^ hy[(corner >> 1) & 1] ^ hz[corner >> 2]) & (hs - 1u);
        const uint32_t rel = idx - lo;
        if (valid && rel < cnt) {
          const float wt = wx[corner & 1] * wy[(corner >> 1) & 1] * wz[corner >> 2];
          if (FIXED) {
            atomicAdd(&tab[rel], pack_fixed(wt * d0, wt * d1, fixed_scale));
          } else {
            atomicAdd(&tabf[2 * rel], wt * d0);
            atomicAdd(&tabf[2 * rel + 1], wt * d1);
          }
        }
      }
      continue;
    }
    // dense level: every lane adds its 8 corners.  Lanes of a wave that hit the SAME LDS address are served one per cycle
    // (measured: ~85 cycles per 64-lane ds_add when a whole wave sits in one level-0 cell), so a task costs about one cycle
    // per (sample, corner) it accumulates: the coarsest levels -- where every sample hits the single slice 8 times -- are
    // therefore split over more sample parts (NS_ENC_PARTS_COARSE).  Tried and dropped: the run-length scan of the atomic
    // kernel (96 ds_bpermute per wave) and a cell-by-cell DPP reduction (16 wave sums per distinct cell): both cost more
    // VALU time at four waves per SIMD than the conflicts they remove.
    if (valid) {
      const float wx[2] = {1.0f - w[0], w[0]}, wy[2] = {1.0f - w[1], w[1]}, wz[2] = {1.0f - w[2], w[2]};
      // dense index x + y res + z res^2 of the 8 corners from one base (grid_index() re-derives it per corner and ends in
      // an integer modulo: ~50 instructions per corner, which made THIS the cost of the dense levels).  A corner on the
      // far border can reach res + res^2 + res^3 < 2 hs: one conditional subtraction is the modulo.
      const uint32_t r2 = res * res;
      const uint32_t base = c[0] + c[1] * res + c[2] * r2;
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        uint32_t idx = base + (corner & 1) + ((corner >> 1) & 1) * res + (corner >> 2) * r2;
        idx = idx >= hs ? idx - hs : idx;
        const uint32_t rel = idx - lo;
        if (rel < cnt) {
          const float wt = wx[corner & 1] * wy[(corner >> 1) & 1] * wz[corner >> 2];
          if (FIXED) {
            atomicAdd(&tab[rel], pack_fixed(wt * d0, wt * d1, fixed_scale));
          } else {
            atomicAdd(&tabf[2 * rel], wt * d0);
            atomicAdd(&tabf[2 * rel + 1], wt * d1);
          }
        }
      }
    }
  }
  __syncthreads();
  // flush: this workgroup is the only writer of its entries (nparts == 1) -> plain read-modify-write of the non-zero ones
  unsigned long long* __restrict__ g64 = reinterpret_cast<unsigned long long*>(grad) + g.offset[l] + lo;
  if (FIXED && partial != nullptr && nparts > 1) {
    // multi-part (dense) slices with a workspace: every part stores its whole slice (zeros included) in its own plane and
    // ngp_enc_dense_reduce_kernel adds the planes -- the merge by global atomics cost more than the accumulation itself
    // (up to 15 same-address memory-side atomics per touched entry)
    unsigned long long* __restrict__ dst = partial + plan.plane_base[l] + (long)part * hs + lo;
    for (uint32_t e = tid; e < cnt; e += 1024) dst[e] = tab[e];
    return;
  }
  for (uint32_t e = tid; e < cnt; e += 1024) {
    const unsigned long long word = tab[e];
    if (word == 0ull) continue;
    if (FIXED) {
      if (nparts == 1) g64[e] += word; else atomicAdd(&g64[e], word);
    } else {
      float* gp = reinterpret_cast<float*>(g64 + e);
      const float a = __builtin_bit_cast(float, (uint32_t)(word & 0xffffffffull));
      const float b = __builtin_bit_cast(float, (uint32_t)(word >> 32));
      if (nparts == 1) {
        gp[0] += a;
        gp[1] += b;
      } else {
        atomicAdd(&gp[0], a);
        atomicAdd(&gp[1], b);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense levels, RUN-LENGTH variant (packed fixed point, unit-major gradient): same tasks (level, slice, part) and the same
// LDS slice as the kernel above, but a LANE owns a run of 8 CONSECUTIVE samples instead of one sample of 64 consecutive ones.
// Consecutive samples are consecutive steps along one ray: on the coarse levels (cell = 1/16 .. 1/56 of the cube, a step
// ~1/500) they stay in one cell for many steps, so the lane sums their packed contributions per corner in registers and
// issues 8 LDS atomics per CELL CHANGE instead of 8 per sample; and the 64 lanes of a wave now sit on 64 different ray
// segments, so the atomics that remain rarely meet on one address (the kernel above is bound by exactly that: ~1 cycle
// per (sample, corner) when a wave sits in one level-0 cell).  Integer sums: bit-identical to every other path.
// A run is 96 B of positions + 2 x 16 B of gradient per lane, loaded as 16-byte vectors one chunk ahead.
// ---------------------------------------------------------------------------------------------
#define NS_RL_K 8
template <int SLICE, int THREADS>
__global__ __launch_bounds__(THREADS) void ngp_encode_bwd_dense_rl_kernel(GridLayout g, EncBwdPlan plan, const float* __restrict__ pos,
                                                                       const _Float16* __restrict__ dpu, float* __restrict__ grad,
                                                                       long N, float fixed_scale,
                                                                       unsigned long long* __restrict__ partial,
                                                                       const int* __restrict__ n_dev) {
  __shared__ unsigned long long tab[SLICE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = gridDim.x >> 3;
  const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (v >= plan.first[16]) return;
  int k = 0;
  while (v >= plan.first[k + 1]) k++;
  const int l = plan.level[k];
  const int local = v - plan.first[k];
  const int nparts = plan.parts[l];
  const int slice = local / nparts, part = local - slice * nparts;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const uint32_t lo = (uint32_t)slice * SLICE;
  const uint32_t cnt = min((uint32_t)SLICE, hs - lo);
  const float scale = g.scale[l];
  const uint32_t res = (uint32_t)g.res[l], r2 = res * res;
  for (uint32_t e = tid; e < cnt; e += THREADS) tab[e] = 0ull;
  __syncthreads();
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  const long chunks = (nvalid + 64 * NS_RL_K - 1) / (64 * NS_RL_K);   // a wave chunk = 64 runs of 8 samples
  const long c_lo = chunks * part / nparts, c_hi = chunks * (part + 1) / nparts;
  const _Float16* __restrict__ d0p = dpu + (long)(2 * l) * N;
  const _Float16* __restrict__ d1p = dpu + (long)(2 * l + 1) * N;
  // cells whose 8 corners all lie outside [lo, lo + cnt) are skipped before any weight is formed (multi-slice levels: a
  // slice is a few z layers of the level).  span = offset of the far corner; a wrapped corner (index >= hs) re-enters at 0.
  const uint32_t span = 1u + res + r2;
  float4 np[6];
  uint4 ng0 = make_uint4(0, 0, 0, 0), ng1 = make_uint4(0, 0, 0, 0);
  auto fetch = [&](long chn) {
    const long i0 = (chn * 64 + lane) * NS_RL_K;
    ng0 = ng1 = make_uint4(0, 0, 0, 0);
    if (chn < c_hi && i0 < nvalid) {                   // (N is a multiple of 8: a run that starts below N ends below N)
      const float4* __restrict__ pp = reinterpret_cast<const float4*>(pos + i0 * 3);
#pragma unroll
      for (int q = 0; q < 6; q++) np[q] = pp[q];
      ng0 = *reinterpret_cast<const uint4*>(d0p + i0);
      ng1 = *reinterpret_cast<const uint4*>(d1p + i0);
    }
  };
  fetch(c_lo + wave);
  for (long ch = c_lo + wave; ch < c_hi; ch += THREADS / 64) {
    float pf[24];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      pf[4 * q] = np[q].x; pf[4 * q + 1] = np[q].y; pf[4 * q + 2] = np[q].z; pf[4 * q + 3] = np[q].w;
    }
    const uint32_t gw0[4] = {ng0.x, ng0.y, ng0.z, ng0.w}, gw1[4] = {ng1.x, ng1.y, ng1.z, ng1.w};
    const long i0 = (ch * 64 + lane) * NS_RL_K;
    fetch(ch + THREADS / 64);
    uint32_t cur = 0xffffffffu;          // dense index of the current cell's (0,0,0) corner
    unsigned long long acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = 0ull;
    auto flush = [&]() {
      if (cur == 0xffffffffu) return;
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        uint32_t idx = cur + (corner & 1) + ((corner >> 1) & 1) * res + (corner >> 2) * r2;
        idx = idx >= hs ? idx - hs : idx;
        const uint32_t rel = idx - lo;
        if (rel < cnt && acc[corner] != 0ull) atomicAdd(&tab[rel], acc[corner]);
        acc[corner] = 0ull;
      }
    };
#pragma unroll
    for (int j = 0; j < NS_RL_K; j++) {
      const uint32_t w0 = gw0[j >> 1], w1 = gw1[j >> 1];
      const float d0 = (float)__builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (w0 >> 16) : (w0 & 0xffffu)));
      const float d1 = (float)__builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (w1 >> 16) : (w1 & 0xffffu)));
      if ((d0 == 0.0f && d1 == 0.0f) || i0 + j >= nvalid) continue;
      float w[3];
      uint32_t c[3];
#pragma unroll
      for (int dd = 0; dd < 3; dd++) {
        const float p = fmaf(scale, pf[3 * j + dd], 0.5f);
        const float fl = floorf(p);
        c[dd] = min((uint32_t)(int)fl, res - 1u);   // (a position outside the unit cube must not index outside the table)
        w[dd] = p - fl;
      }
      const uint32_t base = c[0] + c[1] * res + c[2] * r2;
      // any corner inside this slice?  (corner indices lie in [base, base + span], wrapped once at hs)
      const bool touches = (base + span >= lo && base < lo + cnt) || (base + span >= hs && base + span - hs >= lo) ||
                           (base + span >= hs && lo == 0u);
      if (!touches) continue;
      if (base != cur) {
        flush();
        cur = base;
      }
      const float wx[2] = {1.0f - w[0], w[0]}, wy[2] = {1.0f - w[1], w[1]}, wz[2] = {1.0f - w[2], w[2]};
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        const float wt = wx[corner & 1] * wy[(corner >> 1) & 1] * wz[corner >> 2];
        acc[corner] += pack_fixed(wt * d0, wt * d1, fixed_scale);
      }
    }
    flush();
  }
  __syncthreads();
  unsigned long long* __restrict__ g64 = reinterpret_cast<unsigned long long*>(grad) + g.offset[l] + lo;
  if (partial != nullptr && nparts > 1) {
    unsigned long long* __restrict__ dst = partial + plan.plane_base[l] + (long)part * hs + lo;
    for (uint32_t e = tid; e < cnt; e += THREADS) dst[e] = tab[e];
    return;
  }
  for (uint32_t e = tid; e < cnt; e += THREADS) {
    const unsigned long long word = tab[e];
    if (word == 0ull) continue;
    if (nparts == 1) g64[e] += word; else atomicAdd(&g64[e], word);
  }
}

// Adam state handed to the fused flushes (master == nullptr: gradient only)
struct AdamFuse {
  float* master;
  _Float16* hp;
  float* m1;
  float* m2;
  float c1, c2, lr, beta1, beta2, eps, inv_grad_scale, inv_fixed_scale;
  const int* ctl;
  int es;   // floats from one table entry to the next in master / m1 / m2: 2 = three dense arrays, 8 / 6 = one 32- / 24-byte record per entry
  // replicated trainers (master == nullptr): the flush APPENDS the touched entries, (entry, packed sum), to this list instead
  // of updating or adding to a dense buffer -- what the trainers exchange (ns_ngp_encode_backward_fused_emit_n)
  ulonglong2* emit_list;
  int* emit_count;
};

// Layout of the optimiser state.  Three dense arrays, or ONE record per table entry, [master.xy | m1.xy | m2.xy (| 8 B unused)]:
// a sparsely touched entry then costs one 128-byte line instead of three (sixteen entries per line in each of three arrays, of
// which a step touched 13 % on a fine level at the 2^18 resolution: 89 % of all lines against 43 %).  Round 6: the `_rec` entry
// points are TOLD the record size (not inferred) -- 8 floats (round 5's 32-byte record) or 6: at the 2^22 resolution
// a step touches 45 % of a fine level's entries, nearly every line of the state, and the 8 unused bytes were a quarter of the
// flush's traffic.  record_floats == 0 (the older entry points): told from the pointers as before -- m1 == master + 2 and
// m2 == master + 4 means 32-byte records, anything else three arrays.  A record's fields are read as 16 + 8 bytes (8 floats:
// base 16-byte aligned) or 8 + 8 + 8 (6 floats: 8-byte aligned).  -1: invalid.
static inline int adam_entry_stride(const float* master, const float* m1, const float* m2, int record_floats = 0) {
  const bool inter = (m1 == master + 2 && m2 == master + 4);
  if (record_floats == 0) return inter ? 8 : 2;
  if (record_floats == 2) return inter ? -1 : 2;
  if ((record_floats != 6 && record_floats != 8) || !inter) return -1;
  if (((uintptr_t)master & (record_floats == 8 ? 15 : 7)) != 0) return -1;
  return record_floats;
}

// Adam on the two parameters of table entry `entry` from the packed fixed-point sum `word` (!= 0); parameters whose own
// gradient is zero are skipped, as ngp_adam_kernel does with l2 = 0
__device__ __forceinline__ void adam_entry(const AdamFuse& ad, long entry, unsigned long long word, float c1, float c2) {
  float g0, g1;
  unpack_fixed(word, ad.inv_fixed_scale, g0, g1);
  g0 *= ad.inv_grad_scale;
  g1 *= ad.inv_grad_scale;
  float2 p, a, b;
  if (ad.es == 6) {       // (uniform) one 24-byte record: three 8-byte fields in, the same out
    float2* __restrict__ rec = reinterpret_cast<float2*>(ad.master + entry * 6);
    p = rec[0], a = rec[1], b = rec[2];
    if (g0 != 0.0f) p.x = adam_apply(p.x, g0, 0.0f, a.x, b.x, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    if (g1 != 0.0f) p.y = adam_apply(p.y, g1, 0.0f, a.y, b.y, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    rec[0] = p;
    rec[1] = a;
    rec[2] = b;
  } else if (ad.es == 8) {       // (uniform) one record: 16 + 8 bytes of one line in, the same out
    float* __restrict__ rec = ad.master + entry * 8;
    const float4 pa = *reinterpret_cast<const float4*>(rec);
    b = *reinterpret_cast<const float2*>(rec + 4);
    p = make_float2(pa.x, pa.y);
    a = make_float2(pa.z, pa.w);
    if (g0 != 0.0f) p.x = adam_apply(p.x, g0, 0.0f, a.x, b.x, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    if (g1 != 0.0f) p.y = adam_apply(p.y, g1, 0.0f, a.y, b.y, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    *reinterpret_cast<float4*>(rec) = make_float4(p.x, p.y, a.x, a.y);
    *reinterpret_cast<float2*>(rec + 4) = b;
  } else {
    float2* __restrict__ mp = reinterpret_cast<float2*>(ad.master) + entry;
    float2* __restrict__ ap = reinterpret_cast<float2*>(ad.m1) + entry;
    float2* __restrict__ bp = reinterpret_cast<float2*>(ad.m2) + entry;
    p = *mp, a = *ap, b = *bp;
    if (g0 != 0.0f) p.x = adam_apply(p.x, g0, 0.0f, a.x, b.x, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    if (g1 != 0.0f) p.y = adam_apply(p.y, g1, 0.0f, a.y, b.y, c1, c2, ad.lr, ad.beta1, ad.beta2, ad.eps);
    *mp = p;
    *ap = a;
    *bp = b;
  }
  h2_t h;
  h[0] = (_Float16)p.x;
  h[1] = (_Float16)p.y;
  reinterpret_cast<h2_t*>(ad.hp)[entry] = h;
}

// sum of the partial planes of the dense levels (entries [0, n_dense) of the grid) into the gradient -- or, with Adam state,
// straight into the parameters (the gradient buffer is then not touched).  `reset` (may be null): counters of the fused
// binned path cleared by this, the last kernel of its launch sequence.
__global__ __launch_bounds__(256) void ngp_enc_dense_reduce_kernel(GridLayout g, EncBwdPlan plan, int n_levels,
                                                                   const unsigned long long* __restrict__ partial, long n_dense,
                                                                   float* __restrict__ grad, AdamFuse ad, int* __restrict__ reset) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (reset != nullptr && e == 0) *reset = 0;
  if (e >= n_dense) return;
  int l = 0;
  while (l + 1 < n_levels && (long)g.offset[l + 1] <= e) l++;
  const long hs = (long)(g.offset[l + 1] - g.offset[l]);
  const unsigned long long* __restrict__ src = partial + plan.plane_base[l] + (e - (long)g.offset[l]);
  unsigned long long sum = 0ull;
  for (int p = 0; p < plan.parts[l]; p++) sum += src[(long)p * hs];
  if (sum == 0ull) return;
  if (ad.master != nullptr) {
    const float c1 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C1]) : ad.c1, c2 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C2]) : ad.c2;
    adam_entry(ad, e, sum, c1, c2);
  } else {
    reinterpret_cast<unsigned long long*>(grad)[e] += sum;
  }
}

// ---------------------------------------------------------------------------------------------
// Encode backward, hashed levels, BINNED (what ns_ngp_encode_backward launches when it is given a workspace).
//
// The owner-computes kernel above re-derives the corner indices of every sample in each of the 32 slice owners of a level:
// ~0.5 ms of integer VALU per step on the twelve hashed levels.  Binning does that arithmetic twice instead of 32 times:
//   count    a workgroup takes a tile of 1024 samples of one level, histograms the 8 x 1024 corner indices over the level's
//            bins (bin = index >> 14 = the 16384-entry table slice), and reserves the tile's range in every bin with ONE
//            returning atomic per bin (tile offsets in arrival order: the sums below are integer, hence order independent);
//   scatter  the same tile again: every (sample, corner) becomes a 64-bit record [14-bit index in the slice | two 25-bit
//            Q18 gradient fields], ranked inside its bin through LDS, staged bin-sorted in LDS and written out as contiguous
//            runs (whole 128-byte lines);
//   accum    one workgroup per (level, bin) streams its records, accumulates them in its 128-KB LDS slice (ds_add_u64) and
//            flushes the non-zero entries with plain read-modify-writes (it is the only writer of its slice).
// Traffic: 8 B written + 8 B read per (sample, corner) on the hashed levels (~0.4 GB per 2^18 samples), all of it
// coalesced streams; no global atomics on table entries; bit-identical to the atomic and the owner-computes kernels
// (same per-contribution rounding; the 25-bit fields saturate at |g| >= 64 gradient units, loss_scale included).
// ---------------------------------------------------------------------------------------------
#define NS_BIN_TILE 1024
#define NS_BIN_MAX 32
struct BinPlan {
  int nh;            // hashed levels
  int level[16];     // k -> level
  int nbins[16];     // k -> bins of the level (table entries / 16384, at least 1)
  int ntiles;        // ceil(N / 1024)
};

static void bin_plan_host(const GridLayout& g, int n_levels, long N, BinPlan& b) {
  b.nh = 0;
  for (int l = 0; l < n_levels; l++) {
    if (!level_is_hashed(g, l)) continue;
    const uint32_t hs = g.offset[l + 1] - g.offset[l];
    b.level[b.nh] = l;
    b.nbins[b.nh] = (int)((hs + NS_ENC_SLICE - 1) / NS_ENC_SLICE);
    b.nh++;
  }
  for (int k = b.nh; k < 16; k++) b.level[k] = b.nbins[k] = 0;
  b.ntiles = (int)((N + NS_BIN_TILE - 1) / NS_BIN_TILE);
}

static bool bin_plan_ok(const BinPlan& b) {
  if (b.nh == 0) return false;
  for (int k = 0; k < b.nh; k++)
    if (b.nbins[k] > NS_BIN_MAX) return false;
  return true;
}

// workspace layout (bytes): [tot: nh*32 int32 (kept zero between calls)] [cnt: nh*ntiles*32 int2 {count, offset}] [queue]
static size_t bin_ws_tot_bytes(const BinPlan& b) { return ((size_t)b.nh * NS_BIN_MAX * 4 + 255) / 256 * 256; }
static size_t bin_ws_cnt_bytes(const BinPlan& b) { return ((size_t)b.nh * b.ntiles * NS_BIN_MAX * 8 + 255) / 256 * 256; }
static size_t bin_ws_queue_bytes(const BinPlan& b) { return (size_t)b.nh * 8 * (size_t)(b.ntiles * (long)NS_BIN_TILE) * 8; }
// dense levels = the leading (coarsest) levels of the grid: entries [0, n_dense); one partial plane per sample part
static long dense_prefix_entries(const GridLayout& g, int n_levels) {
  int l = 0;
  while (l < n_levels && !level_is_hashed(g, l)) l++;
  for (int m = l; m < n_levels; m++)
    if (!level_is_hashed(g, m)) return -1;   // a dense level above a hashed one: not a prefix (not produced by grid_layout_host)
  return (long)g.offset[l];
}
static size_t bin_ws_bytes(const BinPlan& b, const GridLayout& g, int n_levels) {
  const long nd = dense_prefix_entries(g, n_levels);
  return bin_ws_tot_bytes(b) + bin_ws_cnt_bytes(b) + bin_ws_queue_bytes(b) +
         (nd > 0 ? (size_t)NS_ENC_PARTS_COARSE * (size_t)nd * 8 : 0);   // upper bound of the per-level plane sets
}

struct BinSample {
  bool valid;
  uint32_t idx[8];
  float wx[2], wy[2], wz[2], d0, d1;
};

struct BinRaw {
  float d0, d1, p[3];
};

// loads only (issued for all of a thread's samples before any of them is processed)
__device__ __forceinline__ BinRaw bin_load(int l, const float* __restrict__ pos, const h2_t* __restrict__ dLdout, long i, long N,
                                           int L, int unit_major, long cnt) {
  BinRaw r;
  r.d0 = r.d1 = 0.0f;
  r.p[0] = r.p[1] = r.p[2] = 0.0f;
  if (i < cnt) {
    if (unit_major) {
      const _Float16* dp = reinterpret_cast<const _Float16*>(dLdout);
      r.d0 = (float)dp[(long)(2 * l) * N + i];
      r.d1 = (float)dp[(long)(2 * l + 1) * N + i];
    } else {
      const h2_t d = dLdout[i * L + l];
      r.d0 = (float)d[0];
      r.d1 = (float)d[1];
    }
    r.p[0] = pos[i * 3];
    r.p[1] = pos[i * 3 + 1];
    r.p[2] = pos[i * 3 + 2];
  }
  return r;
}

__device__ __forceinline__ BinSample bin_sample(const GridLayout& g, int l, uint32_t hs, const BinRaw& r) {
  BinSample s;
  s.d0 = r.d0;
  s.d1 = r.d1;
  s.valid = s.d0 != 0.0f || s.d1 != 0.0f;
  if (!s.valid) return s;
  const float scale = g.scale[l];
  uint32_t c[3];
  float w[3];
#pragma unroll
  for (int dd = 0; dd < 3; dd++) {
    const float p = fmaf(scale, r.p[dd], 0.5f);
    const float fl = floorf(p);
    c[dd] = (uint32_t)(int)fl;
    w[dd] = p - fl;
  }
  s.wx[0] = 1.0f - w[0]; s.wx[1] = w[0];
  s.wy[0] = 1.0f - w[1]; s.wy[1] = w[1];
  s.wz[0] = 1.0f - w[2]; s.wz[1] = w[2];
  const uint32_t hx[2] = {c[0], c[0] + 1u};
  const uint32_t hy0 = c[1] * 2654435761u, hz0 = c[2] * 805459861u;
  const uint32_t hy[2] = {hy0, hy0 + 2654435761u};
  const uint32_t hz[2] = {hz0, hz0 + 805459861u};
#pragma unroll
  for (int corner = 0; corner < 8; corner++)
    s.idx[corner] = (hx[corner & 1] ^ hy[(corner >> 1) & 1] ^ hz[corner >> 2]) & (hs - 1u);   // hashed: hs = 2^log2_hashmap
  return s;
}

__global__ void ngp_zero_ints_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void ngp_enc_bin_count_kernel(GridLayout g, BinPlan bp, const float* __restrict__ pos,
                                                                const h2_t* __restrict__ dLdout, long N, int L, int unit_major,
                                                                int* __restrict__ tot, int2* __restrict__ cnt,
                                                                const int* __restrict__ n_dev) {
  __shared__ int hist[NS_BIN_MAX];
  const int k = blockIdx.y, l = bp.level[k], tile = blockIdx.x, tid = threadIdx.x;
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  if (tid < NS_BIN_MAX) hist[tid] = 0;
  __syncthreads();
  BinRaw raw[NS_BIN_TILE / 256];
#pragma unroll
  for (int j = 0; j < NS_BIN_TILE / 256; j++)
    raw[j] = bin_load(l, pos, dLdout, (long)tile * NS_BIN_TILE + j * 256 + tid, N, L, unit_major, nvalid);
#pragma unroll
  for (int j = 0; j < NS_BIN_TILE / 256; j++) {
    const BinSample s = bin_sample(g, l, hs, raw[j]);
    if (s.valid) {
#pragma unroll
      for (int corner = 0; corner < 8; corner++) atomicAdd(&hist[s.idx[corner] >> 14], 1);
    }
  }
  __syncthreads();
  if (tid < NS_BIN_MAX) {
    const int c = hist[tid];
    const int off = c > 0 ? atomicAdd(&tot[k * NS_BIN_MAX + tid], c) : 0;   // the tile's range in the bin (arrival order)
    cnt[((long)k * bp.ntiles + tile) * NS_BIN_MAX + tid] = make_int2(c, off);
  }
}

__device__ __forceinline__ unsigned long long bin_record(uint32_t rel, float g0, float g1, float S) {
  const float lim = 16777215.0f;   // 25-bit signed fields
  const int a = (int)__float2int_rn(fminf(fmaxf(g0 * S, -lim), lim));
  const int b = (int)__float2int_rn(fminf(fmaxf(g1 * S, -lim), lim));
  return ((unsigned long long)rel << 50) | ((unsigned long long)((uint32_t)b & 0x1ffffffu) << 25) |
         (unsigned long long)((uint32_t)a & 0x1ffffffu);
}

__global__ __launch_bounds__(256) void ngp_enc_bin_scatter_kernel(GridLayout g, BinPlan bp, const float* __restrict__ pos,
                                                                  const h2_t* __restrict__ dLdout, long N, int L, int unit_major,
                                                                  float fixed_scale, const int* __restrict__ tot,
                                                                  const int2* __restrict__ cnt,
                                                                  unsigned long long* __restrict__ queue,
                                                                  const int* __restrict__ n_dev) {
  __shared__ unsigned long long rec[8 * NS_BIN_TILE];   // 64 KB: the tile's records, bin-sorted
  __shared__ int lbase[NS_BIN_MAX + 1], lcnt[NS_BIN_MAX], gdst[NS_BIN_MAX];
  const int k = blockIdx.y, l = bp.level[k], tile = blockIdx.x, tid = threadIdx.x;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  if (tid < 64) {  // one wave: prefix sums over the 32 bins (tile counts -> LDS bases, bin totals -> queue bases)
    const int b = tid & 31;
    const int2 co = cnt[((long)k * bp.ntiles + tile) * NS_BIN_MAX + b];
    int c = co.x, t = tot[k * NS_BIN_MAX + b];
    int ci = c, ti = t;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int uc = __shfl_up(ci, d, 32), ut = __shfl_up(ti, d, 32);
      if (b >= d) { ci += uc; ti += ut; }
    }
    if (tid < 32) {
      lbase[b] = ci - c;
      if (b == 31) lbase[32] = ci;
      lcnt[b] = 0;
      gdst[b] = (ti - t) + co.y;
    }
  }
  BinRaw raw[NS_BIN_TILE / 256];   // (loaded before the barrier: in flight while the prefix sums are formed)
#pragma unroll
  for (int j = 0; j < NS_BIN_TILE / 256; j++)
    raw[j] = bin_load(l, pos, dLdout, (long)tile * NS_BIN_TILE + j * 256 + tid, N, L, unit_major, nvalid);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NS_BIN_TILE / 256; j++) {
    const BinSample s = bin_sample(g, l, hs, raw[j]);
    if (s.valid) {
      int slot[8];   // the eight returning LDS atomics and base reads are issued back to back; the records follow
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        const int b = (int)(s.idx[corner] >> 14);
        slot[corner] = atomicAdd(&lcnt[b], 1) + lbase[b];
      }
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        const float wt = s.wx[corner & 1] * s.wy[(corner >> 1) & 1] * s.wz[corner >> 2];
        rec[slot[corner]] = bin_record(s.idx[corner] & 16383u, wt * s.d0, wt * s.d1, fixed_scale);
      }
    }
  }
  __syncthreads();
  unsigned long long* __restrict__ q = queue + (long)k * 8 * ((long)bp.ntiles * NS_BIN_TILE);
  // copy-out: a wave takes every fourth bin and streams its staged run (consecutive lanes -> consecutive records)
  const int wave = tid >> 6, lane = tid & 63;
  for (int b = wave; b < NS_BIN_MAX; b += 4) {
    const int b0 = lbase[b], n = lbase[b + 1] - b0;
    unsigned long long* __restrict__ dst = q + gdst[b];
    for (int e = lane; e < n; e += 64) dst[e] = rec[b0 + e];
  }
}

__global__ __launch_bounds__(1024) void ngp_enc_bin_accum_kernel(GridLayout g, BinPlan bp, const int* __restrict__ tot,
                                                                 const unsigned long long* __restrict__ queue,
                                                                 float* __restrict__ grad) {
  __shared__ unsigned long long tab[NS_ENC_SLICE];
  __shared__ int s_base, s_tot;
  const int k = blockIdx.y, l = bp.level[k], b = blockIdx.x, tid = threadIdx.x;
  if (b >= bp.nbins[k]) return;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const uint32_t lo = (uint32_t)b * NS_ENC_SLICE;
  const uint32_t n_e = min((uint32_t)NS_ENC_SLICE, hs - lo);
  for (uint32_t e = tid; e < NS_ENC_SLICE; e += 1024) tab[e] = 0ull;
  if (tid < 64) {
    const int bb = tid & 31;
    const int t = tot[k * NS_BIN_MAX + bb];
    int ti = t;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int ut = __shfl_up(ti, d, 32);
      if (bb >= d) ti += ut;
    }
    if (tid == b) {
      s_base = ti - t;
      s_tot = t;
    }
  }
  __syncthreads();
  const unsigned long long* __restrict__ q = queue + (long)k * 8 * ((long)bp.ntiles * NS_BIN_TILE) + s_base;
  const int n = s_tot;
  for (int e0 = 0; e0 < n; e0 += 4 * 1024) {
    unsigned long long r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = e0 + u * 1024 + tid;
      r[u] = e < n ? q[e] : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (e0 + u * 1024 + tid < n) {
        const uint32_t rel = (uint32_t)(r[u] >> 50);
        const long long a = ((long long)(r[u] << 39)) >> 39;          // sign-extended 25-bit fields
        const long long c = ((long long)(r[u] << 14)) >> 39;
        atomicAdd(&tab[rel], (unsigned long long)(a + (c << 32)));
      }
    }
  }
  __syncthreads();
  unsigned long long* __restrict__ g64 = reinterpret_cast<unsigned long long*>(grad) + g.offset[l] + lo;
  for (uint32_t e = tid; e < n_e; e += 1024) {
    const unsigned long long word = tab[e];
    if (word != 0ull) g64[e] += word;
  }
}

// ---------------------------------------------------------------------------------------------
// Binned path, round 3: NO count pass, NO global atomics at all, runs merged before they are written, finer bins, Adam fused
// into the flush.
//
// The count kernel existed to give every (tile, bin) run its place in a densely packed queue, and paid for it with a pass
// over the samples and a returning global atomic per (tile, bin).  Here every (level, bin, tile) owns a FIXED slot of
// NS_FB_SLOT records (4 x the mean load) and the scatter writes the number of records it put there next to it: no
// reservation, no counters to keep zeroed, no dependency on another workgroup anywhere in the kernel.  A run that outgrows its
// slot spills into an overflow list sized for the worst case (one atomic per spilling run; every accumulate workgroup scans
// the list, which is empty unless more than 1/16 of a tile's records meet in one of the 64 bins), so no record is ever lost.
// A lane owns FOUR CONSECUTIVE samples -- consecutive steps of one ray -- and sums the contributions of neighbours that fall
// into the same cell of the level before it emits records (exact: the per-contribution roundings are integers already):
// on the coarser hashed levels a cell holds several steps of a ray, which removes a quarter of the record stream on the
// trainer's rays; the positions arrive as three 16-byte loads, the gradients as 8-byte ones.
// Bins are 8192-entry slices (64 KB of LDS accumulators, two 512-lane workgroups per CU): 768 work items of half the length
// balance better over 256 CUs than 384, and one workgroup's flush (an HBM stream) overlaps its neighbour's record loop.
// The flush applies Adam to the touched entries in place (`AdamFuse`): the sums sit in LDS, the entry's master / moments are
// read once and written once -- the round-2 sequence wrote the sum to the gradient buffer (16 B per entry), and a separate
// pass streamed the whole table (68 B per entry, touched or not: 105 us with real gradients).  Semantics unchanged:
// zero-gradient parameters are skipped (tiny-cuda-nn), arithmetic shared with ngp_adam_kernel through adam_apply().
// ---------------------------------------------------------------------------------------------
// (round 4, measured and not taken: 4096-entry bins -- 128 per level, 32 KB of LDS, four accumulate workgroups per CU -- scatter
//  78 -> 100 us, accumulate 126 -> 149 us, training step 0.287 -> 0.309 ms; four entries per lane and round in the Adam flush with
//  their state loaded up front: 126 -> 133 us, the flush is not latency-bound)
#define NS_FB_SLICE 8192
#define NS_FB_SHIFT 13
#define NS_FB_BINS 64
#define NS_FB_SLOT 512
#define NS_FB_RUN 4      // consecutive samples per lane
struct FusedPlan {
  int nh;            // binned levels: the hashed ones and (dense_too) the dense levels of more than one 16384-entry slice
  int level[16];     // k -> level
  int nbins[16];     // k -> bins of the level (<= 64)
  int slot[16];      // k -> records per (bin, tile) slot: the level's 64 x 512 records per tile, divided among its bins
  int hashed[16];    // k -> hash or dense index
  int merge[16];     // k -> merge the single-cell runs of neighbouring lanes before records are emitted (fb_wave_merge)
  int shift[16];     // k -> log2 of the level's bin size (13 = 8192 entries; small dense levels: smaller bins, >= 8 of them)
  int ntiles;        // ceil(N / 1024)
  long ovf_cap;      // entries of the overflow list
};

// Dense levels of ONE slice keep the owner-computes kernel (a workgroup holds the whole level in LDS); with dense_too the larger
// dense levels go through the bins like the hashed ones: their owner-computes tasks scan every sample once per slice (22 x N
// sample visits for the default grid's five dense levels) in whole-CU workgroups of 128 KB of LDS that nothing runs next to.
// dense_mode 0: every dense level owner-computes; 1: the dense levels of more than one 16384-entry slice through the bins (round
// 3's A/B form); 2 (round 4, default): EVERY dense level through the bins -- the scatter merges the runs of neighbouring lanes
// first (fb_wave_merge), which is what keeps a coarse level's few bins from drowning in records -- and no owner-computes
// workgroup, no partial planes, no reduce pass and no third stream are left in the step.
static int fused_dense_levels(const GridLayout& g, int n_levels, int dense_mode) {
  if (dense_mode >= 2) return 0;
  int l = 0;
  while (l < n_levels && !level_is_hashed(g, l) && (!dense_mode || g.offset[l + 1] - g.offset[l] <= (uint32_t)NS_ENC_SLICE)) l++;
  return l;
}
static int fused_dense_mode() {
  static const int mode = [] {
    const char* e = ns_variant_env("NS_ENC_DENSE_BINNED");
    return e != nullptr && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 2;
  }();
  return mode;
}
// wave-level merge of single-cell lanes (fb_wave_merge) on levels up to this resolution (dense levels always; NS_FB_MERGE_RES)
static int fused_merge_res() {
  static const int r = [] { const char* e = ns_variant_env("NS_FB_MERGE_RES"); return e ? atoi(e) : 0; }();
  return r;
}

static bool fused_plan_host(const GridLayout& g, int n_levels, long N, FusedPlan& f, int dense_too = 0) {
  f.nh = 0;
  for (int l = fused_dense_levels(g, n_levels, dense_too); l < n_levels; l++) {
    const bool hashed = level_is_hashed(g, l);
    if (!hashed && !dense_too) continue;
    const uint32_t hs = g.offset[l + 1] - g.offset[l];
    // a level of few 8192-entry bins (the coarse dense ones) gets smaller bins until there are at least 8: their records --
    // one set per (wave, cell) after the merge, still several times a hashed bin's load on the coarsest level -- then spread
    // over 8+ work items of the accumulate pass instead of 1 / 2 / 7
    int sh = NS_FB_SHIFT;
    if (!hashed && dense_too >= 2)
      while (sh > 9 && (hs + (1u << sh) - 1) >> sh < 8u) sh--;
    const int nb = (int)((hs + (1u << sh) - 1) >> sh);
    if (nb > NS_FB_BINS) return false;
    f.shift[f.nh] = sh;
    f.level[f.nh] = l;
    f.nbins[f.nh] = nb;
    f.hashed[f.nh] = hashed ? 1 : 0;
    f.merge[f.nh] = (!hashed && dense_too >= 2) || g.res[l] <= fused_merge_res() ? 1 : 0;
    const int sl = (NS_FB_BINS * NS_FB_SLOT / nb) & ~3;
    f.slot[f.nh] = sl < 8 * NS_BIN_TILE ? sl : 8 * NS_BIN_TILE;      // (a tile has at most 8 x 1024 records)
    f.nh++;
  }
  // FINEST level first (round 5).  Both passes launch grid.y = k, so the order of the plan is the order in which the levels'
  // workgroups are dispatched; the workspace is addressed by k in both and any order gives the same sums.  With the coarse
  // levels' few, merged records last the tail of each pass is light: mapping leg 4.92 -> 4.84 ms per frame in 3 of 3 paired
  // bench runs (profiles/r05_ab_records.json); the step alone does not move.  NS_VARIANTS=1 NS_FB_LEVEL_ORDER=fwd: coarsest first.
  static const bool rev = [] { const char* e = ns_variant_env("NS_FB_LEVEL_ORDER"); return !(e != nullptr && e[0] == 'f'); }();
  if (rev)
    for (int a = 0, b = f.nh - 1; a < b; a++, b--) {
      std::swap(f.shift[a], f.shift[b]);
      std::swap(f.level[a], f.level[b]);
      std::swap(f.nbins[a], f.nbins[b]);
      std::swap(f.hashed[a], f.hashed[b]);
      std::swap(f.merge[a], f.merge[b]);
      std::swap(f.slot[a], f.slot[b]);
    }
  for (int k = f.nh; k < 16; k++) f.level[k] = f.nbins[k] = f.slot[k] = f.hashed[k] = f.merge[k] = f.shift[k] = 0;
  f.ntiles = (int)((N + NS_BIN_TILE - 1) / NS_BIN_TILE);
  f.ovf_cap = (long)f.ntiles * NS_BIN_TILE * 8 * (f.nh > 0 ? f.nh : 1);
  // the scatter addresses a record as __umul24(bin, ntiles * slot[k]) + rank: the stride of the LARGEST slot of the plan has to
  // stay below 2^24 (a level of few bins has slots of up to 8 x NS_BIN_TILE records, not NS_FB_SLOT: N < 2.1 M samples per call
  // with an 8192-record slot, 4.2 M with the dense levels' 4096)
  int max_slot = NS_FB_SLOT;
  for (int k = 0; k < f.nh; k++) max_slot = f.slot[k] > max_slot ? f.slot[k] : max_slot;
  if ((long)f.ntiles * max_slot >= (1L << 24)) return false;
  return f.nh > 0;
}
// workspace: [ctr: {overflow count, error flag}] [cnt: nh*64*ntiles run lengths] [queue: nh*64*ntiles slots of NS_FB_SLOT
// records] [overflow list] [partial planes of the dense levels]
static size_t fused_ws_ctr_bytes(const FusedPlan&) { return 256; }
static size_t fused_ws_cnt_bytes(const FusedPlan& f) { return ((size_t)f.nh * NS_FB_BINS * f.ntiles * 4 + 255) / 256 * 256; }
static size_t fused_ws_queue_bytes(const FusedPlan& f) { return (size_t)f.nh * NS_FB_BINS * (size_t)f.ntiles * NS_FB_SLOT * 8; }
static size_t fused_ws_ovf_bytes(const FusedPlan& f) { return (size_t)f.ovf_cap * 16; }
static size_t fused_ws_bytes(const FusedPlan& f, const GridLayout& g, int n_levels) {
  const long nd = dense_prefix_entries(g, n_levels);
  return fused_ws_ctr_bytes(f) + fused_ws_cnt_bytes(f) + fused_ws_queue_bytes(f) + fused_ws_ovf_bytes(f) +
         (nd > 0 ? (size_t)NS_ENC_PARTS_COARSE * (size_t)nd * 8 : 0);
}

// one merged run of a lane: the cell and the summed integer contributions of its 8 corners
struct FbRun {
  uint32_t cx, cy, cz;   // (three scalars, not c[3]: as an array member the cell was the one part of a run the compiler kept in
                         //  SCRATCH memory -- 64 bytes per lane, 33 scratch stores and 64 scratch loads in the scatter kernel)
  int a[8], b[8];
};

// (round 5: 24-bit multiplies.  v_mul_lo_u32 issues at a quarter of the rate of v_mul_u32_u24, and the scatter pass is vector-ALU
//  bound.  Hashed level: only the low log2(hs) <= 24 bits of y P1 and z P2 survive the mask, and those are the low
//  bits of y (P1 mod 2^24) -- exact for cell coordinates below 2^24.  Dense level: res^3 <= hs <= 2^24, so every factor of
//  x + (y + z res) res is below 2^24 as well.  Same indices, bit for bit: the bit-identity tests compare with kernels that use
//  grid_index_lvl.)
__device__ __forceinline__ void fb_indices(const uint32_t c[3], uint32_t hs, uint32_t idx[8]) {
  const uint32_t hx[2] = {c[0], c[0] + 1u};
  const uint32_t hy0 = (uint32_t)__umul24(c[1], 2654435761u & 0xffffffu), hz0 = (uint32_t)__umul24(c[2], 805459861u & 0xffffffu);
  const uint32_t hy[2] = {hy0, hy0 + (2654435761u & 0xffffffu)};
  const uint32_t hz[2] = {hz0, hz0 + (805459861u & 0xffffffu)};
#pragma unroll
  for (int corner = 0; corner < 8; corner++) idx[corner] = (hx[corner & 1] ^ hy[(corner >> 1) & 1] ^ hz[corner >> 2]) & (hs - 1u);
}

__device__ __forceinline__ void fb_indices_any(bool hashed, const uint32_t c[3], uint32_t hs, uint32_t res, uint32_t idx[8]) {
  if (hashed) {
    fb_indices(c, hs, idx);
  } else {
    // grid_index_lvl(false, ...) of the 8 corners: clamp, x + (y + z res) res, one conditional subtraction
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    const uint32_t x[2] = {mn(c[0], res), mn(c[0] + 1u, res)};
    const uint32_t y[2] = {mn(c[1], res), mn(c[1] + 1u, res)};
    const uint32_t zr[2] = {(uint32_t)__umul24(mn(c[2], res), res), (uint32_t)__umul24(mn(c[2] + 1u, res), res)};
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const uint32_t i = x[corner & 1] + (uint32_t)__umul24(y[(corner >> 1) & 1] + zr[corner >> 2], res);
      idx[corner] = i >= hs ? i - hs : i;
    }
  }
}

// Merge across the lanes of a wave.  Lane i holds samples [4 i, 4 i + 4) of the sample array: consecutive steps of a ray.  On a
// coarse level a cell holds tens of steps, so whole sequences of lanes sit in ONE cell.  Every lane has a FIRST run (which may
// continue the cell the previous lane ended in) and a LAST run (which the next lane may continue); for a lane of one run they
// are the same run and the chain passes through it.  A segmented inclusive scan over the last runs (segment = a maximal
// chain) leaves the chain's total in the lane that ends it -- in its only run, or in the FIRST run of a lane that goes on to
// other cells -- and the lanes before it drop the run they passed on: one record set per (wave, chain) instead of one per
// (lane, run).  Exact: integer sums of the same per-contribution roundings (any grouping gives the same table sums), as long as
// the record's 25-bit fields hold the total -- only runs below 2^17 per field take part (64 of them stay below 2^23).
// 16 sums x 7 shuffle rounds per wave.  The default grid's coarsest level: ~2.4 -> ~0.5 records per sample.
__device__ __forceinline__ void fb_wave_merge(FbRun (&run)[NS_FB_RUN], int& nrun) {
  const int lane = threadIdx.x & 63;
  // the last run (run[nrun - 1], static indexing only) in registers of its own
  // (member by member: an aggregate copy of an FbRun is what made the compiler keep the runs' cells in scratch memory)
  FbRun v;
  v.cx = run[0].cx; v.cy = run[0].cy; v.cz = run[0].cz;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    v.a[corner] = run[0].a[corner];
    v.b[corner] = run[0].b[corner];
  }
#pragma unroll
  for (int r = 1; r < NS_FB_RUN; r++)
    if (r == nrun - 1) {
      v.cx = run[r].cx; v.cy = run[r].cy; v.cz = run[r].cz;
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        v.a[corner] = run[r].a[corner];
        v.b[corner] = run[r].b[corner];
      }
    }
  int big_f = 0, big_l = 0;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    big_f |= abs(run[0].a[corner]) | abs(run[0].b[corner]);
    big_l |= abs(v.a[corner]) | abs(v.b[corner]);
  }
  const bool can_f = nrun >= 1 && big_f < (1 << 17), can_l = nrun >= 1 && big_l < (1 << 17);
  const uint32_t p0 = __shfl_up(v.cx, 1, 64), p1 = __shfl_up(v.cy, 1, 64), p2 = __shfl_up(v.cz, 1, 64);
  const int pcan = __shfl_up((int)can_l, 1, 64);
  // recv: this lane's first run continues the previous lane's last run; cont: and it is this lane's only run
  const bool recv = lane > 0 && can_f && pcan && p0 == run[0].cx && p1 == run[0].cy && p2 == run[0].cz;
  const bool cont = recv && nrun == 1;
  int head = cont ? 0 : lane;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(head, d, 64);
    if (lane >= d) head = max(head, o);
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const bool take = lane - d >= head;
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const int oa = __shfl_up(v.a[corner], d, 64), ob = __shfl_up(v.b[corner], d, 64);
      if (take) {
        v.a[corner] += oa;
        v.b[corner] += ob;
      }
    }
  }
  // a lane of several runs takes the chain's total into its first run; a lane of one run has it in v already
  const bool recv_multi = recv && nrun > 1;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const int oa = __shfl_up(v.a[corner], 1, 64), ob = __shfl_up(v.b[corner], 1, 64);
    if (recv_multi) {
      run[0].a[corner] += oa;
      run[0].b[corner] += ob;
    } else if (nrun == 1) {
      run[0].a[corner] = v.a[corner];
      run[0].b[corner] = v.b[corner];
    }
  }
  const int nrecv = __shfl_down((int)recv, 1, 64);
  if (lane < 63 && nrecv) nrun--;           // the next lane carries this lane's last run on
}

// The inputs of one lane: its FOUR CONSECUTIVE samples of tile `tile` -- positions (level independent) and the two upstream
// gradient values of level l.
struct FbPos {
  float px[NS_FB_RUN][3];
};
typedef _Float16 fb_h4_t __attribute__((ext_vector_type(4)));
struct FbGrad {            // (halves as loaded: a workgroup holds the rows of several levels)
  fb_h4_t ga, gb;
};
__device__ __forceinline__ void fb_load_pos(int tile, int tid, const float* __restrict__ pos, long nvalid, int vec, FbPos& P) {
  const long i0 = (long)tile * NS_BIN_TILE + (long)tid * NS_FB_RUN;
  if (vec && i0 + NS_FB_RUN <= nvalid) {
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(pos + i0 * 3);
    const float4 A = p4[0], B = p4[1], Cc = p4[2];
    P.px[0][0] = A.x; P.px[0][1] = A.y; P.px[0][2] = A.z;
    P.px[1][0] = A.w; P.px[1][1] = B.x; P.px[1][2] = B.y;
    P.px[2][0] = B.z; P.px[2][1] = B.w; P.px[2][2] = Cc.x;
    P.px[3][0] = Cc.y; P.px[3][1] = Cc.z; P.px[3][2] = Cc.w;
  } else {
#pragma unroll
    for (int j = 0; j < NS_FB_RUN; j++) {
      const long i = i0 + j;
      const bool ok = i < nvalid;
      P.px[j][0] = ok ? pos[i * 3] : 0.0f;
      P.px[j][1] = ok ? pos[i * 3 + 1] : 0.0f;
      P.px[j][2] = ok ? pos[i * 3 + 2] : 0.0f;
    }
  }
}
__device__ __forceinline__ void fb_load_grad(int l, int tile, int tid, const _Float16* __restrict__ dpu, long N, long nvalid, int vec,
                                             FbGrad& G) {
  const long i0 = (long)tile * NS_BIN_TILE + (long)tid * NS_FB_RUN;
  const _Float16* __restrict__ g0p = dpu + (long)(2 * l) * N;
  const _Float16* __restrict__ g1p = g0p + N;
  if (vec && i0 + NS_FB_RUN <= nvalid) {
    G.ga = *reinterpret_cast<const fb_h4_t*>(g0p + i0);
    G.gb = *reinterpret_cast<const fb_h4_t*>(g1p + i0);
  } else {
#pragma unroll
    for (int j = 0; j < NS_FB_RUN; j++) {
      const long i = i0 + j;
      const bool ok = i < nvalid;
      G.ga[j] = ok ? g0p[i] : (_Float16)0;
      G.gb[j] = ok ? g1p[i] : (_Float16)0;
    }
  }
}

// The runs of one lane on level l from its loaded inputs: neighbours that share a cell summed (exact).
__device__ __forceinline__ void fb_runs_of(const GridLayout& g, int l, const FbPos& P, const FbGrad& G, float fixed_scale,
                                           FbRun (&run)[NS_FB_RUN], int& nrun_out) {
  const float (&px)[NS_FB_RUN][3] = P.px;
  float d0[NS_FB_RUN], d1[NS_FB_RUN];
#pragma unroll
  for (int j = 0; j < NS_FB_RUN; j++) {
    d0[j] = (float)G.ga[j];
    d1[j] = (float)G.gb[j];
  }
  // ---- merge neighbours that share a cell ----
#pragma unroll
  for (int corner = 0; corner < 8; corner++) run[0].a[corner] = run[0].b[corner] = 0;   // (read by fb_wave_merge's shuffles)
  run[0].cx = run[0].cy = run[0].cz = 0u;
  int nrun = 0;
  bool open = false, open_small = false;
  const float scale = g.scale[l];
  const float lim = 16777215.0f;   // 25-bit signed record fields
#pragma unroll
  for (int j = 0; j < NS_FB_RUN; j++) {
    // A sample whose upstream gradient is below HALF a fixed-point unit in both features emits nothing: its 8 contributions are
    // fl(fl(wt d) S) with 0 <= wt <= 1, and rounding is monotone, so |fl(wt d)| <= |d| and the product stays below 0.5 -> every
    // one rounds to the integer 0, which is never a record.  Skipped before any weight is formed (exact: same sums, bit for
    // bit).  In a converged scene that is most samples: the mapper's steps inside the pipeline carry a non-zero gradient on
    // 93 % of their samples and emit records for a few per cent of their contributions (zero gradient: the same test).
    if (fabsf(d0[j]) * fixed_scale < 0.5f && fabsf(d1[j]) * fixed_scale < 0.5f) continue;
    uint32_t c[3];
    float w[3];
#pragma unroll
    for (int dd = 0; dd < 3; dd++) {
      const float p = fmaf(scale, px[j][dd], 0.5f);
      const float fl = floorf(p);
      c[dd] = (uint32_t)(int)fl;
      w[dd] = p - fl;
    }
    const float wx[2] = {1.0f - w[0], w[0]}, wy[2] = {1.0f - w[1], w[1]}, wz[2] = {1.0f - w[2], w[2]};
    int ca[8], cb[8];
    float bigf = 0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const float wt = wx[corner & 1] * wy[(corner >> 1) & 1] * wz[corner >> 2];
      // (round 6: the largest magnitude of the 16 fields is taken on the ROUNDED FLOATS -- one v_max3_f32 with |.| source
      //  modifiers per corner -- instead of abs / abs / or / or on the integers: six instructions per corner less in a pass that
      //  is bound by its vector instructions; the rounded values are integers, so the test below is the same test)
      const float fa = rintf(fminf(fmaxf(wt * d0[j] * fixed_scale, -lim), lim)), fb = rintf(fminf(fmaxf(wt * d1[j] * fixed_scale, -lim), lim));
      ca[corner] = (int)fa;
      cb[corner] = (int)fb;
      bigf = fmaxf(fmaxf(bigf, fabsf(fa)), fabsf(fb));
    }
    const bool small = bigf < 2097152.0f;   // (2^21) four such contributions stay inside the 25-bit field
    // (static indexing only: `run` must stay in registers, so the open run is always run[nrun - 1] addressed by unrolled selects)
    bool merged = false;
#pragma unroll
    for (int r = 0; r < NS_FB_RUN; r++) {
      if (r == nrun - 1 && open && open_small && small && run[r].cx == c[0] && run[r].cy == c[1] && run[r].cz == c[2]) {
#pragma unroll
        for (int corner = 0; corner < 8; corner++) {
          run[r].a[corner] += ca[corner];
          run[r].b[corner] += cb[corner];
        }
        merged = true;
      }
    }
    if (!merged) {
#pragma unroll
      for (int r = 0; r < NS_FB_RUN; r++) {
        if (r == nrun) {
          run[r].cx = c[0]; run[r].cy = c[1]; run[r].cz = c[2];
#pragma unroll
          for (int corner = 0; corner < 8; corner++) {
            run[r].a[corner] = ca[corner];
            run[r].b[corner] = cb[corner];
          }
        }
      }
      nrun++;
      open = true;
      open_small = small;
    }
  }
  nrun_out = nrun;
}


// (both steps in one: what the staged kernel calls)
__device__ __forceinline__ void fb_build_runs(const GridLayout& g, int l, int tile, int tid, const float* __restrict__ pos,
                                              const _Float16* __restrict__ dpu, long N, long nvalid, float fixed_scale, int vec,
                                              FbRun (&run)[NS_FB_RUN], int& nrun_out) {
  FbPos P;
  FbGrad G;
  fb_load_pos(tile, tid, pos, nvalid, vec, P);
  fb_load_grad(l, tile, tid, dpu, N, nvalid, vec, G);
  fb_runs_of(g, l, P, G, fixed_scale, run, nrun_out);
}

#ifdef NS_TEST_VARIANTS   // comparison kernel: libnerfslam_hip_variants.so only (common.h)
__global__ __launch_bounds__(256) void ngp_enc_fscatter_kernel(GridLayout g, FusedPlan fp, const float* __restrict__ pos,
                                                               const _Float16* __restrict__ dpu, long N, float fixed_scale,
                                                               int* __restrict__ ctr, int* __restrict__ cnt,
                                                               unsigned long long* __restrict__ queue,
                                                               ulonglong2* __restrict__ ovf, const int* __restrict__ n_dev,
                                                               int vec) {
  __shared__ unsigned long long rec[8 * NS_BIN_TILE];   // 64 KB: the tile's records, bin-sorted
  __shared__ int lbase[NS_FB_BINS + 1], lcnt[NS_FB_BINS], odst[NS_FB_BINS];
  const int k = blockIdx.y, l = fp.level[k], tile = blockIdx.x, tid = threadIdx.x;
  const uint32_t hs = g.offset[l + 1] - g.offset[l], res = (uint32_t)g.res[l];
  const bool hashed = fp.hashed[k] != 0;
  const int slot = fp.slot[k], shift = fp.shift[k];
  const uint32_t bmask = (1u << shift) - 1u;
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  if ((long)tile * NS_BIN_TILE >= nvalid) return;       // (uniform) nothing marched into this tile
  if (tid < NS_FB_BINS) lcnt[tid] = 0;
  FbRun run[NS_FB_RUN];
  int nrun;
  fb_build_runs(g, l, tile, tid, pos, dpu, N, nvalid, fixed_scale, vec, run, nrun);
  if (fp.merge[k]) fb_wave_merge(run, nrun);     // (uniform per workgroup)
  __syncthreads();
  // ---- rank the records inside their bins ----
  int rank[NS_FB_RUN][8];
#pragma unroll
  for (int r = 0; r < NS_FB_RUN; r++) {
    if (r < nrun) {
      uint32_t idx[8];
      {
        const uint32_t rc[3] = {run[r].cx, run[r].cy, run[r].cz};
        fb_indices_any(hashed, rc, hs, res, idx);
      }
#pragma unroll
      for (int corner = 0; corner < 8; corner++)   // (a contribution that rounds to zero in both fields is not a record: most
        rank[r][corner] = (run[r].a[corner] | run[r].b[corner]) != 0   //  corners of a converged scene's tiny gradients)
                              ? atomicAdd(&lcnt[idx[corner] >> shift], 1) : -1;
    }
  }
  __syncthreads();
  if (tid < 64) {   // one wave: the tile's bin counts -> LDS bases, run lengths next to the slots, spills reserved
    const int c = lcnt[tid];
    int ci = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int uc = __shfl_up(ci, d, 64);
      if (tid >= d) ci += uc;
    }
    lbase[tid] = ci - c;
    if (tid == 63) lbase[64] = ci;
    cnt[(long)(k * NS_FB_BINS + tid) * fp.ntiles + tile] = min(c, slot);
    odst[tid] = c > slot ? atomicAdd(&ctr[0], c - slot) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NS_FB_RUN; r++) {
    if (r < nrun) {
      uint32_t idx[8];
      {
        const uint32_t rc[3] = {run[r].cx, run[r].cy, run[r].cz};
        fb_indices_any(hashed, rc, hs, res, idx);
      }
#pragma unroll
      for (int corner = 0; corner < 8; corner++)
        if (rank[r][corner] >= 0)
          rec[lbase[idx[corner] >> shift] + rank[r][corner]] =
              ((unsigned long long)(idx[corner] & bmask) << 50) |
              ((unsigned long long)((uint32_t)run[r].b[corner] & 0x1ffffffu) << 25) |
              (unsigned long long)((uint32_t)run[r].a[corner] & 0x1ffffffu);
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int b = wave; b < NS_FB_BINS; b += 4) {
    const int b0 = lbase[b], n = lbase[b + 1] - b0;
    unsigned long long* __restrict__ dst = queue + (long)k * NS_FB_BINS * fp.ntiles * NS_FB_SLOT + ((long)b * fp.ntiles + tile) * slot;
    for (int e = lane; e < n; e += 64) {
      const unsigned long long r = rec[b0 + e];
      if (e < slot) {
        dst[e] = r;
      } else {
        const long o = (long)odst[b] + (e - slot);
        if (o < fp.ovf_cap) ovf[o] = make_ulonglong2(r, (unsigned long long)((k << 8) | b));
        else ctr[1] = 1;   // cannot happen with the worst-case list; flagged, never silent
      }
    }
  }
}
#endif  // NS_TEST_VARIANTS

// The scatter WITHOUT the LDS staging (round 4, default; NS_FB_SCATTER=1 selects the staged kernel above).  A record's place
// in its (bin, tile) slot is the value its LDS rank atomic returns -- no prefix sum over the bins is needed for that -- so every
// lane stores its records straight from registers: 8-byte stores that land in the slot's few 128-byte lines within one
// workgroup's lifetime and merge in the L2.  What goes: 64 KB of LDS per workgroup (two workgroups, 8 waves, per CU; now the
// registers bound it: 4 waves per SIMD), the pass that wrote the records to LDS, the pass that copied them out and two of the
// four barriers.  Records beyond a slot (rank >= slot: possible only when > 1/16 of a tile's records meet in one bin) go to
// the overflow list as before, after the one barrier that makes the counts final (a bit per record remembers which).
// Where the kernel's time goes (2^18 ray-ordered samples, stage-by-stage early exits, MI355X): launching 4096 workgroups that
// return at once 10.5 us; + loads and runs (VALU: cell, 8 weights, 16 fixed-point contributions per live sample and level) 28 us;
// + wave merge on the dense levels 31 us; + index hashes, ranks and record stores 47 us on a converged scene's gradients (most
// contributions round to zero: few stores) / 87 us on dense ones.  The accumulate pass with NO records costs 20 us (806
// workgroups: zero 64 KB of LDS, run lengths, flush scan); a converged scene's 1.5 M records + 0.8 M touched entries add 36 us,
// most of it the optimiser state: master / two moments / f16 copy are four arrays, so a sparsely touched entry costs seven
// scattered 8-byte accesses.
// (round 4, measured and not kept: d S formed once per sample instead of per corner -- exact for a power-of-two scale -- with the
//  clamps and the merge bound hoisted to the sample: 112 -> 32 instructions per 16 contributions, scatter 46.5 vs 47 us: the
//  "loads + runs" stage waits on its loads, not on the VALU)
// (round 4, measured and not kept: ONE workgroup per tile and FOUR levels -- positions loaded once, the four gradient rows up
//  front, the levels worked off from registers, 1024 workgroups instead of 4096: 174 registers (two waves per SIMD), scatter
//  47 -> 72 us on a converged scene's gradients, 87 -> 105 us on dense ones, training step 0.277 -> 0.310 ms.  The many small
//  workgroups hide each other's latencies better than the few long ones.)
__global__ __launch_bounds__(256) void ngp_enc_fscatter_direct_kernel(GridLayout g, FusedPlan fp, const float* __restrict__ pos,
                                                                      const _Float16* __restrict__ dpu, long N, float fixed_scale,
                                                                      int* __restrict__ ctr, int* __restrict__ cnt,
                                                                      unsigned long long* __restrict__ queue,
                                                                      ulonglong2* __restrict__ ovf, const int* __restrict__ n_dev,
                                                                      int vec) {
  __shared__ int lcnt[NS_FB_BINS], odst[NS_FB_BINS];
  __shared__ int s_spill;
  const int k = blockIdx.y, l = fp.level[k], tile = blockIdx.x, tid = threadIdx.x;
  const uint32_t hs = g.offset[l + 1] - g.offset[l], res = (uint32_t)g.res[l];
  const bool hashed = fp.hashed[k] != 0;
  const int slot = fp.slot[k], shift = fp.shift[k];
  const uint32_t bmask = (1u << shift) - 1u;
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  if ((long)tile * NS_BIN_TILE >= nvalid) return;       // (uniform) nothing marched into this tile
  if (tid < NS_FB_BINS) lcnt[tid] = 0;
  if (tid == 0) s_spill = 0;
  FbRun run[NS_FB_RUN];
  int nrun;
  fb_build_runs(g, l, tile, tid, pos, dpu, N, nvalid, fixed_scale, vec, run, nrun);
  if (fp.merge[k]) fb_wave_merge(run, nrun);     // (uniform per workgroup)
  __syncthreads();
  unsigned long long* __restrict__ qlev = queue + (long)k * NS_FB_BINS * fp.ntiles * NS_FB_SLOT + (long)tile * slot;
  // record (bin b, rank) of this tile's slots sits at qlev[b * bstride + rank].  In 32 bits -- b < 64, bstride = tiles x slot < 2^24
  // (checked by the host) -- it is ONE full-rate v_mad_u32_u24; as `(long)b * bstride + rank` it was two v_mul_lo_u32 and a
  // v_mad_u64_u32 per record store, quarter-rate instructions in a pass whose vector ALU is two thirds busy.
  const uint32_t bstride = (uint32_t)fp.ntiles * (uint32_t)slot;
  uint32_t spill = 0u;            // bit (8 r + corner): that record found its slot full
#pragma unroll
  for (int r = 0; r < NS_FB_RUN; r++) {
    int any = 0;                    // (a run whose 16 fields all rounded to zero -- common in a converged scene -- costs no hashes)
#pragma unroll
    for (int corner = 0; corner < 8; corner++) any |= run[r].a[corner] | run[r].b[corner];
    if (r < nrun && any != 0) {
      uint32_t idx[8];
      {
        const uint32_t rc[3] = {run[r].cx, run[r].cy, run[r].cz};
        fb_indices_any(hashed, rc, hs, res, idx);
      }
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        if ((run[r].a[corner] | run[r].b[corner]) == 0) continue;   // rounds to zero in both fields: not a record
        const int b = (int)(idx[corner] >> shift);
        const int rank = atomicAdd(&lcnt[b], 1);
        if (rank < slot)
          qlev[__umul24((uint32_t)b, bstride) + (uint32_t)rank] = ((unsigned long long)(idx[corner] & bmask) << 50) |
                                           ((unsigned long long)((uint32_t)run[r].b[corner] & 0x1ffffffu) << 25) |
                                           (unsigned long long)((uint32_t)run[r].a[corner] & 0x1ffffffu);
        else
          spill |= 1u << (8 * r + corner);
      }
    }
  }
  if (spill) s_spill = 1;
  __syncthreads();
  if (tid < NS_FB_BINS) {
    const int c = lcnt[tid];
    cnt[(long)(k * NS_FB_BINS + tid) * fp.ntiles + tile] = min(c, slot);
    odst[tid] = c > slot ? atomicAdd(&ctr[0], c - slot) : 0;
    lcnt[tid] = 0;                // (ranks among the spilled records, spilling tiles only)
  }
  if (!s_spill) return;           // (uniform)
  __syncthreads();
  // rare: a bin took more than its slot; the records that found it full go to the list, ranked among themselves
#pragma unroll
  for (int r = 0; r < NS_FB_RUN; r++) {
    if (r < nrun && ((spill >> (8 * r)) & 0xffu)) {
      uint32_t idx[8];
      {
        const uint32_t rc[3] = {run[r].cx, run[r].cy, run[r].cz};
        fb_indices_any(hashed, rc, hs, res, idx);
      }
#pragma unroll
      for (int corner = 0; corner < 8; corner++) {
        if (!((spill >> (8 * r + corner)) & 1u)) continue;
        const int b = (int)(idx[corner] >> shift);
        const long o = (long)odst[b] + atomicAdd(&lcnt[b], 1);
        const unsigned long long rec = ((unsigned long long)(idx[corner] & bmask) << 50) |
                                       ((unsigned long long)((uint32_t)run[r].b[corner] & 0x1ffffffu) << 25) |
                                       (unsigned long long)((uint32_t)run[r].a[corner] & 0x1ffffffu);
        if (o < fp.ovf_cap) ovf[o] = make_ulonglong2(rec, (unsigned long long)((k << 8) | b));
        else ctr[1] = 1;   // cannot happen with the worst-case list; flagged, never silent
      }
    }
  }
}

__device__ __forceinline__ void fb_add(unsigned long long* tab, unsigned long long r) {
  const uint32_t rel = (uint32_t)(r >> 50);
  const long long a = ((long long)(r << 39)) >> 39;          // sign-extended 25-bit fields
  const long long c = ((long long)(r << 14)) >> 39;
  atomicAdd(&tab[rel], (unsigned long long)(a + (c << 32)));
}

#define NS_FB_THREADS 512
#define NS_FB_GROUP 4     // slots a wave has in flight (two 64-record chunks each)
__global__ __launch_bounds__(NS_FB_THREADS) void ngp_enc_faccum_kernel(GridLayout g, FusedPlan fp, const int* __restrict__ ctr,
                                                                       const int* __restrict__ cnt,
                                                                       const unsigned long long* __restrict__ queue,
                                                                       const ulonglong2* __restrict__ ovf,
                                                                       float* __restrict__ grad, AdamFuse ad, long N,
                                                                       const int* __restrict__ n_dev, int* __restrict__ ctr_rw,
                                                                       int n_groups) {
  __shared__ unsigned long long tab[NS_FB_SLICE];
  __shared__ int scnt[NS_FB_THREADS], spre[NS_FB_THREADS], swave[NS_FB_THREADS / 64];
  __shared__ int s_novf;
  const int k = blockIdx.y, l = fp.level[k], b = blockIdx.x, tid = threadIdx.x;
  if (b >= fp.nbins[k]) return;
  const uint32_t hs = g.offset[l + 1] - g.offset[l];
  const uint32_t bsize = 1u << fp.shift[k];
  const uint32_t lo = (uint32_t)b * bsize;
  const uint32_t n_e = min(bsize, hs - lo);
  const long nvalid = n_dev ? min(N, (long)*n_dev) : N;
  const int ntv = (int)((nvalid + NS_BIN_TILE - 1) / NS_BIN_TILE);   // tiles the scatter pass wrote
  for (uint32_t e = tid; e < bsize; e += NS_FB_THREADS) tab[e] = 0ull;
  if (tid == 0) {
    s_novf = (int)min((long)ctr[0], fp.ovf_cap);
    // the LAST workgroup to have read the overflow count clears it (and this arrival counter): the list is empty for the next
    // call without a trailing launch
    __threadfence();
    if (atomicAdd(&ctr_rw[2], 1) == n_groups - 1) {
      ctr_rw[0] = 0;
      ctr_rw[2] = 0;
    }
  }
  __syncthreads();
  const long row = (long)(k * NS_FB_BINS + b) * fp.ntiles;
  const int slot = fp.slot[k];
  const unsigned long long* __restrict__ qbin = queue + (long)k * NS_FB_BINS * fp.ntiles * NS_FB_SLOT + (long)b * fp.ntiles * slot;
  const int wave = tid >> 6, lane = tid & 63;
  for (int t0 = 0; t0 < ntv; t0 += NS_FB_THREADS) {
    __syncthreads();
    const int mycnt = t0 + tid < ntv ? cnt[row + t0 + tid] : 0;
    scnt[tid] = mycnt;
    // inclusive prefix sums of the run lengths over the chunk's slots (wave scan, then the 8 wave totals)
    int incl = mycnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) swave[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < NS_FB_THREADS / 64; w++) woff += (w < wave) ? swave[w] : 0;
    spre[tid] = woff + incl - mycnt;                       // exclusive prefix: first record of slot tid in the chunk's order
    int total = 0;
#pragma unroll
    for (int w = 0; w < NS_FB_THREADS / 64; w++) total += swave[w];
    __syncthreads();
    const int nt = min(NS_FB_THREADS, ntv - t0);
    // SPARSE bin (a converged scene: a handful of records per slot).  The slot-by-slot loop below spends a load round trip on
    // every group of four slots whatever they hold -- 7 dependent rounds per wave for a 213-tile step, most of a work item's
    // time when the slots are nearly empty.  Here record k of the chunk (k < total) is found by binary search in the prefix
    // sums, one record per lane and round: ceil(total / 512) rounds.
    if (total < 8 * nt) {                                   // (uniform per workgroup)
      for (int k0 = 0; k0 < total; k0 += NS_FB_THREADS) {
        const int k = k0 + tid;
        if (k < total) {
          int lo = 0, hi = nt - 1;                          // last slot s with spre[s] <= k
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (spre[mid] <= k) lo = mid; else hi = mid - 1;
          }
          const int e = k - spre[lo];
          if (e < scnt[lo]) fb_add(tab, qbin[(long)(t0 + lo) * slot + e]);
        }
      }
      continue;
    }
    for (int tt = wave * NS_FB_GROUP; tt < nt; tt += (NS_FB_THREADS / 64) * NS_FB_GROUP) {
      int c[NS_FB_GROUP];
      unsigned long long r[NS_FB_GROUP][2];
#pragma unroll
      for (int u = 0; u < NS_FB_GROUP; u++) {
        c[u] = tt + u < nt ? scnt[tt + u] : 0;
        const unsigned long long* __restrict__ q = qbin + (long)(t0 + tt + u) * slot;
        r[u][0] = lane < c[u] ? q[lane] : 0ull;          // (a zero record adds zero to entry 0: no branch in the add loop)
        r[u][1] = lane + 64 < c[u] ? q[lane + 64] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < NS_FB_GROUP; u++) {
        if (r[u][0] != 0ull) fb_add(tab, r[u][0]);
        if (r[u][1] != 0ull) fb_add(tab, r[u][1]);
      }
#pragma unroll
      for (int u = 0; u < NS_FB_GROUP; u++) {            // slots filled beyond the mean: the rest of the run
        const unsigned long long* __restrict__ q = qbin + (long)(t0 + tt + u) * slot;
        for (int e = lane + 128; e < c[u]; e += 64) fb_add(tab, q[e]);
      }
    }
  }
  const int novf = s_novf;     // (written before the first barrier of the tile loop / the one below)
  if (novf > 0) {
    const unsigned long long tag = (unsigned long long)((k << 8) | b);
    for (int e = tid; e < novf; e += NS_FB_THREADS) {
      const ulonglong2 r = ovf[e];
      if (r.y == tag) fb_add(tab, r.x);
    }
  }
  __syncthreads();
  const long base = (long)g.offset[l] + lo;
  if (ad.master != nullptr) {
    const float c1 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C1]) : ad.c1, c2 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C2]) : ad.c2;
    for (uint32_t e = tid; e < n_e; e += NS_FB_THREADS) {
      const unsigned long long word = tab[e];
      if (word != 0ull) adam_entry(ad, base + e, word, c1, c2);
    }
  } else if (ad.emit_list != nullptr) {
    // compact the bin's touched entries into the trainer's list: per-lane counts -> workgroup prefix sums (the arrays of the tile
    // loop are free again) -> ONE atomic per workgroup reserves the range -> (entry, sum) pairs.  The list's order depends on
    // the workgroups' arrival; what is done with it (integer sums per entry) does not.
    int mine = 0;
    for (uint32_t e = tid; e < n_e; e += NS_FB_THREADS) mine += tab[e] != 0ull ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) swave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NS_FB_THREADS / 64; w++) {
      woff += (w < wave) ? swave[w] : 0;
      total += swave[w];
    }
    if (tid == 0) s_novf = total > 0 ? atomicAdd(ad.emit_count, total) : 0;
    __syncthreads();
    int pos = s_novf + woff + incl - mine;
    for (uint32_t e = tid; e < n_e; e += NS_FB_THREADS) {
      const unsigned long long word = tab[e];
      if (word != 0ull) ad.emit_list[pos++] = make_ulonglong2((unsigned long long)(base + e), word);
    }
  } else {
    unsigned long long* __restrict__ g64 = reinterpret_cast<unsigned long long*>(grad) + base;
    for (uint32_t e = tid; e < n_e; e += NS_FB_THREADS) {
      const unsigned long long word = tab[e];
      if (word != 0ull) g64[e] += word;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Replicated trainers: the exchanged lists -> the table.  `lists` holds `n_lists` lists of `stride` (entry, packed sum) pairs each
// (list r = what trainer r's flush emitted, counts[r] of them valid).
//   accumulate: acc[entry] += sum, 64-bit integer atomics (exact and order-free: the same word whatever the lists' order)
//   apply:      the first lane to exchange an entry's accumulator for 0 applies Adam with the complete sum (every other pair of
//               the same entry finds 0 and does nothing): every trainer computes the same update from the same sums, no
//               parameter has to travel back, and acc is zero again for the next step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ngp_sparse_accumulate_kernel(const ulonglong2* __restrict__ lists, const int* __restrict__ counts,
                                                                    long stride, unsigned long long* __restrict__ acc) {
  const int r = blockIdx.y;
  const int n = counts[r];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const ulonglong2 it = lists[(long)r * stride + i];
    atomicAdd(acc + it.x, it.y);
  }
}

__global__ __launch_bounds__(256) void ngp_sparse_apply_kernel(const ulonglong2* __restrict__ lists, const int* __restrict__ counts,
                                                               long stride, unsigned long long* __restrict__ acc, AdamFuse ad) {
  const int r = blockIdx.y;
  const int n = counts[r];
  const float c1 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C1]) : ad.c1, c2 = ad.ctl ? __int_as_float(ad.ctl[NS_CTL_C2]) : ad.c2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long entry = (long)lists[(long)r * stride + i].x;
    const unsigned long long word = atomicExch(acc + entry, 0ull);
    if (word != 0ull) adam_entry(ad, entry, word, c1, c2);
  }
}

// ---------------------------------------------------------------------------------------------
// Camera-pose refinement (`optimize_extrinsics`, nerf_fusion.py:99,123 [EXTERNAL arithmetic: instant-ngp]).
//   dL/dpos of every sample  = the encoding's input gradient (trilinear weights differentiated),
//   per ray:  g_o = sum dL/dpos,   g_d = sum t * dL/dpos            (pos = o + t d)
//   per image: dL/d(translation) = sum g_o,   dL/d(rotation, left perturbation exp(w) R) = sum d x g_d
//   Adam on the 6 dof of every image, applied as c2w <- [exp(dw) R | t + dt].
// ---------------------------------------------------------------------------------------------
// one lane per sample, loop over the levels (no atomics); positions in the unit cube; dLdfeatT unit-major [2L][N];
// out[N,3] = dL/d(unit position)
__global__ __launch_bounds__(256) void ngp_encode_bwd_input_kernel(GridLayout g, const float* __restrict__ pos,
                                                                   const h2_t* __restrict__ params,
                                                                   const _Float16* __restrict__ dLdfeatT,
                                                                   float* __restrict__ dLdpos, long N, int L,
                                                                   const int* __restrict__ n_dev) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N || (n_dev != nullptr && i >= (long)*n_dev)) return;
  const float px = pos[i * 3], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int l = 0; l < L; l++) {
    const float d0 = (float)dLdfeatT[(long)(2 * l) * N + i], d1 = (float)dLdfeatT[(long)(2 * l + 1) * N + i];
    if (d0 == 0.0f && d1 == 0.0f) continue;
    const uint32_t hs = g.offset[l + 1] - g.offset[l];
    const float scale = g.scale[l];
    const uint32_t res = (uint32_t)g.res[l];
    const bool hashed = grid_level_hashed(hs, res);
    float w[3];
    uint32_t c[3];
    const float pp[3] = {px, py, pz};
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const float p = fmaf(scale, pp[d], 0.5f);
      const float fl = floorf(p);
      c[d] = (uint32_t)(int)fl;
      w[d] = p - fl;
    }
    const h2_t* __restrict__ tab = params + g.offset[l];
    float s[8];  // dL/dfeat . value of corner
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const h2_t v = tab[grid_index_lvl(hashed, hs, res, c[0] + (corner & 1), c[1] + ((corner >> 1) & 1), c[2] + (corner >> 2))];
      s[corner] = d0 * (float)v[0] + d1 * (float)v[1];
    }
    // d/dx of sum_c wx(c) wy(c) wz(c) s_c = sum over the 4 (y,z) edges of wy wz (s_{x=1} - s_{x=0}), times scale
    const float wy0 = 1.0f - w[1], wy1 = w[1], wz0 = 1.0f - w[2], wz1 = w[2], wx0 = 1.0f - w[0], wx1 = w[0];
    gx += scale * (wy0 * wz0 * (s[1] - s[0]) + wy1 * wz0 * (s[3] - s[2]) + wy0 * wz1 * (s[5] - s[4]) + wy1 * wz1 * (s[7] - s[6]));
    gy += scale * (wx0 * wz0 * (s[2] - s[0]) + wx1 * wz0 * (s[3] - s[1]) + wx0 * wz1 * (s[6] - s[4]) + wx1 * wz1 * (s[7] - s[5]));
    gz += scale * (wx0 * wy0 * (s[4] - s[0]) + wx1 * wy0 * (s[5] - s[1]) + wx0 * wy1 * (s[6] - s[2]) + wx1 * wy1 * (s[7] - s[3]));
  }
  dLdpos[i * 3] = gx;
  dLdpos[i * 3 + 1] = gy;
  dLdpos[i * 3 + 2] = gz;
}

// dL/dpos from the Jacobian rows the forward pass wrote (jacT [6 L][N] f16, without the level scale) and dL/dfeature
// ([2 L][N] f16): 256 B streamed per sample, no gathers
__global__ __launch_bounds__(256) void ngp_encode_jac_dot_kernel(GridLayout g, const _Float16* __restrict__ jacT,
                                                                 const _Float16* __restrict__ dLdfeatT, float* __restrict__ dLdpos,
                                                                 long N, int L, const int* __restrict__ n_dev) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N || (n_dev != nullptr && i >= (long)*n_dev)) return;
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int l = 0; l < L; l++) {
    const float sc = g.scale[l];
    const float d0 = (float)dLdfeatT[(long)(2 * l) * N + i] * sc, d1 = (float)dLdfeatT[(long)(2 * l + 1) * N + i] * sc;
    const _Float16* __restrict__ j = jacT + (long)(6 * l) * N + i;
    gx += d0 * (float)j[0] + d1 * (float)j[3 * N];
    gy += d0 * (float)j[N] + d1 * (float)j[4 * N];
    gz += d0 * (float)j[2 * N] + d1 * (float)j[5 * N];
  }
  dLdpos[i * 3] = gx;
  dLdpos[i * 3 + 1] = gy;
  dLdpos[i * 3 + 2] = gz;
}

// one wave per ray: reduce the sample gradients to the 6-dof gradient of the ray's camera, atomically into cam_grad[img]
__global__ __launch_bounds__(256) void ngp_camera_grad_kernel(const float* __restrict__ dLdpos, const float* __restrict__ tmid,
                                                              const float* __restrict__ rays_d,
                                                              const int* __restrict__ ray_start, const int* __restrict__ ray_n,
                                                              const int* __restrict__ ray_img, float pos_inv,
                                                              float* __restrict__ cam_grad, int Rcap,
                                                              const int* __restrict__ ctl, float* __restrict__ ray_g) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int R = ctl ? min(ctl[NS_CTL_RAYS], Rcap) : Rcap;
  if (r >= R) return;
  const int s0 = ray_start[r], n = ray_n[r];
  if (n <= 0) return;
  float o0 = 0, o1 = 0, o2 = 0, d0 = 0, d1 = 0, d2 = 0;
  // four independent 64-sample chunks per round: the walk along a ray is a chain of dependent round trips otherwise
  // (a 1024-sample ray: 16 of them; measured 114 us per step for ~2000 rays, the slowest wave's chain)
  for (int k0 = 0; k0 < n; k0 += 256) {
    float t[4], a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u * 64 + lane;
      const long s = (long)s0 + (k < n ? k : 0);
      const float m = k < n ? pos_inv : 0.0f;
      t[u] = tmid[s];
      a[u] = dLdpos[s * 3] * m;
      b[u] = dLdpos[s * 3 + 1] * m;
      c[u] = dLdpos[s * 3 + 2] * m;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      o0 += a[u]; o1 += b[u]; o2 += c[u];
      d0 += t[u] * a[u]; d1 += t[u] * b[u]; d2 += t[u] * c[u];
    }
  }
  o0 = wave_sum(o0); o1 = wave_sum(o1); o2 = wave_sum(o2);
  d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
  if (lane == 0) {
    const float x = rays_d[r * 3], y = rays_d[r * 3 + 1], z = rays_d[r * 3 + 2];
    if (ray_g != nullptr) {   // two-stage form: the per-image sums are formed by ngp_camera_grad_reduce_kernel
      float* gp = ray_g + (long)r * 6;
      gp[0] = o0; gp[1] = o1; gp[2] = o2;
      gp[3] = y * d2 - z * d1; gp[4] = z * d0 - x * d2; gp[5] = x * d1 - y * d0;
      return;
    }
    float* gp = cam_grad + (long)ray_img[r] * 6;
    atomicAdd(gp + 0, o0);
    atomicAdd(gp + 1, o1);
    atomicAdd(gp + 2, o2);
    atomicAdd(gp + 3, y * d2 - z * d1);  // d x g_d
    atomicAdd(gp + 4, z * d0 - x * d2);
    atomicAdd(gp + 5, x * d1 - y * d0);
  }
}

// Second stage of the camera gradient: ONE workgroup adds the per-ray vectors into a per-image table in LDS and adds the
// table to cam_grad.  (Thousands of rays share a few dozen images: as global atomics those are same-address atomics,
// which this part executes at the memory side one after the other -- 113 us per step for ~2000 rays.)
#define NS_CAM_LDS_IMAGES 4096
__global__ __launch_bounds__(1024) void ngp_camera_grad_reduce_kernel(const float* __restrict__ ray_g, const int* __restrict__ ray_n,
                                                                      const int* __restrict__ ray_img, float* __restrict__ cam_grad,
                                                                      int Rcap, int n_images, const int* __restrict__ ctl) {
  __shared__ float tab[NS_CAM_LDS_IMAGES * 6];
  const int R = ctl ? min(ctl[NS_CTL_RAYS], Rcap) : Rcap;
  const int nimg = min(n_images, NS_CAM_LDS_IMAGES);
  for (int e = threadIdx.x; e < nimg * 6; e += 1024) tab[e] = 0.0f;
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 1024) {
    if (ray_n[r] <= 0) continue;
    const int img = ray_img[r];
    if (img < 0 || img >= nimg) continue;
#pragma unroll
    for (int k = 0; k < 6; k++) atomicAdd(&tab[img * 6 + k], ray_g[(long)r * 6 + k]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nimg * 6; e += 1024) {
    const float v = tab[e];
    if (v != 0.0f) cam_grad[e] += v;
  }
}

// The same for up to NS_CAM_SMALL views (the SLAM case: a dozen keyframes in the mapper at a time), as a workgroup that fits
// ANYWHERE: 4 waves, 24 KB of LDS (a private table per wave, added up in wave order), four rays per lane with all their loads in
// flight.  The kernel above is one 1024-thread workgroup with 98 KB of LDS at the tail of the pose chain, while two accumulate
// workgroups of the table gradient sit on every CU: it waited for a whole CU to drain (21-58 us in the step's timeline).
#define NS_CAM_SMALL 256
__global__ __launch_bounds__(256) void ngp_camera_grad_reduce_small_kernel(const float* __restrict__ ray_g, const int* __restrict__ ray_n,
                                                                           const int* __restrict__ ray_img, float* __restrict__ cam_grad,
                                                                           int Rcap, int n_images, const int* __restrict__ ctl) {
  __shared__ float tab[4 * NS_CAM_SMALL * 6];
  const int R = ctl ? min(ctl[NS_CTL_RAYS], Rcap) : Rcap;
  const int nimg = min(n_images, NS_CAM_SMALL), stride = nimg * 6;
  for (int e = threadIdx.x; e < 4 * stride; e += 256) tab[e] = 0.0f;
  __syncthreads();
  float* mine = tab + (threadIdx.x >> 6) * stride;
  for (int r0 = threadIdx.x; r0 < R; r0 += 4 * 256) {
    int img[4];
    float gv[4][6];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + u * 256;
      const bool ok = r < R;
      const int rr = ok ? r : 0;
      const int n = ray_n[rr], im = ray_img[rr];
      const float2* __restrict__ gp = reinterpret_cast<const float2*>(ray_g + (long)rr * 6);
      const float2 a = gp[0], b = gp[1], c = gp[2];
      gv[u][0] = a.x; gv[u][1] = a.y; gv[u][2] = b.x; gv[u][3] = b.y; gv[u][4] = c.x; gv[u][5] = c.y;
      img[u] = (ok && n > 0 && im >= 0 && im < nimg) ? im : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (img[u] < 0) continue;
#pragma unroll
      for (int k = 0; k < 6; k++) atomicAdd(&mine[img[u] * 6 + k], gv[u][k]);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < stride; e += 256) {
    const float v = (tab[e] + tab[stride + e]) + (tab[2 * stride + e] + tab[3 * stride + e]);
    if (v != 0.0f) cam_grad[e] += v;
  }
}

// one lane per image: Adam on (dt, dw), then c2w <- [exp(dw) R | t + dt]; clears the gradient
__global__ __launch_bounds__(64) void ngp_camera_step_kernel(float* __restrict__ c2w, float* __restrict__ cam_grad,
                                                             float* __restrict__ m1, float* __restrict__ m2, int n, float c1,
                                                             float c2, float lr_pos, float lr_rot, float beta1, float beta2,
                                                             float eps, float inv_grad_scale, const int* __restrict__ ctl) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (ctl) {  // graph-captured step: view count and bias corrections from the device control block
    n = min(n, ctl[NS_CTL_VIEWS]);
    const float st = (float)(ctl[NS_CTL_STEP] + 1);
    c1 = 1.0f - powf(beta1, st);
    c2 = 1.0f - powf(beta2, st);
  }
  if (i >= n) return;
  float step[6];
  bool any = false;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float gk = cam_grad[i * 6 + k] * inv_grad_scale;
    cam_grad[i * 6 + k] = 0.0f;
    step[k] = 0.0f;
    if (gk != 0.0f) {
      any = true;
      const float a = beta1 * m1[i * 6 + k] + (1.0f - beta1) * gk;
      const float b = beta2 * m2[i * 6 + k] + (1.0f - beta2) * gk * gk;
      m1[i * 6 + k] = a;
      m2[i * 6 + k] = b;
      step[k] = -(k < 3 ? lr_pos : lr_rot) * (a / c1) / (sqrtf(b / c2) + eps);
    }
  }
  if (!any) return;  // image not hit by a ray this step
  float* M = c2w + (long)i * 12;
  M[3] += step[0];
  M[7] += step[1];
  M[11] += step[2];
  // Rodrigues: exp(w) = I + sin(th)/th K + (1 - cos th)/th^2 K^2
  const float wx = step[3], wy = step[4], wz = step[5];
  const float th2 = wx * wx + wy * wy + wz * wz, th = sqrtf(th2);
  const float A = th < 1e-6f ? 1.0f - th2 / 6.0f : sinf(th) / th;
  const float B = th < 1e-6f ? 0.5f - th2 / 24.0f : (1.0f - cosf(th)) / th2;
  const float E[9] = {1.0f - B * (wy * wy + wz * wz), B * wx * wy - A * wz,        B * wx * wz + A * wy,
                      B * wx * wy + A * wz,        1.0f - B * (wx * wx + wz * wz), B * wy * wz - A * wx,
                      B * wx * wz - A * wy,        B * wy * wz + A * wx,        1.0f - B * (wx * wx + wy * wy)};
  float Rn[9];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) Rn[a * 3 + b] = E[a * 3] * M[b] + E[a * 3 + 1] * M[4 + b] + E[a * 3 + 2] * M[8 + b];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) M[a * 4 + b] = Rn[a * 3 + b];
}

// ---------------------------------------------------------------------------------------------
// Adam: f32 master parameters + moments, f16 working copy refreshed (tiny-cuda-nn semantics:
// entries with a zero gradient and no weight decay are skipped so untouched hash cells keep their
// moments).  One streaming pass, float4-vectorised.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ngp_adam_kernel(float* __restrict__ master, _Float16* __restrict__ hp,
                                                       float* __restrict__ grad, float* __restrict__ m1,
                                                       float* __restrict__ m2, long n, float c1, float c2, float lr,
                                                       float beta1, float beta2, float eps, float l2,
                                                       float inv_grad_scale, float inv_fixed_scale,
                                                       const int* __restrict__ ctl, int es) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (ctl) {  // graph-captured step: bias corrections of step ctl[0] + 1, precomputed by ns_ngp_step_advance
    c1 = __int_as_float(ctl[NS_CTL_C1]);
    c2 = __int_as_float(ctl[NS_CTL_C2]);
  }
  float g;
  if (inv_fixed_scale > 0.0f) {  // packed fixed-point pairs (see pack_fixed): both lanes of a pair read the word
    float g0, g1;
    unpack_fixed(reinterpret_cast<const unsigned long long*>(grad)[i >> 1], inv_fixed_scale, g0, g1);
    g = ((i & 1) ? g1 : g0) * inv_grad_scale;
  } else {
    g = grad[i] * inv_grad_scale;
  }
  __builtin_amdgcn_wave_barrier();
  grad[i] = 0.0f;  // leaves the gradient buffer ready for the next step (each lane clears its half of the word)
  // es = 2: dense arrays, parameter i at [i].  es = 8 / 6 (adam_entry_stride): parameter i is field (i & 1) of entry i >> 1's record;
  // an untouched parameter's record is then not read at all -- its working copy already is the rounded master (every writer
  // of one writes the other) -- so that a sparse gradient costs the touched records, not a sweep over 32 B per entry.
  const long k = es == 2 ? i : (i >> 1) * es + (i & 1);
  const bool live = !(g == 0.0f && l2 == 0.0f);
  if (es != 2 && !live) return;
  float p = master[k];
  if (live) {
    float a = m1[k], b = m2[k];
    p = adam_apply(p, g, l2, a, b, c1, c2, lr, beta1, beta2, eps);
    m1[k] = a;
    m2[k] = b;
    master[k] = p;
  }
  hp[i] = (_Float16)p;
}

// ---------------------------------------------------------------------------------------------
// Training-ray sampling: one lane per ray picks (image, pixel) with a counter-based hash of (seed, ray),
// builds the ray from the camera-to-world matrix, clips it against the render box and gathers the
// supervision (linear rgb, depth, depth covariance) of that pixel.  Replaces ~20 small torch launches per
// training step (randint x3, index gathers, einsum, normalisation, slab test).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t ns_pcg(uint32_t v) {
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

struct SampleRaysArgs {
  const float* images;  // [n,H,W,4]
  const float* depths;  // [n,H,W]
  const float* covs;    // [n,H,W]
  const float* c2w;     // [n,3,4]
  float fx, fy, cx, cy, box_lo, box_hi, near;
  int n, H, W, R;
  uint32_t seed;
  float *rays_o, *rays_d, *t_range, *gt_rgb, *gt_depth, *gt_cov;
  int* ray_img;  // optional [R]: image index of every ray (camera-pose refinement)
  const int* ctl;
  int step_offset;  // ctl mode: the rays are those of step ctl[0] + step_offset (1: sampled ahead, while step ctl[0] finishes)
};

__global__ __launch_bounds__(256) void ngp_sample_rays_kernel(SampleRaysArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int R = a.ctl ? min(a.ctl[NS_CTL_RAYS], a.R) : a.R;
  if (r >= R) return;
  // same seed schedule as the host path (nerfslam/ngp.py): seed * 0x9E3779B1 + step * 0x85EBCA77
  const uint32_t seed = a.ctl ? (uint32_t)a.ctl[NS_CTL_SEED] * 0x9E3779B1u +
                                    (uint32_t)(a.ctl[NS_CTL_STEP] + a.step_offset) * 0x85EBCA77u
                              : a.seed;
  const int nimg = a.ctl ? a.ctl[NS_CTL_VIEWS] : a.n;
  const uint32_t base = seed + (uint32_t)r * 3u;
  const int img = (int)(ns_pcg(base) % (uint32_t)nimg);
  const int u = (int)(ns_pcg(base + 1u) % (uint32_t)a.W);
  const int v = (int)(ns_pcg(base + 2u) % (uint32_t)a.H);
  const float* M = a.c2w + (long)img * 12;
  const float dcx = ((float)u + 0.5f - a.cx) / a.fx, dcy = ((float)v + 0.5f - a.cy) / a.fy;
  float d[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    d[k] = M[k * 4] * dcx + M[k * 4 + 1] * dcy + M[k * 4 + 2];
    o[k] = M[k * 4 + 3];
  }
  const float inv_n = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    d[k] *= inv_n;
    const float inv = 1.0f / (fabsf(d[k]) < 1e-9f ? 1e-9f : d[k]);
    const float t0 = (a.box_lo - o[k]) * inv, t1 = (a.box_hi - o[k]) * inv;
    tmin = fmaxf(tmin, fminf(t0, t1));
    tmax = fminf(tmax, fmaxf(t0, t1));
  }
  tmin = fmaxf(tmin, a.near);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    a.rays_o[r * 3 + k] = o[k];
    a.rays_d[r * 3 + k] = d[k];
  }
  a.t_range[r * 2] = tmin;
  a.t_range[r * 2 + 1] = fmaxf(tmax, tmin);
  const long pix = ((long)img * a.H + v) * a.W + u;
  a.gt_rgb[r * 3] = a.images[pix * 4];
  a.gt_rgb[r * 3 + 1] = a.images[pix * 4 + 1];
  a.gt_rgb[r * 3 + 2] = a.images[pix * 4 + 2];
  a.gt_depth[r] = a.depths[pix];
  a.gt_cov[r] = fmaxf(a.covs[pix], 1e-6f);
  if (a.ray_img != nullptr) a.ray_img[r] = img;
}

// ---------------------------------------------------------------------------------------------
// Occupancy grid: `ncasc` cascades of G^3 bits; cascade m covers [0.5 - 2^(m-1), 0.5 + 2^(m-1)]^3.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int mip_of(float x, float y, float z, float dt, int G, int ncasc) {
  const float m = fmaxf(fabsf(x - 0.5f), fmaxf(fabsf(y - 0.5f), fabsf(z - 0.5f)));
  int mip = 0;
  while (mip < ncasc - 1 && m >= 0.5f * (float)(1 << mip)) mip++;
  while (mip < ncasc - 1 && dt * (float)G > (float)(1 << mip)) mip++;
  return mip;
}

__device__ __forceinline__ bool occupied(const uint8_t* __restrict__ bits, float x, float y, float z, float dt,
                                         int G, int ncasc) {
  const int mip = mip_of(x, y, z, dt, G, ncasc);
  const float s = 1.0f / (float)(1 << mip);
  const int cx = (int)floorf(((x - 0.5f) * s + 0.5f) * (float)G);
  const int cy = (int)floorf(((y - 0.5f) * s + 0.5f) * (float)G);
  const int cz = (int)floorf(((z - 0.5f) * s + 0.5f) * (float)G);
  if (cx < 0 || cx >= G || cy < 0 || cy >= G || cz < 0 || cz >= G) return false;
  const long idx = ((long)mip * G + cz) * G * G + (long)cy * G + cx;
  return (bits[idx >> 3] >> (idx & 7)) & 1;
}

struct MarchArgs {
  const uint8_t* bits;
  const float* rays_o;   // [R,3]
  const float* rays_d;   // [R,3] unit
  const float* t_range;  // [R,2] entry / exit distance of the ray through the render box
  float cone, min_step, max_step;
  float pos_lo, pos_inv;  // written positions = (p - pos_lo) * pos_inv  (0, 1: scene coordinates)
  int G, ncasc, max_per_ray;
  long max_samples;
  int* counter;          // [3]: samples requested, rays with samples, end of the last reserved range (zeroed by the caller)
  int* ray_start;        // [R]
  int* ray_n;            // [R]
  float* pos;            // [max_samples,3]
  float* dirs;           // [max_samples,3]
  float* dt;             // [max_samples]
  float* tmid;           // [max_samples]
  int R;
  const int* ctl;
  unsigned long long* order;   // optional [1 + ceil(R / 16)], all zero between launches: ranges handed out in WORKGROUP ORDER
};

// Ray marching, 16 lanes per ray (one DPP row), 4 rays per wave.
//
// The step sequence t_{k+1} = t_k + clamp(t_k * cone, min_step, max_step) depends on t0 only, not on
// the grid, and costs three VALU ops per step; the expensive part of a serial marcher is the dependent
// bit fetch per step (the previous one-lane-per-ray kernel: 1.09 ms for 4096 rays, 64 waves on the
// whole chip).  Here lane `sub` of a row replays the cheap recurrence up to its own block of 64
// consecutive steps (bit-identical t values to a serial loop), fetches the 64 occupancy bits in batches
// of 8 independent loads into a 64-bit mask, and a row prefix sum of the popcounts gives every lane its
// offset in the ray's sample range.  Rays longer than 1024 steps take further rounds.  One atomic per
// ray reserves the range (same protocol as before: a ray that does not fit gets nothing).
#define MARCH_SPL 64
#define MARCH_ROUND (16 * MARCH_SPL)
#define MARCH_WIN 256  // samples of one ray staged in LDS at a time (16 rays x 256 x 8 B = 32 KiB per workgroup)

__device__ __forceinline__ float march_dt(float t, const MarchArgs& a) {
  return fminf(fmaxf(t * a.cone, a.min_step), a.max_step);
}

__device__ __forceinline__ float march_advance(float t, int steps, float t1, const MarchArgs& a) {
  for (int k = 0; k < steps && t < t1; k++) t += march_dt(t, a);
  return t;
}

__device__ __forceinline__ long march_cell(float x, float y, float z, float dt, int G, int ncasc) {
  const int mip = mip_of(x, y, z, dt, G, ncasc);
  const float s = 1.0f / (float)(1 << mip);
  const int cx = (int)floorf(((x - 0.5f) * s + 0.5f) * (float)G);
  const int cy = (int)floorf(((y - 0.5f) * s + 0.5f) * (float)G);
  const int cz = (int)floorf(((z - 0.5f) * s + 0.5f) * (float)G);
  if (cx < 0 || cx >= G || cy < 0 || cy >= G || cz < 0 || cz >= G) return -1;
  return ((long)mip * G + cz) * G * G + (long)cy * G + cx;
}

struct MarchRay {
  float ox, oy, oz, dx, dy, dz, t1;
};

// occupancy bits of the 64 steps that start at t
// (t is advanced to the end of the block, or to the first step at or beyond t1 where the block stops early)
__device__ __forceinline__ uint64_t march_scan(const MarchArgs& a, const MarchRay& ry, float& t) {
  uint64_t mask = 0;
  for (int g = 0; g < MARCH_SPL; g += 8) {
    if (!(t < ry.t1)) break;
    uint32_t byte[8], sh[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const float dt = march_dt(t, a);
      const long idx = (t < ry.t1) ? march_cell(__fmaf_rn(t, ry.dx, ry.ox), __fmaf_rn(t, ry.dy, ry.oy),
                                                __fmaf_rn(t, ry.dz, ry.oz), dt, a.G, a.ncasc)
                                   : -1;
      byte[q] = idx >= 0 ? (uint32_t)a.bits[idx >> 3] : 0u;
      sh[q] = (uint32_t)(idx & 7);
      t += dt;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) mask |= (uint64_t)((byte[q] >> sh[q]) & 1u) << (g + q);
  }
  return mask;
}

__device__ __forceinline__ int row_inclusive_sum(int v, int sub) {  // over the 16 lanes of a row
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    const int u = __shfl_up(v, d, 16);
    if (sub >= d) v += u;
  }
  return v;
}

__global__ __launch_bounds__(256) void ngp_march_kernel(MarchArgs a) {
  const int sub = threadIdx.x & 15;
  // Ordered mode: the ray block of a workgroup is the TICKET it draws on entry, not blockIdx.x.  The look-back below
  // waits for the words of all LOWER blocks; with tickets those belong to workgroups that are already running, whatever order the
  // dispatcher starts workgroups in and whatever else occupies the CUs (a graph's other branches): deadlock-free by construction.
  // Block b still marches rays [16 b, 16 b + 16) and gets the range after blocks 0 .. b-1: the batch is the same on every run.
  __shared__ int s_block;
  if (a.order != nullptr) {
    if (threadIdx.x == 0) s_block = (int)atomicAdd(&a.order[1], 1ull);
    __syncthreads();
  }
  const int vb = a.order != nullptr ? s_block : (int)blockIdx.x;
  const int r = vb * 16 + (threadIdx.x >> 4);
  const int R = a.ctl ? min(a.ctl[NS_CTL_RAYS], a.R) : a.R;
  if (vb * 16 >= R) {  // workgroup-uniform: the grid is sized for the capacity
    // (an idle workgroup still counts out: the last of ALL workgroups of the launch clears the words and the two counters)
    if (a.order != nullptr && threadIdx.x == 0 && atomicAdd(&a.order[0], 1ull) == (unsigned long long)(gridDim.x - 1)) {
      const int nwg = (R + 15) >> 4;
      for (int j = 0; j < nwg; j++) a.order[2 + j] = 0ull;
      a.order[0] = 0ull;
      a.order[1] = 0ull;
    }
    return;
  }
  const bool live = r < R;            // row-uniform; dead rows run with an empty interval
  const int rr = live ? r : R - 1;
  MarchRay ry{a.rays_o[rr * 3], a.rays_o[rr * 3 + 1], a.rays_o[rr * 3 + 2],
              a.rays_d[rr * 3], a.rays_d[rr * 3 + 1], a.rays_d[rr * 3 + 2], 0.0f};
  const float t0 = a.t_range[rr * 2];
  ry.t1 = live ? a.t_range[rr * 2 + 1] : t0;
  const float tb = march_advance(t0, sub * MARCH_SPL, ry.t1, a);  // start of this lane's block in round 0
  // pass 1: count
  uint64_t mask0 = 0;
  int total = 0, rounds = 0;
  float tc = tb;
  for (;;) {
    float te = tc;
    const uint64_t m = march_scan(a, ry, te);
    if (rounds == 0) mask0 = m;
    total += __shfl(row_inclusive_sum(__popcll(m), sub), 15, 16);
    rounds++;
    // another round only if the last lane's block ended before t1; only then is the next block start
    // worth the 960-step replay (it used to be replayed unconditionally: a serial chain as long as the prologue's)
    const int more = __shfl((int)(te < ry.t1), 15, 16);
    if (!more || total >= a.max_per_ray) break;
    tc = march_advance(te, MARCH_ROUND - MARCH_SPL, ry.t1, a);
  }
  int n = min(total, a.max_per_ray);
  // One reservation per WORKGROUP (16 rays), not per ray: thousands of returning atomics on one address serialise at
  // ~10 ns each, which was the whole run time of this kernel (143 us for ~5000 rays whatever the marching cost).
  // Rays keep their order inside the block's range; a ray whose range would cross max_samples is refused, and so
  // are all later ones (their bases are larger), so the accepted ranges still tile [0, counter[2]) without holes.
  __shared__ int row_n[16], row_base[16];
  const int row = threadIdx.x >> 4;
  if (sub == 0) row_n[row] = live ? n : 0;
  __syncthreads();
  // Ordered mode (round 5: a.order != nullptr, what the training step uses).  With one atomicAdd per workgroup (the legacy mode
  // below) the base of a workgroup's range is whatever the arrival order made it: the samples of a batch land in a different order on every run -- and, at a full
  // budget, a different set of rays is refused -- so nothing downstream (f32 weight-gradient sums over sample tiles) repeats
  // bit for bit.  Here workgroup b's base is the sum of the counts of workgroups 0 .. b-1: every workgroup publishes its count
  // (flag in bit 63), the first wave reads its predecessors' words 64 at a time, spinning on the ones not yet there (a workgroup
  // only ever waits for LOWER tickets, whose holders are running and wait for nobody above them); the last workgroup through
  // clears the words.  The counts are ready at about the same time, so the look-back is a couple of L2 round trips.
  __shared__ int s_prefix;
  if (a.order != nullptr) {
    const int nwg = (R + 15) >> 4;
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      int tot = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) tot += row_n[k];
      if (lane == 0)
        __hip_atomic_store(&a.order[2 + vb], (1ull << 63) | (unsigned long long)(unsigned)tot, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
      int sum = 0;
      for (int j0 = 0; j0 < vb; j0 += 64) {
        const int j = j0 + lane;
        unsigned long long v = 1ull << 63;
        if (j < vb) {
          do {
            v = __hip_atomic_load(&a.order[2 + j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          } while (!(v >> 63));
        }
        sum += (int)(v & 0x7fffffffull);
      }
      sum = wave_sum_i(sum);
      if (lane == 0) s_prefix = sum;
      // every predecessor's word has been read: count this workgroup out; the last one out of the whole launch zeroes the words
      // and the counters for the next launch
      unsigned long long out = 0;
      if (lane == 0) out = atomicAdd(&a.order[0], 1ull);
      out = __shfl(out, 0, 64);
      if (out == (unsigned long long)(gridDim.x - 1)) {
        for (int j = lane; j < nwg; j += 64) a.order[2 + j] = 0ull;
        if (lane == 0) {
          a.order[0] = 0ull;
          a.order[1] = 0ull;
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) tot += row_n[k];
    int b;
    if (a.order != nullptr) {
      b = s_prefix;
      if (tot > 0) atomicAdd(&a.counter[0], tot);      // (the total requested: a sum, whatever the order)
    } else {
      b = tot > 0 ? atomicAdd(&a.counter[0], tot) : 0;
    }
    int accepted = 0, end = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int nk = row_n[k];
      row_base[k] = b;
      if (nk > 0) {
        if ((long)b + nk > a.max_samples) {
          row_n[k] = -1;  // batch is full: the ray is refused (not part of this batch)
        } else {
          accepted++;
          end = b + nk;
        }
      }
      b += nk;
    }
    if (accepted > 0) {
      atomicAdd(&a.counter[1], accepted);
      atomicMax(&a.counter[2], end);
    }
  }
  __syncthreads();
  n = row_n[row];
  int base = row_base[row];
  if (sub == 0 && live) {
    a.ray_start[r] = base;
    a.ray_n[r] = n;
  }
  if (n <= 0) return;  // row-uniform
  // pass 2: write.  A lane's samples are scattered over the ray's range, and 8 four-byte stores per sample with 64
  // unrelated addresses per instruction kept the texture addresser busy (144 us for ~5000 rays).  So the (t, dt) of
  // the row's samples are first compacted into an LDS window in ray order (same wave: no barrier needed), and the 16
  // lanes of the row then write 16 CONSECUTIVE samples per instruction.
  __shared__ float2 sbuf[16][MARCH_WIN];
  float2* win = sbuf[threadIdx.x >> 4];
  int off = 0;
  tc = tb;
  for (int round = 0; round < rounds; round++) {
    float tscan = tc;
    const uint64_t m = round == 0 ? mask0 : march_scan(a, ry, tscan);
    const int c = __popcll(m);
    const int incl = row_inclusive_sum(c, sub);
    const int k0 = off + incl - c;  // index of this lane's first sample within the ray
    const int round_end = min(off + __shfl(incl, 15, 16), n);
    for (int w0 = off; w0 < round_end; w0 += MARCH_WIN) {
      float t = tc;
      int k = k0;
      for (int q = 0; q < MARCH_SPL && k < w0 + MARCH_WIN && k < n && (m >> q) != 0; q++) {
        const float dt = march_dt(t, a);
        if ((m >> q) & 1) {
          if (k >= w0) win[k - w0] = make_float2(t, dt);
          k++;
        }
        t += dt;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row lives in one wave: its LDS writes are now visible
      const int cnt = min(MARCH_WIN, round_end - w0);
      for (int i = sub; i < cnt; i += 16) {
        const float2 td = win[i];
        const long s = (long)base + w0 + i;
        a.pos[s * 3] = (__fmaf_rn(td.x, ry.dx, ry.ox) - a.pos_lo) * a.pos_inv;
        a.pos[s * 3 + 1] = (__fmaf_rn(td.x, ry.dy, ry.oy) - a.pos_lo) * a.pos_inv;
        a.pos[s * 3 + 2] = (__fmaf_rn(td.x, ry.dz, ry.oz) - a.pos_lo) * a.pos_inv;
        a.dirs[s * 3] = ry.dx;
        a.dirs[s * 3 + 1] = ry.dy;
        a.dirs[s * 3 + 2] = ry.dz;
        a.dt[s] = td.y;
        a.tmid[s] = td.x;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // window consumed before it is refilled
    }
    off += __shfl(incl, 15, 16);
    if (round + 1 < rounds) tc = march_advance(tc, MARCH_ROUND, ry.t1, a);
  }
}

// ---------------------------------------------------------------------------------------------
// Volume rendering + loss + gradients w.r.t. the network outputs; one lane per ray, two passes.
// net_out [S,4] f16 = (r,g,b raw, log-density).  pos is given in the unit cube; the MLP reads
// warped positions, the compositing only needs dt and the distances.
// ---------------------------------------------------------------------------------------------
struct CompositeArgs {
  const _Float16* net_out;  // [S,4]
  const float* dt;
  const float* tmid;
  const int* ray_start;
  const int* ray_n;
  const float* gt_rgb;        // [R,3]
  const float* gt_depth;      // [R]  (<= 0: no depth supervision for this ray)
  const float* gt_depth_cov;  // [R]
  float depth_lambda, loss_scale;
  float* out_rgb;    // [R,3]
  float* out_depth;  // [R]
  float* loss;       // [1] accumulated sum over rays of the per-ray loss (caller zeroes, divides by R)
  float* ray_loss;   // [R] or null: per-ray loss WRITTEN here instead (0 for rays beyond the batch); `loss` is not touched
  _Float16* dLdout;  // [S,4] or null (inference)
  int R;
  const int* ctl;
};

// wave64 inclusive scans on the VALU (DPP row shifts + the two row broadcasts): `__shfl_up` lowers to ds_bpermute_b32 (~100
// cycles per step, six dependent steps per scan, five scans per round of the backward pass).  A lane without a source in
// its row keeps `old` = the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v, float identity) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_sum(float v, int) {
  v += dpp_take<0x111, 0xf>(v, 0.0f);  // row_shr:1
  v += dpp_take<0x112, 0xf>(v, 0.0f);  // row_shr:2
  v += dpp_take<0x114, 0xf>(v, 0.0f);  // row_shr:4
  v += dpp_take<0x118, 0xf>(v, 0.0f);  // row_shr:8  -> scan within every 16-lane row
  v += dpp_take<0x142, 0xa>(v, 0.0f);  // row_bcast15 into rows 1, 3
  v += dpp_take<0x143, 0xc>(v, 0.0f);  // row_bcast31 into rows 2, 3
  return v;
}
__device__ __forceinline__ float wave_incl_prod(float v, int) {
  v *= dpp_take<0x111, 0xf>(v, 1.0f);
  v *= dpp_take<0x112, 0xf>(v, 1.0f);
  v *= dpp_take<0x114, 0xf>(v, 1.0f);
  v *= dpp_take<0x118, 0xf>(v, 1.0f);
  v *= dpp_take<0x142, 0xa>(v, 1.0f);
  v *= dpp_take<0x143, 0xc>(v, 1.0f);
  return v;
}
__device__ __forceinline__ float wave_prev(float v, float first) { return dpp_take<0x138, 0xf>(v, first); }   // wave_shr:1
__device__ __forceinline__ float wave_last(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct CompSample {
  float sigma, alpha, c0, c1, c2, tm, dt;
};

__device__ __forceinline__ CompSample comp_load(const CompositeArgs& a, long s, bool valid) {
  CompSample q{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (valid) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 o = *reinterpret_cast<const f16x4*>(a.net_out + s * 4);
    q.dt = a.dt[s];
    q.tm = a.tmid[s];
    q.sigma = __expf((float)o[3]);
    q.alpha = 1.0f - __expf(-q.sigma * q.dt);
    q.c0 = 1.0f / (1.0f + __expf(-(float)o[0]));
    q.c1 = 1.0f / (1.0f + __expf(-(float)o[1]));
    q.c2 = 1.0f / (1.0f + __expf(-(float)o[2]));
  }
  return q;
}

// One wave per ray, 64 samples per round: transmittance = exclusive prefix product over the lanes, the
// suffix sums of the backward pass = totals minus inclusive prefix sums (wave scans), carried across
// rounds in wave-uniform registers.  (The one-lane-per-ray version walked up to 1024 samples serially:
// 157 us for ~2500 live rays.)
// The kernel's time is its LONGEST ray's (every ray has its wave resident from the start), and that ray's time was one
// exposed load latency per round and pass (load -> scans -> next load ...: 42 us with rays of 1024 samples in the batch).
// Now the first CP_NC rounds (256 samples: all but a few rays) are loaded up front and stay in registers for the backward
// pass; rounds beyond that are loaded one round ahead of their scans.
#define CP_NC 4
__global__ __launch_bounds__(256) void ngp_composite_kernel(CompositeArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int R = a.ctl ? min(a.ctl[NS_CTL_RAYS], a.R) : a.R;
  if (r >= R) {        // wave-uniform
    if (a.ray_loss != nullptr && lane == 0 && r < a.R) a.ray_loss[r] = 0.0f;
    return;
  }
  const int s0 = a.ray_start[r], n = a.ray_n[r];
  CompSample cache[CP_NC];
#pragma unroll
  for (int i = 0; i < CP_NC; i++) cache[i] = comp_load(a, (long)s0 + 64 * i + lane, 64 * i + lane < n);
  float Tin = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, D = 0.0f;
  auto fwd = [&](const CompSample& q) __attribute__((always_inline)) {
    const float incl = wave_incl_prod(1.0f - q.alpha, lane);
    const float excl = wave_prev(incl, 1.0f);
    const float wgt = q.alpha * Tin * excl;
    C0 += wave_sum(wgt * q.c0);
    C1 += wave_sum(wgt * q.c1);
    C2 += wave_sum(wgt * q.c2);
    D += wave_sum(wgt * q.tm);
    Tin *= wave_last(incl);
  };
#pragma unroll
  for (int i = 0; i < CP_NC; i++)
    if (64 * i < n) fwd(cache[i]);                                   // (wave-uniform)
  if (64 * CP_NC < n) {
    CompSample cur = comp_load(a, (long)s0 + 64 * CP_NC + lane, 64 * CP_NC + lane < n);
    for (int k0 = 64 * CP_NC; k0 < n; k0 += 64) {
      const CompSample nxt = comp_load(a, (long)s0 + k0 + 64 + lane, k0 + 64 + lane < n);
      fwd(cur);
      cur = nxt;
    }
  }
  if (lane == 0) {
    a.out_rgb[r * 3] = C0;
    a.out_rgb[r * 3 + 1] = C1;
    a.out_rgb[r * 3 + 2] = C2;
    a.out_depth[r] = D;
  }
  if (a.dLdout == nullptr) return;
  const float e0 = C0 - a.gt_rgb[r * 3], e1 = C1 - a.gt_rgb[r * 3 + 1], e2 = C2 - a.gt_rgb[r * 3 + 2];
  float l = n < 0 ? 0.0f : (e0 * e0 + e1 * e1 + e2 * e2) / 3.0f;  // n < 0: refused by the marcher, not in the batch
  const float dC0 = 2.0f * e0 / 3.0f, dC1 = 2.0f * e1 / 3.0f, dC2 = 2.0f * e2 / 3.0f;
  float dD = 0.0f;
  const float gd = a.gt_depth[r];
  if (n >= 0 && gd > 0.0f && a.depth_lambda > 0.0f) {
    const float ed = D - gd, icov = 1.0f / a.gt_depth_cov[r];
    l += a.depth_lambda * ed * ed * icov;
    dD = a.depth_lambda * 2.0f * ed * icov;
  }
  // (one atomicAdd per ray on ONE address: ~4000 same-address atomics serialise in the L2 at ~10 ns each -- 36 of this kernel's
  //  46 us, on the step's critical path.  The trainer takes the per-ray values and sums them when somebody asks for the loss.)
  if (lane == 0) {
    if (a.ray_loss != nullptr) a.ray_loss[r] = l;
    else atomicAdd(a.loss, l);
  }
  const float sc = a.loss_scale / (float)R;
  Tin = 1.0f;
  float P0 = 0.0f, P1 = 0.0f, P2 = 0.0f, PD = 0.0f;  // prefix sums carried across rounds
  auto bwd = [&](const CompSample& q, int k0) __attribute__((always_inline)) {
    const bool valid = k0 + lane < n;
    const long s = (long)s0 + k0 + lane;
    const float incl = wave_incl_prod(1.0f - q.alpha, lane);
    const float excl = wave_prev(incl, 1.0f);
    const float T = Tin * excl, wgt = q.alpha * T, Tn = Tin * incl;
    const float p0 = P0 + wave_incl_sum(wgt * q.c0, lane), p1 = P1 + wave_incl_sum(wgt * q.c1, lane);
    const float p2 = P2 + wave_incl_sum(wgt * q.c2, lane), pd = PD + wave_incl_sum(wgt * q.tm, lane);
    if (valid) {
      const float gsum = dC0 * (Tn * q.c0 - (C0 - p0)) + dC1 * (Tn * q.c1 - (C1 - p1)) + dC2 * (Tn * q.c2 - (C2 - p2)) +
                         dD * (Tn * q.tm - (D - pd));
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 g = {(_Float16)(sc * wgt * dC0 * q.c0 * (1.0f - q.c0)), (_Float16)(sc * wgt * dC1 * q.c1 * (1.0f - q.c1)),
                       (_Float16)(sc * wgt * dC2 * q.c2 * (1.0f - q.c2)), (_Float16)(sc * q.dt * gsum * q.sigma)};
      *reinterpret_cast<f16x4*>(a.dLdout + s * 4) = g;
    }
    P0 = wave_last(p0);
    P1 = wave_last(p1);
    P2 = wave_last(p2);
    PD = wave_last(pd);
    Tin *= wave_last(incl);
  };
#pragma unroll
  for (int i = 0; i < CP_NC; i++)
    if (64 * i < n) bwd(cache[i], 64 * i);
  if (64 * CP_NC < n) {
    CompSample cur = comp_load(a, (long)s0 + 64 * CP_NC + lane, 64 * CP_NC + lane < n);
    for (int k0 = 64 * CP_NC; k0 < n; k0 += 64) {
      const CompSample nxt = comp_load(a, (long)s0 + k0 + 64 + lane, k0 + 64 + lane < n);
      bwd(cur, k0);
      cur = nxt;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int ns_ngp_grid_layout(int n_levels, int n_features, int log2_hashmap, int base_res,
                                  float per_level_scale, float* scale_host, int* res_host, uint32_t* offset_host) {
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_grid_layout: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  for (int l = 0; l < n_levels; l++) {
    if (scale_host) scale_host[l] = g.scale[l];
    if (res_host) res_host[l] = g.res[l];
    if (offset_host) offset_host[l] = g.offset[l];
  }
  if (offset_host) offset_host[n_levels] = g.offset[n_levels];
  return NS_OK;
}

extern "C" int ns_ngp_encode_forward(int n_levels, int n_features, int log2_hashmap, int base_res,
                                     float per_level_scale, const float* positions, const void* params, void* out,
                                     int unit_major, long N, void* stream) {
  return ns_ngp_encode_forward_n(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, params, out, unit_major,
                                 N, nullptr, stream);
}

extern "C" int ns_ngp_encode_forward_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                     float per_level_scale, const float* positions, const void* params, void* out,
                                     int unit_major, long N, const int* n_dev, void* stream) {
  return ns_ngp_encode_forward_j_n(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, params, out, unit_major,
                                   nullptr, N, n_dev, stream);
}

extern "C" int ns_ngp_encode_forward_j_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                       float per_level_scale, const float* positions, const void* params, void* out,
                                       int unit_major, void* jacT, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(positions && params && out, "ns_ngp_encode_forward: null pointer");
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_encode_forward: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  if (N <= 0) return NS_OK;
  EncFwdSched sc;
  {
    // the (up to) 8 finest hashed levels are owned by one XCD each; everything else is shared, hashed levels first
    int hashed[16], nh = 0;
    for (int l = n_levels - 1; l >= 0; l--)
      if ((uint64_t)g.res[l] * g.res[l] * g.res[l] > (uint64_t)(g.offset[l + 1] - g.offset[l])) hashed[nh++] = l;   // finest first
    static const bool no_own = ns_variant_env("NS_ENC_FWD_NO_XCD") != nullptr;   // A/B: every level shared (round 2's access pattern)
    const int n_own = no_own ? 0 : (nh < 8 ? nh : 8);
    for (int x = 0; x < 8; x++) sc.own[x] = x < n_own ? hashed[x] : -1;
    sc.n_shared = 0;
    for (int k = n_own; k < nh; k++) sc.shared[sc.n_shared++] = hashed[k];
    for (int l = n_levels - 1; l >= 0; l--)
      if (!((uint64_t)g.res[l] * g.res[l] * g.res[l] > (uint64_t)(g.offset[l + 1] - g.offset[l]))) sc.shared[sc.n_shared++] = l;
    for (int k = sc.n_shared; k < 16; k++) sc.shared[k] = 0;
    sc.nblk = ns_cdiv(N, 256);
    sc.nblk8 = (sc.nblk + 7) / 8;
    sc.seg = n_own > 0 ? sc.nblk : 0;
  }
  const long nwg = 8L * (sc.seg + (long)sc.n_shared * sc.nblk8);
  hipLaunchKernelGGL(ngp_encode_fwd_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g,
                     positions, (const h2_t*)params, (h2_t*)out, N, n_levels, unit_major, n_dev, (_Float16*)jacT, sc);
  NS_CHECK_LAUNCH("ngp_encode_fwd_kernel");
  return NS_OK;
}

// dL/dpos [N,3] = sum over levels of scale_l * J_l^T dL/dfeature_l, J from ns_ngp_encode_forward_j_n
extern "C" int ns_ngp_encode_jacobian_dot_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                          const void* jacT, const void* dLdoutT, float* dLdpos, long N, const int* n_dev,
                                          void* stream) {
  NS_REQUIRE(jacT && dLdoutT && dLdpos, "ns_ngp_encode_jacobian_dot: null pointer");
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_encode_jacobian_dot: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  if (N <= 0) return NS_OK;
  hipLaunchKernelGGL(ngp_encode_jac_dot_kernel, dim3(ns_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, g, (const _Float16*)jacT,
                     (const _Float16*)dLdoutT, dLdpos, N, n_levels, n_dev);
  NS_CHECK_LAUNCH("ngp_encode_jac_dot_kernel");
  return NS_OK;
}

// sum the replicas into the gradient and clear them (only entries that were touched are written back)
#ifdef NS_TEST_VARIANTS   // comparison kernel: libnerfslam_hip_variants.so only (common.h)
__global__ __launch_bounds__(256) void ngp_encode_bwd_reduce_kernel(GridLayout g, ReplicaPlan rp, int n_levels,
                                                                    float* __restrict__ ws, float* __restrict__ grad,
                                                                    float fixed_scale) {
  const int l = blockIdx.y;
  const uint32_t rep = rp.rep[l];
  if (rep <= 1) return;
  const uint64_t n = (uint64_t)(g.offset[l + 1] - g.offset[l]);  // entries (two floats each)
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) {
    float s0 = 0.0f, s1 = 0.0f;
    float2* p = reinterpret_cast<float2*>(ws + rp.ws_off[l]) + e;
    for (uint32_t k = 0; k < rep; k++) {
      const float2 v = p[(uint64_t)k * n];
      if (v.x != 0.0f || v.y != 0.0f) {
        s0 += v.x;
        s1 += v.y;
        p[(uint64_t)k * n] = make_float2(0.0f, 0.0f);
      }
    }
    if (s0 != 0.0f || s1 != 0.0f) {
      float* gp = grad + ((uint64_t)g.offset[l] + e) * 2;
      if (fixed_scale > 0.0f) {
        *reinterpret_cast<unsigned long long*>(gp) += pack_fixed(s0, s1, fixed_scale);
      } else {
        gp[0] += s0;
        gp[1] += s1;
      }
    }
  }
}
#endif  // NS_TEST_VARIANTS

extern "C" long ns_ngp_encode_backward_workspace_bytes(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                       float per_level_scale, long max_samples) {
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) return -1;
  ReplicaPlan rp;
  replica_plan_host(g, n_levels, rp);
  BinPlan bp;
  bin_plan_host(g, n_levels, max_samples > 0 ? max_samples : 1, bp);
  const size_t bin = bin_plan_ok(bp) ? bin_ws_bytes(bp, g, n_levels) : 0;
  const size_t rep_b = rp.total_floats * sizeof(float);   // (round-1 atomic path, NS_ENC_BWD_ATOMIC)
  return (long)(bin > rep_b ? bin : rep_b);
}

extern "C" int ns_ngp_encode_backward(int n_levels, int n_features, int log2_hashmap, int base_res,
                                      float per_level_scale, const float* positions, const void* dLdout,
                                      int unit_major, float* grad_params, float* workspace, size_t workspace_bytes,
                                      float fixed_scale, long N, void* stream) {
  return ns_ngp_encode_backward_n(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, dLdout, unit_major,
                                  grad_params, workspace, workspace_bytes, fixed_scale, N, nullptr, stream);
}

extern "C" int ns_ngp_encode_backward_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                      float per_level_scale, const float* positions, const void* dLdout,
                                      int unit_major, float* grad_params, float* workspace, size_t workspace_bytes,
                                      float fixed_scale, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(positions && dLdout && grad_params, "ns_ngp_encode_backward: null pointer");
  // (the workspace layout follows from THIS call's N; a buffer sized for fewer samples must not be written past
  //  its end -- such a call takes the owner-computes kernels, which need no workspace)
  if (workspace == nullptr) workspace_bytes = 0;
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_encode_backward: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  if (N <= 0) return NS_OK;
  static const bool atomic_path = ns_variant_env("NS_ENC_BWD_ATOMIC") != nullptr;  // round-1 kernel, kept for A/B profiling
  if (!atomic_path) {
    static const bool no_bins = ns_variant_env("NS_ENC_BWD_NO_BINS") != nullptr;   // A/B switch: owner-computes kernel on every level
    BinPlan bp;
    bin_plan_host(g, n_levels, N, bp);
    const bool binned = workspace != nullptr && fixed_scale > 0.0f && bin_plan_ok(bp) && !no_bins &&
                        workspace_bytes >= bin_ws_bytes(bp, g, n_levels);
    EncBwdPlan plan;
    // run-length kernel for the dense levels (NS_ENC_BWD_NO_RL=1: the one-sample-per-lane kernel, A/B runs); fewer, longer
    // parts: its tasks are bound by the scan of the samples, not by LDS atomics (NS_ENC_RL_PARTS=coarse,multi overrides)
    static const bool no_rl = ns_variant_env("NS_ENC_BWD_NO_RL") != nullptr;
    static int rl_pc = 16, rl_pm = 8;
    static const bool rl_env = [] {
      const char* e = ns_variant_env("NS_ENC_RL_PARTS");
      int a = 0, b = 0;
      if (e && sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= NS_ENC_PARTS_COARSE && b >= 1 && b <= NS_ENC_PARTS_COARSE) {
        rl_pc = a;
        rl_pm = b;
      }
      return true;
    }();
    (void)rl_env;
    const bool rl = binned && !no_rl && unit_major && N % 8 == 0 && ((uintptr_t)positions % 16) == 0 && ((uintptr_t)dLdout % 16) == 0;
    const int tasks = rl ? enc_bwd_plan_host(g, n_levels, plan, true, rl_pc, rl_pm) : enc_bwd_plan_host(g, n_levels, plan, binned);
    const int blocks = (tasks + 7) / 8 * 8;
    if (binned) {
      int* tot = reinterpret_cast<int*>(workspace);
      int2* cnt = reinterpret_cast<int2*>(reinterpret_cast<char*>(workspace) + bin_ws_tot_bytes(bp));
      unsigned long long* queue = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) +
                                                                        bin_ws_tot_bytes(bp) + bin_ws_cnt_bytes(bp));
      // (a kernel, not hipMemsetAsync: a memset node in the captured training step faulted on its second replay, ROCm 7.2)
      hipLaunchKernelGGL(ngp_zero_ints_kernel, dim3(ns_cdiv(bp.nh * NS_BIN_MAX, 256)), dim3(256), 0, (hipStream_t)stream, tot,
                         bp.nh * NS_BIN_MAX);
      NS_CHECK_LAUNCH("ngp_zero_ints_kernel");
      hipLaunchKernelGGL(ngp_enc_bin_count_kernel, dim3(bp.ntiles, bp.nh), dim3(256), 0, (hipStream_t)stream, g, bp, positions,
                         (const h2_t*)dLdout, N, n_levels, unit_major, tot, cnt, n_dev);
      NS_CHECK_LAUNCH("ngp_enc_bin_count_kernel");
      hipLaunchKernelGGL(ngp_enc_bin_scatter_kernel, dim3(bp.ntiles, bp.nh), dim3(256), 0, (hipStream_t)stream, g, bp, positions,
                         (const h2_t*)dLdout, N, n_levels, unit_major, fixed_scale, tot, cnt, queue, n_dev);
      NS_CHECK_LAUNCH("ngp_enc_bin_scatter_kernel");
      hipLaunchKernelGGL(ngp_enc_bin_accum_kernel, dim3(NS_BIN_MAX, bp.nh), dim3(1024), 0, (hipStream_t)stream, g, bp, tot, queue,
                         grad_params);
      NS_CHECK_LAUNCH("ngp_enc_bin_accum_kernel");
      if (tasks == 0) return NS_OK;
      const long nd = dense_prefix_entries(g, n_levels);
      if (nd > 0) {
        unsigned long long* partial = queue + bin_ws_queue_bytes(bp) / 8;
        if (rl) {
          hipLaunchKernelGGL((ngp_encode_bwd_dense_rl_kernel<NS_ENC_SLICE, 1024>), dim3(blocks), dim3(1024), 0, (hipStream_t)stream, g, plan, positions,
                             (const _Float16*)dLdout, grad_params, N, fixed_scale, partial, n_dev);
          NS_CHECK_LAUNCH("ngp_encode_bwd_dense_rl_kernel");
        } else {
          hipLaunchKernelGGL(ngp_encode_bwd_lds_kernel<true>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, g, plan, positions,
                             (const h2_t*)dLdout, grad_params, N, n_levels, n_levels, unit_major, fixed_scale, partial, n_dev);
          NS_CHECK_LAUNCH("ngp_encode_bwd_lds_kernel");
        }
        AdamFuse no_adam{};
        hipLaunchKernelGGL(ngp_enc_dense_reduce_kernel, dim3(ns_cdiv(nd, 256)), dim3(256), 0, (hipStream_t)stream, g, plan, n_levels,
                           partial, nd, grad_params, no_adam, (int*)nullptr);
        NS_CHECK_LAUNCH("ngp_enc_dense_reduce_kernel");
        return NS_OK;
      }
    }
    if (fixed_scale > 0.0f)
      hipLaunchKernelGGL(ngp_encode_bwd_lds_kernel<true>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, g, plan, positions,
                         (const h2_t*)dLdout, grad_params, N, n_levels, n_levels, unit_major, fixed_scale,
                         (unsigned long long*)nullptr, n_dev);
    else
      hipLaunchKernelGGL(ngp_encode_bwd_lds_kernel<false>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, g, plan, positions,
                         (const h2_t*)dLdout, grad_params, N, n_levels, n_levels, unit_major, fixed_scale,
                         (unsigned long long*)nullptr, n_dev);
    NS_CHECK_LAUNCH("ngp_encode_bwd_lds_kernel");
    return NS_OK;
  }
#ifdef NS_TEST_VARIANTS    // round 1's atomic scatter (NS_ENC_BWD_ATOMIC): comparison path, variants library only
  ReplicaPlan rp;
  replica_plan_host(g, n_levels, rp);
  if (workspace_bytes < rp.total_floats * sizeof(float)) workspace = nullptr;   // (then: global atomics on every level)
  hipLaunchKernelGGL(ngp_encode_bwd_kernel, dim3(ns_cdiv(N, 256), n_levels), dim3(256), 0, (hipStream_t)stream, g,
                     positions, (const h2_t*)dLdout, grad_params, N, n_levels, 0, rp, workspace, unit_major, fixed_scale);
  NS_CHECK_LAUNCH("ngp_encode_bwd_kernel");
  if (workspace != nullptr && rp.total_floats > 0) {
    hipLaunchKernelGGL(ngp_encode_bwd_reduce_kernel, dim3(256, n_levels), dim3(256), 0, (hipStream_t)stream, g, rp,
                       n_levels, workspace, grad_params, fixed_scale);
    NS_CHECK_LAUNCH("ngp_encode_bwd_reduce_kernel");
  }
#endif
  return NS_OK;
}

extern "C" size_t ns_ngp_encode_backward_fused_workspace_bytes(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                               float per_level_scale, long max_samples) {
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) return 0;
  FusedPlan fp;
  if (!fused_plan_host(g, n_levels, max_samples > 0 ? max_samples : 1, fp, 2)) return 0;   // (the largest of the three plans)
  return fused_ws_bytes(fp, g, n_levels);
}

extern "C" int ns_ngp_encode_backward_fused_dense_levels(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                         float per_level_scale) {
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) return 0;
  return fused_dense_levels(g, n_levels, fused_dense_mode());
}

static int fused_backward_impl(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                               const float* positions, const void* dLdoutT, float* grad_params, void* workspace,
                               size_t workspace_bytes, float fixed_scale, long N, const int* n_dev, float* master,
                               void* half_params, float* m1, float* m2, int step, float lr, float beta1, float beta2, float eps,
                               float grad_scale, const int* ctl, int parts, void* stream, ulonglong2* emit_list, int* emit_count,
                               int record_floats = 0) {
  NS_REQUIRE(positions && dLdoutT && workspace, "ns_ngp_encode_backward_fused: null pointer");
  NS_REQUIRE(parts >= 1 && parts <= 15, "ns_ngp_encode_backward_fused: parts is a mask of 1 | 2 | 4 | 8");
  NS_REQUIRE(fixed_scale > 0.0f, "ns_ngp_encode_backward_fused: packed fixed-point sums only (fixed_scale > 0)");
  const bool adam = master != nullptr;
  NS_REQUIRE(adam || grad_params || emit_list, "ns_ngp_encode_backward_fused: neither a gradient buffer nor Adam state nor a list");
  NS_REQUIRE(!emit_list || (emit_count && !adam && !(parts & 12)), "ns_ngp_encode_backward_fused: the list form is parts 1 | 2 without Adam");
  NS_REQUIRE(!adam || (half_params && m1 && m2 && grad_scale > 0.0f && (ctl || step >= 1)),
             "ns_ngp_encode_backward_fused: incomplete Adam state");
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_encode_backward_fused: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  if (N <= 0) return NS_OK;
  // NS_ENC_DENSE_BINNED=1: the multi-slice dense levels (30^3, 42^3, 58^3 entries) go through the bins too (per-level slot sizes,
  // dense indices in the scatter).  Measured: the training step 0.372 -> 0.365 ms (192 -> 32 whole-CU owner-computes workgroups),
  // but on the all-live micro-bench the accumulate pass 135 -> 301 us -- a dense level has 4 / 10 / 26 bins for the records that a
  // hashed level spreads over 64, and every sample of a ray hits the same few cells of it: long, conflict-ridden bins set the
  // kernel's time.  Off by default; kept for A/B (sums bit-identical either way: test_fused_table_gradient_matches_the_other_paths).
  const int dense_too = fused_dense_mode();
  FusedPlan fp;
  if (!fused_plan_host(g, n_levels, N, fp, dense_too)) {
    ns_set_error("ns_ngp_encode_backward_fused: no hashed level / tables above 64 x 8192 entries: use ns_ngp_encode_backward");
    return NS_ENOSUP;
  }
  const size_t need = fused_ws_bytes(fp, g, n_levels);
  NS_REQUIRE(workspace_bytes >= need, "ns_ngp_encode_backward_fused: workspace of %zu B, N=%ld needs %zu", workspace_bytes, N, need);
  hipStream_t st = (hipStream_t)stream;
  char* ws = reinterpret_cast<char*>(workspace);
  int* ctr = reinterpret_cast<int*>(ws);
  int* cnt = reinterpret_cast<int*>(ws + fused_ws_ctr_bytes(fp));
  unsigned long long* queue = reinterpret_cast<unsigned long long*>(ws + fused_ws_ctr_bytes(fp) + fused_ws_cnt_bytes(fp));
  ulonglong2* ovf = reinterpret_cast<ulonglong2*>(ws + fused_ws_ctr_bytes(fp) + fused_ws_cnt_bytes(fp) + fused_ws_queue_bytes(fp));
  unsigned long long* partial = reinterpret_cast<unsigned long long*>(ws + fused_ws_ctr_bytes(fp) + fused_ws_cnt_bytes(fp) +
                                                                      fused_ws_queue_bytes(fp) + fused_ws_ovf_bytes(fp));
  AdamFuse ad{};
  if (adam) {
    ad.master = master;
    ad.hp = (_Float16*)half_params;
    ad.m1 = m1;
    ad.m2 = m2;
    ad.es = adam_entry_stride(master, m1, m2, record_floats);
    NS_REQUIRE(ad.es > 0, "ns_ngp_encode_backward_fused: record_floats = %d does not describe master / m1 / m2 (6 or 8: m1 == master + 2, "
               "m2 == master + 4 floats, base 8- / 16-byte aligned; 2: three arrays)", record_floats);
    NS_REQUIRE(ad.es != 8 || ((uintptr_t)master & 15) == 0,
               "ns_ngp_encode_backward_fused: interleaved optimiser records need a 16-byte aligned base (float4 accesses)");
    ad.c1 = 1.0f - powf(beta1, (float)(step < 1 ? 1 : step));
    ad.c2 = 1.0f - powf(beta2, (float)(step < 1 ? 1 : step));
    ad.lr = lr;
    ad.beta1 = beta1;
    ad.beta2 = beta2;
    ad.eps = eps;
    ad.inv_grad_scale = 1.0f / grad_scale;
    ad.inv_fixed_scale = 1.0f / fixed_scale;
    ad.ctl = ctl;
  }
  ad.emit_list = emit_list;
  ad.emit_count = emit_count;
  NS_REQUIRE(dense_prefix_entries(g, n_levels) >= 0, "ns_ngp_encode_backward_fused: dense levels above hashed ones");
  NS_REQUIRE(!emit_list || fused_dense_levels(g, n_levels, dense_too) == 0,
             "ns_ngp_encode_backward_fused: the list form needs every level on the binned path");
  const int n_rl = fused_dense_levels(g, n_levels, dense_too);     // dense levels that keep the owner-computes path
  const long nd = (long)g.offset[n_rl];
  if (parts & 1) {
    const int vec = (N % 4 == 0 && ((uintptr_t)positions % 16) == 0 && ((uintptr_t)dLdoutT % 8) == 0) ? 1 : 0;
    bool staged = false;
#ifdef NS_TEST_VARIANTS
    static const bool staged_env = [] { const char* e = ns_variant_env("NS_FB_SCATTER"); return e != nullptr && e[0] == '1'; }();
    staged = staged_env;
    if (staged) {
      hipLaunchKernelGGL(ngp_enc_fscatter_kernel, dim3(fp.ntiles, fp.nh), dim3(256), 0, st, g, fp, positions, (const _Float16*)dLdoutT,
                         N, fixed_scale, ctr, cnt, queue, ovf, n_dev, vec);
      NS_CHECK_LAUNCH("ngp_enc_fscatter_kernel");
    }
#endif
    if (!staged) {
      hipLaunchKernelGGL(ngp_enc_fscatter_direct_kernel, dim3(fp.ntiles, fp.nh), dim3(256), 0, st, g, fp, positions,
                         (const _Float16*)dLdoutT, N, fixed_scale, ctr, cnt, queue, ovf, n_dev, vec);
      NS_CHECK_LAUNCH("ngp_enc_fscatter_direct_kernel");
    }
  }
  if (parts & 2) {
    int n_groups = 0;
    for (int k = 0; k < fp.nh; k++) n_groups += fp.nbins[k];
    hipLaunchKernelGGL(ngp_enc_faccum_kernel, dim3(NS_FB_BINS, fp.nh), dim3(NS_FB_THREADS), 0, st, g, fp, ctr, cnt, queue, ovf,
                       grad_params, ad, N, n_dev, ctr, n_groups);
    NS_CHECK_LAUNCH("ngp_enc_faccum_kernel");
  }
  if (!(parts & 12) || nd == 0) return NS_OK;
  EncBwdPlan plan;
  const bool rl = N % 8 == 0 && ((uintptr_t)positions % 16) == 0 && ((uintptr_t)dLdoutT % 16) == 0;
  static int rl_pc = 16, rl_pm = 8;      // parts of the single-slice / multi-slice dense levels (NS_ENC_RL_PARTS=coarse,multi)
  static bool rl_set = false;
  static const bool rl_env = [] {
    const char* e = ns_variant_env("NS_ENC_RL_PARTS");
    int a = 0, b = 0;
    if (e && sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= NS_ENC_PARTS_COARSE && b >= 1 && b <= NS_ENC_PARTS_COARSE) {
      rl_pc = a;
      rl_pm = b;
      rl_set = true;
    }
    return true;
  }();
  (void)rl_env;
  // (with the multi-slice levels binned, the two or three single-slice levels are alone on this path: 64 parts each, or their
  //  32 workgroups take 130 us when nothing runs next to them)
  // NS_ENC_RL_HALF=1: 8192-entry slices in 512-thread workgroups (64 KB of LDS: they fit next to a scatter workgroup).  Measured:
  // training step unchanged (0.385-0.389 ms either way), all-live micro-bench 279 -> 311 us (twice the slices scan the samples).
  static const bool half_slices = ns_variant_env("NS_ENC_RL_HALF") != nullptr;
  const int tasks = rl ? enc_bwd_plan_host(g, n_levels, plan, true, dense_too && !rl_set ? NS_ENC_PARTS_COARSE : rl_pc, rl_pm, n_rl,
                                           half_slices ? NS_ENC_SLICE / 2 : NS_ENC_SLICE)
                       : enc_bwd_plan_host(g, n_levels, plan, true, NS_ENC_PARTS_COARSE, NS_ENC_PARTS_BINNED, n_rl);
  const int blocks = (tasks + 7) / 8 * 8;
  // the dense levels always go through their partial planes here (the reduce pass is where Adam is applied)
  if (parts & 4) {
#ifdef NS_TEST_VARIANTS
    if (rl && half_slices) {
      hipLaunchKernelGGL((ngp_encode_bwd_dense_rl_kernel<NS_ENC_SLICE / 2, 512>), dim3(blocks), dim3(512), 0, st, g, plan, positions,
                         (const _Float16*)dLdoutT, grad_params, N, fixed_scale, partial, n_dev);
      NS_CHECK_LAUNCH("ngp_encode_bwd_dense_rl_kernel<half>");
    } else
#endif
    if (rl) {
      hipLaunchKernelGGL((ngp_encode_bwd_dense_rl_kernel<NS_ENC_SLICE, 1024>), dim3(blocks), dim3(1024), 0, st, g, plan, positions,
                         (const _Float16*)dLdoutT, grad_params, N, fixed_scale, partial, n_dev);
      NS_CHECK_LAUNCH("ngp_encode_bwd_dense_rl_kernel");
    } else {
      hipLaunchKernelGGL(ngp_encode_bwd_lds_kernel<true>, dim3(blocks), dim3(1024), 0, st, g, plan, positions,
                         (const h2_t*)dLdoutT, grad_params, N, n_levels, n_levels, 1, fixed_scale, partial, n_dev);
      NS_CHECK_LAUNCH("ngp_encode_bwd_lds_kernel");
    }
  }
  if (!(parts & 8)) return NS_OK;
  hipLaunchKernelGGL(ngp_enc_dense_reduce_kernel, dim3(ns_cdiv(nd, 256)), dim3(256), 0, st, g, plan, n_levels, partial, nd,
                     grad_params, ad, (int*)nullptr);
  NS_CHECK_LAUNCH("ngp_enc_dense_reduce_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_encode_backward_fused_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                              float per_level_scale, const float* positions, const void* dLdoutT,
                                              float* grad_params, void* workspace, size_t workspace_bytes, float fixed_scale,
                                              long N, const int* n_dev, float* master, void* half_params, float* m1,
                                              float* m2, int step, float lr, float beta1, float beta2, float eps,
                                              float grad_scale, const int* ctl, int parts, void* stream) {
  return fused_backward_impl(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, dLdoutT, grad_params, workspace,
                             workspace_bytes, fixed_scale, N, n_dev, master, half_params, m1, m2, step, lr, beta1, beta2, eps,
                             grad_scale, ctl, parts, stream, nullptr, nullptr);
}

extern "C" int ns_ngp_encode_backward_fused_rec_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                  float per_level_scale, const float* positions, const void* dLdoutT,
                                                  float* grad_params, void* workspace, size_t workspace_bytes, float fixed_scale,
                                                  long N, const int* n_dev, float* master, void* half_params, float* m1,
                                                  float* m2, int record_floats, int step, float lr, float beta1, float beta2,
                                                  float eps, float grad_scale, const int* ctl, int parts, void* stream) {
  NS_REQUIRE(record_floats == 2 || record_floats == 6 || record_floats == 8, "ns_ngp_encode_backward_fused_rec: record_floats is 2, 6 or 8");
  return fused_backward_impl(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, dLdoutT, grad_params, workspace,
                             workspace_bytes, fixed_scale, N, n_dev, master, half_params, m1, m2, step, lr, beta1, beta2, eps,
                             grad_scale, ctl, parts, stream, nullptr, nullptr, record_floats);
}

extern "C" int ns_ngp_encode_backward_fused_emit_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                   float per_level_scale, const float* positions, const void* dLdoutT,
                                                   void* workspace, size_t workspace_bytes, float fixed_scale, long N,
                                                   const int* n_dev, void* list, int* list_count, int parts, void* stream) {
  NS_REQUIRE(list && list_count, "ns_ngp_encode_backward_fused_emit: null list");
  NS_REQUIRE(((uintptr_t)list & 15) == 0, "ns_ngp_encode_backward_fused_emit: the list holds 16-byte pairs");
  return fused_backward_impl(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, dLdoutT, nullptr, workspace,
                             workspace_bytes, fixed_scale, N, n_dev, nullptr, nullptr, nullptr, nullptr, 1, 0.0f, 0.0f, 0.0f, 0.0f,
                             1.0f, nullptr, parts, stream, reinterpret_cast<ulonglong2*>(list), list_count);
}

extern "C" int ns_ngp_sparse_table_update(const void* lists, const int* counts, int n_lists, long stride, long max_count, void* acc,
                                          float* master, void* half_params, float* m1, float* m2, int step, float lr, float beta1,
                                          float beta2, float eps, float grad_scale, float fixed_scale, const int* ctl,
                                          void* stream) {
  return ns_ngp_sparse_table_update_rec(lists, counts, n_lists, stride, max_count, acc, master, half_params, m1, m2, 0, step, lr, beta1,
                                        beta2, eps, grad_scale, fixed_scale, ctl, stream);
}

extern "C" int ns_ngp_sparse_table_update_rec(const void* lists, const int* counts, int n_lists, long stride, long max_count, void* acc,
                                              float* master, void* half_params, float* m1, float* m2, int record_floats, int step,
                                              float lr, float beta1, float beta2, float eps, float grad_scale, float fixed_scale,
                                              const int* ctl, void* stream) {
  NS_REQUIRE(lists && counts && acc && master && half_params && m1 && m2, "ns_ngp_sparse_table_update: null pointer");
  NS_REQUIRE(n_lists >= 1 && stride >= max_count && max_count >= 0, "ns_ngp_sparse_table_update: bad sizes");
  NS_REQUIRE(grad_scale > 0.0f && fixed_scale > 0.0f && (ctl || step >= 1), "ns_ngp_sparse_table_update: incomplete Adam state");
  if (max_count == 0) return NS_OK;
  AdamFuse ad{};
  ad.master = master;
  ad.hp = (_Float16*)half_params;
  ad.m1 = m1;
  ad.m2 = m2;
  ad.es = adam_entry_stride(master, m1, m2, record_floats);
  NS_REQUIRE(ad.es > 0, "ns_ngp_sparse_table_update: record_floats = %d does not describe master / m1 / m2", record_floats);
  NS_REQUIRE(ad.es != 8 || ((uintptr_t)master & 15) == 0, "ns_ngp_sparse_table_update: interleaved optimiser records need a 16-byte aligned base");
  ad.c1 = 1.0f - powf(beta1, (float)(step < 1 ? 1 : step));
  ad.c2 = 1.0f - powf(beta2, (float)(step < 1 ? 1 : step));
  ad.lr = lr;
  ad.beta1 = beta1;
  ad.beta2 = beta2;
  ad.eps = eps;
  ad.inv_grad_scale = 1.0f / grad_scale;
  ad.inv_fixed_scale = 1.0f / fixed_scale;
  ad.ctl = ctl;
  hipStream_t st = (hipStream_t)stream;
  const int bx = (int)((max_count + 1023) / 1024 < 4096 ? (max_count + 1023) / 1024 : 4096);   // four pairs per lane and sweep
  const dim3 grid(bx < 1 ? 1 : bx, n_lists);
  hipLaunchKernelGGL(ngp_sparse_accumulate_kernel, grid, dim3(256), 0, st, (const ulonglong2*)lists, counts, stride,
                     (unsigned long long*)acc);
  NS_CHECK_LAUNCH("ngp_sparse_accumulate_kernel");
  hipLaunchKernelGGL(ngp_sparse_apply_kernel, grid, dim3(256), 0, st, (const ulonglong2*)lists, counts, stride,
                     (unsigned long long*)acc, ad);
  NS_CHECK_LAUNCH("ngp_sparse_apply_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_adam(float* master, void* half_params, float* grad, float* m1, float* m2, long n, int step,
                           float lr, float beta1, float beta2, float eps, float l2, float grad_scale,
                           float fixed_scale, void* stream) {
  return ns_ngp_adam_ctl(master, half_params, grad, m1, m2, n, step, lr, beta1, beta2, eps, l2, grad_scale, fixed_scale,
                         nullptr, stream);
}

extern "C" int ns_ngp_adam_ctl(float* master, void* half_params, float* grad, float* m1, float* m2, long n, int step,
                               float lr, float beta1, float beta2, float eps, float l2, float grad_scale,
                               float fixed_scale, const int* ctl, void* stream) {
  return ns_ngp_adam_rec_ctl(master, half_params, grad, m1, m2, 0, n, step, lr, beta1, beta2, eps, l2, grad_scale, fixed_scale, ctl, stream);
}

extern "C" int ns_ngp_adam_rec_ctl(float* master, void* half_params, float* grad, float* m1, float* m2, int record_floats, long n,
                                   int step, float lr, float beta1, float beta2, float eps, float l2, float grad_scale,
                                   float fixed_scale, const int* ctl, void* stream) {
  NS_REQUIRE(master && half_params && grad && m1 && m2, "ns_ngp_adam: null pointer");
  NS_REQUIRE((ctl || step >= 1) && grad_scale > 0.0f, "ns_ngp_adam: step must be >= 1 and grad_scale > 0");
  NS_REQUIRE(fixed_scale == 0.0f || n % 2 == 0, "ns_ngp_adam: packed gradients come in pairs");
  if (n <= 0) return NS_OK;
  const float c1 = 1.0f - powf(beta1, (float)(step < 1 ? 1 : step)), c2 = 1.0f - powf(beta2, (float)(step < 1 ? 1 : step));
  const int es = adam_entry_stride(master, m1, m2, record_floats);
  NS_REQUIRE(es > 0, "ns_ngp_adam: record_floats = %d does not describe master / m1 / m2", record_floats);
  NS_REQUIRE(es == 2 || n % 2 == 0, "ns_ngp_adam: interleaved records hold two parameters each");
  NS_REQUIRE(es != 8 || ((uintptr_t)master & 15) == 0, "ns_ngp_adam: interleaved optimiser records need a 16-byte aligned base");
  hipLaunchKernelGGL(ngp_adam_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, master,
                     (_Float16*)half_params, grad, m1, m2, n, c1, c2, lr, beta1, beta2, eps, l2, 1.0f / grad_scale,
                     fixed_scale > 0.0f ? 1.0f / fixed_scale : 0.0f, ctl, es);
  NS_CHECK_LAUNCH("ngp_adam_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_encode_backward_input(int n_levels, int n_features, int log2_hashmap, int base_res,
                                            float per_level_scale, const float* positions, const void* params,
                                            const void* dLdoutT, float* dLdpos, long N, void* stream) {
  return ns_ngp_encode_backward_input_n(n_levels, n_features, log2_hashmap, base_res, per_level_scale, positions, params, dLdoutT,
                                        dLdpos, N, nullptr, stream);
}

extern "C" int ns_ngp_encode_backward_input_n(int n_levels, int n_features, int log2_hashmap, int base_res,
                                            float per_level_scale, const float* positions, const void* params,
                                            const void* dLdoutT, float* dLdpos, long N, const int* n_dev, void* stream) {
  NS_REQUIRE(positions && params && dLdoutT && dLdpos, "ns_ngp_encode_backward_input: null pointer");
  GridCfg c{n_levels, n_features, log2_hashmap, base_res, per_level_scale};
  GridLayout g;
  if (grid_layout_host(c, g) != NS_OK) {
    ns_set_error("ns_ngp_encode_backward_input: need 1..16 levels and 2 features per level");
    return NS_ENOSUP;
  }
  if (N <= 0) return NS_OK;
  hipLaunchKernelGGL(ngp_encode_bwd_input_kernel, dim3(ns_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, g, positions,
                     (const h2_t*)params, (const _Float16*)dLdoutT, dLdpos, N, n_levels, n_dev);
  NS_CHECK_LAUNCH("ngp_encode_bwd_input_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_camera_gradient(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                                      const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R,
                                      void* stream) {
  return ns_ngp_camera_gradient_ctl(dLdpos, tmid, rays_d, ray_start, ray_n, ray_img, pos_inv, cam_grad, R, nullptr, stream);
}

extern "C" int ns_ngp_camera_gradient_ctl(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                                          const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R,
                                          const int* ctl, void* stream) {
  return ns_ngp_camera_gradient_2stage(dLdpos, tmid, rays_d, ray_start, ray_n, ray_img, pos_inv, cam_grad, R, ctl, nullptr, 0,
                                       stream);
}

extern "C" int ns_ngp_camera_gradient_2stage(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                                             const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R,
                                             const int* ctl, float* ray_scratch, int n_images, void* stream) {
  NS_REQUIRE(ray_scratch == nullptr || (n_images >= 1 && n_images <= NS_CAM_LDS_IMAGES),
             "ns_ngp_camera_gradient_2stage: 1..%d images", NS_CAM_LDS_IMAGES);
  NS_REQUIRE(dLdpos && tmid && rays_d && ray_start && ray_n && ray_img && cam_grad, "ns_ngp_camera_gradient: null pointer");
  if (R <= 0) return NS_OK;
  hipLaunchKernelGGL(ngp_camera_grad_kernel, dim3(ns_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dLdpos, tmid, rays_d,
                     ray_start, ray_n, ray_img, pos_inv, cam_grad, R, ctl, ray_scratch);
  NS_CHECK_LAUNCH("ngp_camera_grad_kernel");
  if (ray_scratch != nullptr) {
    if (n_images <= NS_CAM_SMALL)
      hipLaunchKernelGGL(ngp_camera_grad_reduce_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ray_scratch, ray_n, ray_img,
                         cam_grad, R, n_images, ctl);
    else
      hipLaunchKernelGGL(ngp_camera_grad_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ray_scratch, ray_n, ray_img,
                         cam_grad, R, n_images, ctl);
    NS_CHECK_LAUNCH("ngp_camera_grad_reduce_kernel");
  }
  return NS_OK;
}

extern "C" int ns_ngp_camera_step(float* c2w, float* cam_grad, float* m1, float* m2, int n_images, int step, float lr_pos,
                                  float lr_rot, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return ns_ngp_camera_step_ctl(c2w, cam_grad, m1, m2, n_images, step, lr_pos, lr_rot, beta1, beta2, eps, grad_scale, nullptr,
                                stream);
}

extern "C" int ns_ngp_camera_step_ctl(float* c2w, float* cam_grad, float* m1, float* m2, int n_images, int step, float lr_pos,
                                      float lr_rot, float beta1, float beta2, float eps, float grad_scale, const int* ctl,
                                      void* stream) {
  NS_REQUIRE(c2w && cam_grad && m1 && m2, "ns_ngp_camera_step: null pointer");
  NS_REQUIRE((ctl || step >= 1) && grad_scale > 0.0f, "ns_ngp_camera_step: step must be >= 1 and grad_scale > 0");
  if (n_images <= 0) return NS_OK;
  const float st = (float)(step < 1 ? 1 : step);
  const float c1 = 1.0f - powf(beta1, st), c2 = 1.0f - powf(beta2, st);
  hipLaunchKernelGGL(ngp_camera_step_kernel, dim3(ns_cdiv(n_images, 64)), dim3(64), 0, (hipStream_t)stream, c2w, cam_grad,
                     m1, m2, n_images, c1, c2, lr_pos, lr_rot, beta1, beta2, eps, 1.0f / grad_scale, ctl);
  NS_CHECK_LAUNCH("ngp_camera_step_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_sample_rays(const float* images, const float* depths, const float* depth_covs, const float* c2w,
                                  int n_images, int H, int W, float fx, float fy, float cx, float cy, float box_lo,
                                  float box_hi, float near, unsigned seed, int R, float* rays_o, float* rays_d,
                                  float* t_range, float* gt_rgb, float* gt_depth, float* gt_depth_cov, int* ray_img,
                                  void* stream) {
  return ns_ngp_sample_rays_ctl(images, depths, depth_covs, c2w, n_images, H, W, fx, fy, cx, cy, box_lo, box_hi, near, seed, R,
                                rays_o, rays_d, t_range, gt_rgb, gt_depth, gt_depth_cov, ray_img, nullptr, 0, stream);
}

extern "C" int ns_ngp_sample_rays_ctl(const float* images, const float* depths, const float* depth_covs, const float* c2w,
                                      int n_images, int H, int W, float fx, float fy, float cx, float cy, float box_lo,
                                      float box_hi, float near, unsigned seed, int R, float* rays_o, float* rays_d,
                                      float* t_range, float* gt_rgb, float* gt_depth, float* gt_depth_cov, int* ray_img,
                                      const int* ctl, int step_offset, void* stream) {
  NS_REQUIRE(images && depths && depth_covs && c2w && rays_o && rays_d && t_range && gt_rgb && gt_depth && gt_depth_cov,
             "ns_ngp_sample_rays: null pointer");
  NS_REQUIRE(n_images > 0 && H > 0 && W > 0 && fx != 0.0f && fy != 0.0f && box_hi > box_lo,
             "ns_ngp_sample_rays: bad image set / intrinsics / box");
  if (R <= 0) return NS_OK;
  SampleRaysArgs a{images, depths, depth_covs, c2w, fx, fy, cx, cy, box_lo, box_hi, near, n_images, H, W, R, seed,
                   rays_o, rays_d, t_range, gt_rgb, gt_depth, gt_depth_cov, ray_img, ctl, step_offset};
  hipLaunchKernelGGL(ngp_sample_rays_kernel, dim3(ns_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_sample_rays_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_march(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d,
                            const float* t_range, int R, float cone, float min_step, float max_step, float pos_lo,
                            float pos_inv, int max_per_ray, long max_samples, int* counter, int* ray_start, int* ray_n, float* pos, float* dirs,
                            float* dt, float* tmid, void* stream) {
  return ns_ngp_march_ctl(bits, G, ncasc, rays_o, rays_d, t_range, R, cone, min_step, max_step, pos_lo, pos_inv, max_per_ray,
                          max_samples, counter, ray_start, ray_n, pos, dirs, dt, tmid, nullptr, stream);
}

extern "C" int ns_ngp_march_ctl(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d,
                                const float* t_range, int R, float cone, float min_step, float max_step, float pos_lo,
                                float pos_inv, int max_per_ray, long max_samples, int* counter, int* ray_start, int* ray_n,
                                float* pos, float* dirs, float* dt, float* tmid, const int* ctl, void* stream) {
  NS_REQUIRE(bits && rays_o && rays_d && t_range && counter && ray_start && ray_n && pos && dirs && dt && tmid,
             "ns_ngp_march: null pointer");
  NS_REQUIRE(G > 0 && ncasc >= 1 && ncasc <= 8 && min_step > 0.0f, "ns_ngp_march: bad grid");
  if (R <= 0) return NS_OK;
  return ns_ngp_march_ordered(bits, G, ncasc, rays_o, rays_d, t_range, R, cone, min_step, max_step, pos_lo, pos_inv, max_per_ray,
                              max_samples, counter, ray_start, ray_n, pos, dirs, dt, tmid, ctl, nullptr, stream);
}

extern "C" int ns_ngp_march_ordered(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d,
                                    const float* t_range, int R, float cone, float min_step, float max_step, float pos_lo,
                                    float pos_inv, int max_per_ray, long max_samples, int* counter, int* ray_start, int* ray_n,
                                    float* pos, float* dirs, float* dt, float* tmid, const int* ctl, unsigned long long* order_ws,
                                    void* stream) {
  NS_REQUIRE(bits && rays_o && rays_d && t_range && counter && ray_start && ray_n && pos && dirs && dt && tmid,
             "ns_ngp_march: null pointer");
  NS_REQUIRE(G > 0 && ncasc >= 1 && ncasc <= 8 && min_step > 0.0f, "ns_ngp_march: bad grid");
  NS_REQUIRE(order_ws == nullptr || ((uintptr_t)order_ws % 8) == 0, "ns_ngp_march_ordered: order_ws must be 8-byte aligned");
  if (R <= 0) return NS_OK;
  MarchArgs a{bits, rays_o, rays_d, t_range, cone, min_step, max_step, pos_lo, pos_inv, G, ncasc, max_per_ray, max_samples, counter,
              ray_start, ray_n, pos, dirs, dt, tmid, R, ctl, order_ws};
  hipLaunchKernelGGL(ngp_march_kernel, dim3(ns_cdiv(R, 16)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_march_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_composite(const void* net_out, const float* dt, const float* tmid, const int* ray_start,
                                const int* ray_n, int R, const float* gt_rgb, const float* gt_depth,
                                const float* gt_depth_cov, float depth_lambda, float loss_scale, float* out_rgb,
                                float* out_depth, float* loss, void* dLdout, void* stream) {
  return ns_ngp_composite_ctl(net_out, dt, tmid, ray_start, ray_n, R, gt_rgb, gt_depth, gt_depth_cov, depth_lambda, loss_scale,
                              out_rgb, out_depth, loss, dLdout, nullptr, stream);
}

extern "C" int ns_ngp_composite_ctl(const void* net_out, const float* dt, const float* tmid, const int* ray_start,
                                    const int* ray_n, int R, const float* gt_rgb, const float* gt_depth,
                                    const float* gt_depth_cov, float depth_lambda, float loss_scale, float* out_rgb,
                                    float* out_depth, float* loss, void* dLdout, const int* ctl, void* stream) {
  return ns_ngp_composite_rays(net_out, dt, tmid, ray_start, ray_n, R, gt_rgb, gt_depth, gt_depth_cov, depth_lambda, loss_scale,
                               out_rgb, out_depth, loss, nullptr, dLdout, ctl, stream);
}

extern "C" int ns_ngp_composite_rays(const void* net_out, const float* dt, const float* tmid, const int* ray_start,
                                     const int* ray_n, int R, const float* gt_rgb, const float* gt_depth,
                                     const float* gt_depth_cov, float depth_lambda, float loss_scale, float* out_rgb,
                                     float* out_depth, float* loss, float* ray_loss, void* dLdout, const int* ctl, void* stream) {
  NS_REQUIRE(net_out && dt && tmid && ray_start && ray_n && out_rgb && out_depth, "ns_ngp_composite: null pointer");
  NS_REQUIRE(dLdout == nullptr || (gt_rgb && gt_depth && gt_depth_cov && (loss || ray_loss)),
             "ns_ngp_composite: training mode needs the ground truth and a loss pointer");
  if (R <= 0) return NS_OK;
  CompositeArgs a{(const _Float16*)net_out, dt, tmid, ray_start, ray_n, gt_rgb, gt_depth, gt_depth_cov,
                  depth_lambda, loss_scale, out_rgb, out_depth, loss, ray_loss, (_Float16*)dLdout, R, ctl};
  hipLaunchKernelGGL(ngp_composite_kernel, dim3(ns_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
  NS_CHECK_LAUNCH("ngp_composite_kernel");
  return NS_OK;
}

// End of a graph-captured training step, in two halves so that the NEXT step's rays can be sampled and marched (side stream
// of the graph) while this step's optimiser pass runs:
//   ngp_step_rays_kernel   after the backward pass: keep a copy of this step's march counters for (lazy) host reads, adapt the
//                          ray count of the next batch so that the sample budget stays ~`fill` full without refusing rays
//                          (the rule of nerfslam/ngp.py; instant-ngp adapts its batch likewise), clear the counters;
//   ngp_step_count_kernel  after the optimiser AND the next march have finished: count the step, Adam's bias corrections
//                          for the next one.
__global__ void ngp_step_rays_kernel(int* __restrict__ ctl, int* __restrict__ counter, int* __restrict__ last, float fill,
                                     long max_samples, int min_rays, int max_rays) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int requested = counter[0], R = ctl[NS_CTL_RAYS];
  last[0] = requested;
  last[1] = counter[1];
  last[2] = counter[2];
  last[3] = R;
  const float want = (float)R * fill * (float)max_samples / (float)max(requested, 1);
  int Rn = (int)fminf(fmaxf(want, (float)min_rays), (float)max_rays);
  Rn = Rn / 128 * 128;
  ctl[NS_CTL_RAYS] = max(Rn, 128);
  counter[0] = counter[1] = counter[2] = 0;
}

__global__ void ngp_step_count_kernel(int* __restrict__ ctl, float beta1, float beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int done = ctl[NS_CTL_STEP] + 1;
  ctl[NS_CTL_STEP] = done;
  ctl[NS_CTL_C1] = __float_as_int(1.0f - powf(beta1, (float)(done + 1)));
  ctl[NS_CTL_C2] = __float_as_int(1.0f - powf(beta2, (float)(done + 1)));
}

// Double-buffered form (round 3): every step owns a SET {control block, march counters, ray tables, sample arrays}; the rays of
// step k + 1 are sampled and marched into the other set at the START of step k's launch sequence, on a side stream, next to
// step k's forward / backward passes (the marcher is latency bound and reads only the occupancy bits and the images).  This
// kernel opens that side branch: it derives the next step's control block from this step's -- step number + 1, ray count
// adapted from THIS step's sample count (known since its march finished, one step earlier than the in-line rule could use
// it), Adam's bias corrections -- records this step's march counters for lazy host reads and clears the other set's.
__global__ void ngp_step_prepare_kernel(const int* __restrict__ ctl_src, int* __restrict__ ctl_dst,
                                        const int* __restrict__ counter_src, int* __restrict__ counter_dst, int* __restrict__ last,
                                        float fill, long max_samples, int min_rays, int max_rays, float beta1, float beta2,
                                        float* __restrict__ loss_dst) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (loss_dst != nullptr) *loss_dst = 0.0f;
  const int requested = counter_src[0], R = ctl_src[NS_CTL_RAYS];
  last[0] = requested;
  last[1] = counter_src[1];
  last[2] = counter_src[2];
  last[3] = R;
  const float want = (float)R * fill * (float)max_samples / (float)max(requested, 1);
  int Rn = (int)fminf(fmaxf(want, (float)min_rays), (float)max_rays);
  Rn = Rn / 128 * 128;
  const int done = ctl_src[NS_CTL_STEP] + 1;
  ctl_dst[NS_CTL_STEP] = done;
  ctl_dst[NS_CTL_RAYS] = max(Rn, 128);
  ctl_dst[NS_CTL_SEED] = ctl_src[NS_CTL_SEED];
  ctl_dst[NS_CTL_VIEWS] = ctl_src[NS_CTL_VIEWS];
  ctl_dst[NS_CTL_C1] = __float_as_int(1.0f - powf(beta1, (float)(done + 1)));
  ctl_dst[NS_CTL_C2] = __float_as_int(1.0f - powf(beta2, (float)(done + 1)));
  counter_dst[0] = counter_dst[1] = counter_dst[2] = 0;
}

extern "C" int ns_ngp_step_prepare(const int* ctl_src, int* ctl_dst, const int* counter_src, int* counter_dst, int* last, float fill,
                                   long max_samples, int min_rays, int max_rays, float beta1, float beta2, float* loss_dst,
                                   void* stream) {
  NS_REQUIRE(ctl_src && ctl_dst && counter_src && counter_dst && last, "ns_ngp_step_prepare: null pointer");
  NS_REQUIRE(ctl_src != ctl_dst && counter_src != counter_dst, "ns_ngp_step_prepare: the two sets must be distinct");
  NS_REQUIRE(fill > 0.0f && max_samples > 0 && min_rays >= 128 && max_rays >= min_rays, "ns_ngp_step_prepare: bad limits");
  hipLaunchKernelGGL(ngp_step_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl_src, ctl_dst, counter_src, counter_dst,
                     last, fill, max_samples, min_rays, max_rays, beta1, beta2, loss_dst);
  NS_CHECK_LAUNCH("ngp_step_prepare_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_step_rays(int* ctl, int* counter, int* last, float fill, long max_samples, int min_rays, int max_rays,
                                void* stream) {
  NS_REQUIRE(ctl && counter && last, "ns_ngp_step_rays: null pointer");
  NS_REQUIRE(fill > 0.0f && max_samples > 0 && min_rays >= 128 && max_rays >= min_rays, "ns_ngp_step_rays: bad limits");
  hipLaunchKernelGGL(ngp_step_rays_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, counter, last, fill, max_samples,
                     min_rays, max_rays);
  NS_CHECK_LAUNCH("ngp_step_rays_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_step_count(int* ctl, float beta1, float beta2, void* stream) {
  NS_REQUIRE(ctl, "ns_ngp_step_count: null pointer");
  hipLaunchKernelGGL(ngp_step_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, beta1, beta2);
  NS_CHECK_LAUNCH("ngp_step_count_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_step_advance(int* ctl, int* counter, int* last, float fill, long max_samples, int min_rays, int max_rays,
                                   float beta1, float beta2, void* stream) {
  const int rc = ns_ngp_step_rays(ctl, counter, last, fill, max_samples, min_rays, max_rays, stream);
  return rc != NS_OK ? rc : ns_ngp_step_count(ctl, beta1, beta2, stream);
}

// ---------------------------------------------------------------------------------------------
// Occupancy-grid refresh (instant-ngp `update_density_grid_nerf`, the subset form of nerfslam/ngp.py): what used to be
// ~35 torch launches per update (randint / rand / stack / index_put / mean / compare / pack ...: 0.26 ms of GPU time and as
// much host time on the mapper thread, once per 16 optimiser steps) as five kernels around the encode + density network:
//   ngp_grid_cells     n cells drawn uniformly over all cascades (PCG hash of seed + index), a jittered point in each, in the
//                      unit cube of the render box;
//   (ns_ngp_encode_forward + ns_ngp_mlp_forward on those points)
//   ngp_grid_decay     grid *= decay  (every cell)
//   ngp_grid_max       grid[cell] = max(grid[cell], exp(log-density) * min_step): integer atomicMax on the float bits
//                      (both sides >= 0), order independent -> replicas stay in lockstep;
//   ngp_grid_sum / ngp_grid_bits   mean of the grid (two-stage, fixed order), occupied = grid > min(mean, threshold),
//                      8 cells per output byte (bit i of byte j = cell 8 j + i, the marcher's layout).
// ---------------------------------------------------------------------------------------------
// ctl != nullptr: the draw's seed comes from the step's control block -- seed ^ pcg(optimiser steps completed + 1) -- so that a
// refresh CAPTURED into the step graph (nerfslam/ngp.py: the refresh rides on the step before an update, round 6) draws new
// cells on every replay
__global__ __launch_bounds__(256) void ngp_grid_cells_kernel(int G, int ncasc, uint32_t seed, int n, float box_lo, float inv_box,
                                                             int* __restrict__ cells, float* __restrict__ pos_unit,
                                                             const int* __restrict__ ctl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (ctl != nullptr) seed ^= ns_pcg((uint32_t)ctl[NS_CTL_STEP] + 1u);
  const uint32_t G3 = (uint32_t)G * G * G, total = G3 * (uint32_t)ncasc;
  const uint32_t base = seed + 4u * (uint32_t)i;
  const uint32_t cell = ns_pcg(base) % total;
  const uint32_t mip = cell / G3, r = cell - mip * G3;
  const uint32_t x = r % (uint32_t)G, y = (r / (uint32_t)G) % (uint32_t)G, z = r / ((uint32_t)G * G);
  const float scale = (float)(1u << mip), inv_g = 1.0f / (float)G;
  const float j0 = (float)(ns_pcg(base + 1u) >> 8) * (1.0f / 16777216.0f), j1 = (float)(ns_pcg(base + 2u) >> 8) * (1.0f / 16777216.0f),
              j2 = (float)(ns_pcg(base + 3u) >> 8) * (1.0f / 16777216.0f);
  const float sx = (((float)x + j0) * inv_g - 0.5f) * scale + 0.5f, sy = (((float)y + j1) * inv_g - 0.5f) * scale + 0.5f,
              sz = (((float)z + j2) * inv_g - 0.5f) * scale + 0.5f;
  cells[i] = (int)cell;
  pos_unit[3 * (long)i] = (sx - box_lo) * inv_box;
  pos_unit[3 * (long)i + 1] = (sy - box_lo) * inv_box;
  pos_unit[3 * (long)i + 2] = (sz - box_lo) * inv_box;
}

__global__ __launch_bounds__(256) void ngp_grid_decay_kernel(float* __restrict__ grid, long n4, float decay) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(grid)[i];
  v.x *= decay; v.y *= decay; v.z *= decay; v.w *= decay;
  reinterpret_cast<float4*>(grid)[i] = v;
}

__global__ __launch_bounds__(256) void ngp_grid_max_kernel(const _Float16* __restrict__ net_out, const int* __restrict__ cells, int n,
                                                           float min_step, float* __restrict__ grid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dens = __expf((float)net_out[4 * (long)i + 3]) * min_step;
  if (dens == dens) atomicMax(reinterpret_cast<int*>(grid) + cells[i], __float_as_int(fmaxf(dens, 0.0f)));
}

// The subset refresh re-evaluates 2^18 of the 3 x 128^3 cells per update.  Decaying EVERY cell on every update (the rule above,
// instant-ngp's -- which however re-evaluates HALF of the grid each time) lets a moderately dense cell fall to 0.95^25 ~ 0.28 of
// its value before it is drawn again: cells drop below the occupancy threshold although nothing changed.  The
// sampled form advances a cell's moving maximum only when the cell is observed:
//   grid[c] = max(decay * grid[c], max over this update's samples of c)       for the drawn cells, nothing elsewhere.
// Three passes over the samples, deterministic whatever the number of draws of a cell (`tmp`: a zeroed float per cell, left
// zeroed): maximum into tmp (integer atomicMax on the float bits), the ONE lane that swaps a cell's maximum out applies it, reset.
__global__ __launch_bounds__(256) void ngp_grid_tmpmax_kernel(const _Float16* __restrict__ net_out, const int* __restrict__ cells, int n,
                                                              float min_step, float* __restrict__ tmp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dens = __expf((float)net_out[4 * (long)i + 3]) * min_step;
  if (dens == dens) atomicMax(reinterpret_cast<int*>(tmp) + cells[i], __float_as_int(fmaxf(dens, 0.0f)));
}
__global__ __launch_bounds__(256) void ngp_grid_apply_kernel(const int* __restrict__ cells, int n, float decay, float* __restrict__ tmp,
                                                             float* __restrict__ grid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = cells[i];
  const int v = atomicExch(reinterpret_cast<int*>(tmp) + c, (int)0x80000000);   // -0.0f: taken
  if (v >= 0) grid[c] = fmaxf(grid[c] * decay, __int_as_float(v));
}
__global__ __launch_bounds__(256) void ngp_grid_tmpclear_kernel(const int* __restrict__ cells, int n, float* __restrict__ tmp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) tmp[cells[i]] = 0.0f;
}

#define NS_GRID_PARTS 256
__global__ __launch_bounds__(256) void ngp_grid_sum_kernel(const float* __restrict__ grid, long n, double* __restrict__ partial) {
  __shared__ double red[256];
  const long per = (n + NS_GRID_PARTS - 1) / NS_GRID_PARTS;
  const long lo = (long)blockIdx.x * per, hi = min(n, lo + per);
  double s = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += (double)grid[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ngp_grid_bits_kernel(const float* __restrict__ grid, long n, const double* __restrict__ partial,
                                                            float max_thr, uint8_t* __restrict__ bits) {
  __shared__ double red[256];
  red[threadIdx.x] = partial[threadIdx.x];     // NS_GRID_PARTS == blockDim.x
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float thr = fminf((float)(red[0] / (double)n), max_thr);
  const long j = (long)blockIdx.x * 256 + threadIdx.x;       // output byte
  if (j * 8 >= n) return;
  const float4 a = reinterpret_cast<const float4*>(grid)[2 * j], b = reinterpret_cast<const float4*>(grid)[2 * j + 1];
  const uint32_t byte = (a.x > thr ? 1u : 0u) | (a.y > thr ? 2u : 0u) | (a.z > thr ? 4u : 0u) | (a.w > thr ? 8u : 0u) |
                        (b.x > thr ? 16u : 0u) | (b.y > thr ? 32u : 0u) | (b.z > thr ? 64u : 0u) | (b.w > thr ? 128u : 0u);
  bits[j] = (uint8_t)byte;
}

extern "C" int ns_ngp_grid_cells(int grid_size, int n_cascades, unsigned seed, int n, float box_lo, float box_hi, int* cells,
                                 float* pos_unit, void* stream) {
  NS_REQUIRE(cells && pos_unit, "ns_ngp_grid_cells: null pointer");
  NS_REQUIRE(grid_size > 0 && grid_size <= 512 && n_cascades >= 1 && n_cascades <= 8 && box_hi > box_lo && n >= 0,
             "ns_ngp_grid_cells: bad grid / box");
  if (n == 0) return NS_OK;
  hipLaunchKernelGGL(ngp_grid_cells_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, grid_size, n_cascades,
                     (uint32_t)seed, n, box_lo, 1.0f / (box_hi - box_lo), cells, pos_unit, (const int*)nullptr);
  NS_CHECK_LAUNCH("ngp_grid_cells_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_grid_cells_ctl(int grid_size, int n_cascades, unsigned seed, int n, float box_lo, float box_hi, int* cells,
                                     float* pos_unit, const int* ctl, void* stream) {
  NS_REQUIRE(cells && pos_unit && ctl, "ns_ngp_grid_cells_ctl: null pointer");
  NS_REQUIRE(grid_size > 0 && grid_size <= 512 && n_cascades >= 1 && n_cascades <= 8 && box_hi > box_lo && n >= 0,
             "ns_ngp_grid_cells_ctl: bad grid / box");
  if (n == 0) return NS_OK;
  hipLaunchKernelGGL(ngp_grid_cells_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, grid_size, n_cascades,
                     (uint32_t)seed, n, box_lo, 1.0f / (box_hi - box_lo), cells, pos_unit, ctl);
  NS_CHECK_LAUNCH("ngp_grid_cells_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_grid_update_sampled(const void* net_out, const int* cells, int n, float min_step, float decay,
                                          float max_threshold, float* density_grid, float* tmp_grid, long n_cells_total,
                                          double* partial_ws, unsigned char* bits, void* stream) {
  NS_REQUIRE(net_out && cells && density_grid && tmp_grid && partial_ws && bits, "ns_ngp_grid_update_sampled: null pointer");
  NS_REQUIRE(n >= 0 && n_cells_total > 0 && n_cells_total % 8 == 0, "ns_ngp_grid_update_sampled: the grid must hold a multiple of 8 cells");
  hipStream_t st = (hipStream_t)stream;
  if (n > 0) {
    hipLaunchKernelGGL(ngp_grid_tmpmax_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, st, (const _Float16*)net_out, cells, n, min_step,
                       tmp_grid);
    NS_CHECK_LAUNCH("ngp_grid_tmpmax_kernel");
    hipLaunchKernelGGL(ngp_grid_apply_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, st, cells, n, decay, tmp_grid, density_grid);
    NS_CHECK_LAUNCH("ngp_grid_apply_kernel");
    hipLaunchKernelGGL(ngp_grid_tmpclear_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, st, cells, n, tmp_grid);
    NS_CHECK_LAUNCH("ngp_grid_tmpclear_kernel");
  }
  hipLaunchKernelGGL(ngp_grid_sum_kernel, dim3(NS_GRID_PARTS), dim3(256), 0, st, density_grid, n_cells_total, partial_ws);
  NS_CHECK_LAUNCH("ngp_grid_sum_kernel");
  hipLaunchKernelGGL(ngp_grid_bits_kernel, dim3(ns_cdiv(n_cells_total / 8, 256)), dim3(256), 0, st, density_grid, n_cells_total,
                     partial_ws, max_threshold, bits);
  NS_CHECK_LAUNCH("ngp_grid_bits_kernel");
  return NS_OK;
}

extern "C" int ns_ngp_grid_update(const void* net_out, const int* cells, int n, float min_step, float decay, float max_threshold,
                                  float* density_grid, long n_cells_total, double* partial_ws, unsigned char* bits, void* stream) {
  NS_REQUIRE(net_out && cells && density_grid && partial_ws && bits, "ns_ngp_grid_update: null pointer");
  NS_REQUIRE(n >= 0 && n_cells_total > 0 && n_cells_total % 8 == 0 && ((uintptr_t)density_grid % 16) == 0,
             "ns_ngp_grid_update: the grid must hold a multiple of 8 cells and be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ngp_grid_decay_kernel, dim3(ns_cdiv(n_cells_total / 4, 256)), dim3(256), 0, st, density_grid,
                     n_cells_total / 4, decay);
  NS_CHECK_LAUNCH("ngp_grid_decay_kernel");
  if (n > 0) {
    hipLaunchKernelGGL(ngp_grid_max_kernel, dim3(ns_cdiv(n, 256)), dim3(256), 0, st, (const _Float16*)net_out, cells, n, min_step,
                       density_grid);
    NS_CHECK_LAUNCH("ngp_grid_max_kernel");
  }
  hipLaunchKernelGGL(ngp_grid_sum_kernel, dim3(NS_GRID_PARTS), dim3(256), 0, st, density_grid, n_cells_total, partial_ws);
  NS_CHECK_LAUNCH("ngp_grid_sum_kernel");
  hipLaunchKernelGGL(ngp_grid_bits_kernel, dim3(ns_cdiv(n_cells_total / 8, 256)), dim3(256), 0, st, density_grid, n_cells_total,
                     partial_ws, max_threshold, bits);
  NS_CHECK_LAUNCH("ngp_grid_bits_kernel");
  return NS_OK;
}
