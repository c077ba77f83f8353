// conv.hip -- channels-last f16 convolution (3x3 pad 1, or 1x1) as an implicit GEMM on the MFMA units, for the update
// operator of the tracker (networks/droid_net.py:78-150, networks/modules/gru.py:5-34 -- ConvGRU gates,
// correlation / flow encoders, the delta / weight heads and GraphAgg are all 3x3 or 1x1 convolutions over [E, 128..448, 60, 80]).
// torch + MIOpen run that operator at ~260 TFLOP/s (tools/nets_bench.py: 4.1 ms for E=48) with a dozen cat / bias /
// activation / layout kernels in between; this kernel takes the concatenation as a LIST of source tensors, adds a
// (per-image) bias, applies the activation and writes f16 straight into a channel slice of a channels-last tensor.
//
// GEMM view: out^T[cout][pixel] = sum over (tap, cin) of W[cout][tap, cin] * in[pixel + tap][cin], computed with
// v_mfma_f32_32x32x16_f16 (A = 32 couts x 16 cin, B = 16 cin x 32 pixels); the accumulator layout then gives every lane 4
// CONSECUTIVE couts of one pixel, i.e. 8-byte channels-last stores.
//
// Workgroup = 4 CG waves = a tile of (8 UT) rows x 16 columns of one image x (32 MT) couts.  Per 16-channel chunk of the
// input the workgroup stages into LDS (two stages, one barrier per chunk)
//   * the input slab: tile + halo, 48-byte pixel stride (16 channels + pad: the lanes of a b128 read group are adjacent
//     pixels, 48 B = 12 banks apart), zero-filled outside the image -- global -> registers -> LDS;
//   * the chunk's weights for all taps, pre-packed on the host in fragment order (a wave's A fragment is 1 KB contiguous)
//     -- global -> LDS directly (LDS-DMA).
// Wave w owns pixel rows 2 UT (w % 4).. (UT column tiles of 2 rows x 16 pixels) and the couts of A tiles
// [MT/CG (w / 4), MT/CG (w / 4 + 1)): per tap it reads UT B fragments and MT/CG A fragments (ds_read_b128) for UT MT/CG MFMAs.
// Instantiated: 128-cout tile with 8 waves (CG = 2, two waves per SIMD) and 16- or 32-row tiles (UT = 2 for 1x1, 4 for 3x3);
// 64- and 32-cout tiles with 4 waves for the small layers.  DESIGN.md section 8 has the measurements behind each choice.
#include "common.h"

typedef _Float16 cv_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_f16x4 __attribute__((ext_vector_type(4)));
typedef float cv_f32x16 __attribute__((ext_vector_type(16)));

#define CV_TC 16       // tile columns
#define CV_SP 24       // slab pixel stride in halves (48 B)
#define CV_MAXSRC 4
// Taps of a chunk over which the NEXT chunk's weight LDS-DMAs are issued.  Rounds 2-4 spread them over all nine; issued with the
// first tap they have the whole chunk to land before the vmcnt(0) that publishes the stage (round 5, two runs per arm on one
// box, E = 48: 448 -> 256 539 / 539 -> 529 / 523 us, 448 -> 128 249 / 248 -> 238 / 242, 128 -> 128 92.2 / 91.7 -> 89.9 / 90.9;
// three taps: in between; writing the next slab to LDS two taps before the end instead of after the last MFMA: no further
// gain at 256 registers) -- a 2-3 % effect: the chunk boundary is not where this kernel's other 60 % of the matrix core goes.
#ifndef CV_WTAPS
#define CV_WTAPS 1
#endif

struct ConvArgs {
  const _Float16* src[CV_MAXSRC];  // virtual concatenation along channels, each [N,H,W,src_ch[s]]
  int src_ch[CV_MAXSRC];
  int src_stride[CV_MAXSRC];       // pixel stride in halves (>= src_ch: a source may be a channel slice of a wider tensor)
  int src_start[CV_MAXSRC + 1];    // cumulative channel offsets
  int nsrc;
  int N, H, W, CI, CO, COP;        // COP = CO rounded up to the workgroup's cout tile (the packed weights are padded)
  const cv_f16x8* wp;              // [CI/16][taps][COP/32][64 lanes] fragments
  const float* bias;               // bias[n * bias_nstride + co], or nullptr
  long bias_nstride;
  _Float16* out;                   // out[((n H + y) W + x) * ostride + ooff + co]
  int ostride, ooff;
  int act;                         // NS_ACT_*
  int vec;                         // output slice 4-channel aligned: 8-byte stores
  int fuse;                        // NS_CONV_FUSE_*: extra elementwise step of the ConvGRU in the epilogue
  const _Float16* e0;              // MUL_HI: the factor [N,H,W,e0s] for the upper half of the couts; GRU: z
  const _Float16* e1;              // GRU: h
  int e0s, e1s;                    // pixel strides (elements)
  int tiles_y, tiles_x;
};

__device__ __forceinline__ float cv_act(float v, int act) {
  switch (act) {
    case NS_ACT_RELU: return fmaxf(v, 0.0f);
    case NS_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case NS_ACT_TANH: return 2.0f / (1.0f + __expf(-2.0f * v)) - 1.0f;
    default: return v;
  }
}

template <bool B>
struct cv_bool { static constexpr bool value = B; };

// UT = column tiles (2 rows x 16 pixels) per wave; CG = cout groups: the workgroup is 4 CG waves, wave w owns pixel rows
// 2 UT (w % 4) .. and the couts of A tiles [MTW (w / 4), MTW (w / 4 + 1)), MTW = MT / CG.  CG = 2 puts two waves on every
// SIMD (64 accumulator registers each): one's waits (fragment reads, LDS-DMA issue) are the other's MFMA time.
template <int KS, int MT, int UT, int CG>
__global__ __launch_bounds__(256 * CG) void conv_nhwc_kernel(ConvArgs a) {
  constexpr int NT = 256 * CG, MTW = MT / CG;
  constexpr int HALO = KS / 2, T = KS * KS;
  constexpr int TR = 8 * UT;
  constexpr int SR = TR + 2 * HALO, SC = CV_TC + 2 * HALO, SPIX = SR * SC;
  // Row stride of the slab in halves: a multiple of 256 B.  A B fragment is one ds_read_b128 whose 16-lane service groups are NOT
  // 16 consecutive lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md): every group mixes pixels of the lane's row with pixels
  // of the NEXT row, so the two rows must be a multiple of the 64 banks apart or two pairs of lanes of every group meet on a
  // bank (the packed 18 x 48 B rows of the 3x3 kernel: SQ_LDS_BANK_CONFLICT = 40 % of the LDS-active cycles of the gate
  // convolution).  16 x 48 B = 768 B (1x1) already is; 18 x 48 B is padded to 1024 B.
  constexpr int RSH = ((SC * CV_SP * 2 + 255) / 256 * 256) / 2;
  constexpr int NPS = (SPIX * 2 + NT - 1) / NT;  // 16-byte slab pieces per thread
  constexpr int WV = T * MT * 64;                // weight fragments (16 B) per stage
  constexpr int NPW = (WV + NT - 1) / NT;        // LDS-DMA instructions per thread and chunk
  constexpr int ERS = MTW * 32 + 4;              // epilogue row stride in halves (+8 B: conflict-free ds_write_b64)
  // one LDS block: [slab stage 0 | slab stage 1 | weight stage 0 | weight stage 1]; the epilogue's transposition tiles
  // (per wave 32 pixels x 32 MT couts) reuse it from offset 0 once the last chunk is done
  constexpr int SLAB_H = SR * RSH;                             // halves per slab stage
  constexpr int MAIN_BYTES = 2 * SLAB_H * 2 + 2 * WV * 16, EPI_BYTES = 4 * CG * 32 * ERS * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  _Float16* const slab0 = reinterpret_cast<_Float16*>(lds_raw);
  cv_f16x8* const wl0 = reinterpret_cast<cv_f16x8*>(lds_raw + 2 * SLAB_H * 2);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pg = wv & 3, cg = wv >> 2;              // pixel-row group / cout group of this wave
  const int j = lane & 31, h = lane >> 5;
  // 1-D grid.  Workgroup ids are dealt round-robin to the 8 XCDs, so the cout tiles of one pixel tile get ids 8 apart:
  // same XCD (same L2), dispatched back to back -- the second one finds the input slab in L2 instead of fetching it again
  // (448 -> 256: two cout tiles, 2 x 262 MB of input otherwise).
  const int L = blockIdx.x, nz = a.COP / (32 * MT);
  const int cz = (L >> 3) % nz;
  const int tile = (L / (8 * nz)) * 8 + (L & 7);
  if (tile >= a.tiles_x * a.tiles_y * a.N) return;          // (workgroup-uniform: before any barrier)
  const int bx = tile % a.tiles_x, by = tile / a.tiles_x;
  const int n = by / a.tiles_y;
  const int y0 = (by - n * a.tiles_y) * TR, x0 = bx * CV_TC;
  const int nchunk = a.CI >> 4;
  const int ctiles = a.COP >> 5;

  cv_f32x16 acc[MTW][UT];
#pragma unroll
  for (int m = 0; m < MTW; m++)
#pragma unroll
    for (int u = 0; u < UT; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][u][r] = 0.0f;

  // ---- staging ----
  // slab: this thread's 16-byte pieces go through registers (the padded pixel stride rules out an LDS-DMA image);
  // weights: global -> LDS directly (global_load_lds_dwordx4: LDS address = wave-uniform base + 16 lane, which is exactly
  // the fragment order they are packed in), no registers, no ds_write pass.
  uint4 ps[NPS];
  int s_off[NPS];        // slab offset (halves) of piece q, or -1
  long g_pix[NPS];       // pixel index (n H + yy) W + xx of the piece; outside the image: pixel 0 (loaded, then zeroed)
  bool g_ok[NPS];
#pragma clang loop unroll(full)
  for (int q = 0; q < NPS; q++) {
    const int p = tid + q * NT;
    const int sp = p >> 1;
    s_off[q] = -1;
    g_pix[q] = 0;
    g_ok[q] = false;
    if (sp < SPIX) {
      const int sr = sp / SC, sc = sp - sr * SC;
      const int yy = y0 + sr - HALO, xx = x0 + sc - HALO;
      s_off[q] = sr * RSH + sc * CV_SP + (p & 1) * 8;
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
        g_pix[q] = ((long)n * a.H + yy) * a.W + xx;
        g_ok[q] = true;
      }
    }
  }
  const int e8 = (tid & 1) * 8;
  // source tensor / channel offset of chunk c
  const _Float16* cbase = nullptr;
  int csch = 0, ccoff = 0;
  const cv_f16x8* cwsrc = nullptr;
  // The source table lives in scalar registers, read from the kernel arguments ONCE: indexed dynamically (a.src[s]) every chunk
  // began with a chain of dependent scalar loads from the argument segment, one per source walked (round 5: the one-edge gate
  // convolutions over [net | inp | corr | flow] 30.8 -> 28.3 us and 29.6 -> 26.8 us; single-source layers and E = 48 unchanged).
  const _Float16 *const sp0 = a.src[0], *const sp1 = a.src[1], *const sp2 = a.src[2], *const sp3 = a.src[3];
  const int ss0 = a.src_stride[0], ss1 = a.src_stride[1], ss2 = a.src_stride[2], ss3 = a.src_stride[3];
  const int st1 = a.nsrc > 1 ? a.src_start[1] : 0x7fffffff, st2 = a.nsrc > 2 ? a.src_start[2] : 0x7fffffff,
            st3 = a.nsrc > 3 ? a.src_start[3] : 0x7fffffff;       // (entries past nsrc are not initialised by the host)
  const intptr_t dp1 = reinterpret_cast<intptr_t>(sp1) - reinterpret_cast<intptr_t>(sp0),
                 dp2 = reinterpret_cast<intptr_t>(sp2) - reinterpret_cast<intptr_t>(sp1),
                 dp3 = reinterpret_cast<intptr_t>(sp3) - reinterpret_cast<intptr_t>(sp2);
  auto chunk_source = [&](int c) __attribute__((always_inline)) {
    const int cb = c << 4;
    // (starts are increasing: g3 => g2 => g1.  Written as sums of differences: a chain of selects over the four entries is
    //  turned back into an indexed table by the compiler -- in scratch memory)
    const bool g1 = cb >= st1, g2 = cb >= st2, g3 = cb >= st3;
    cbase = reinterpret_cast<const _Float16*>(reinterpret_cast<intptr_t>(sp0) + (g1 ? dp1 : 0) + (g2 ? dp2 : 0) + (g3 ? dp3 : 0));
    csch = ss0 + (g1 ? ss1 - ss0 : 0) + (g2 ? ss2 - ss1 : 0) + (g3 ? ss3 - ss2 : 0);
    ccoff = cb - ((g1 ? st1 : 0) + (g2 ? st2 - st1 : 0) + (g3 ? st3 - st2 : 0)) + e8;
    cwsrc = a.wp + ((long)c * T * ctiles + cz * MT) * 64;
  };
  auto load_slab_piece = [&](int q) __attribute__((always_inline)) {   // unconditional: no branch inside the MFMA stream
    // Through an address-space-1 pointer: `cbase` is rebuilt from integers (above), so a plain dereference is a FLAT load -- and
    // flat loads count on lgkmcnt as well as vmcnt (they might hit LDS): every `s_waitcnt lgkmcnt(0)` in front of a tap's first
    // MFMA then also waited for the next chunk's slab pieces to come back from L2 / HBM (round 6, found in the ISA).
    typedef uint32_t cv_u32x4 __attribute__((ext_vector_type(4)));
    const cv_u32x4 d = *reinterpret_cast<const __attribute__((address_space(1))) cv_u32x4*>(
        reinterpret_cast<uintptr_t>(cbase + g_pix[q] * csch + ccoff));
    ps[q] = make_uint4(d[0], d[1], d[2], d[3]);
  };
  // The LDS-DMA is issued from inline asm, invisible to the compiler: as a builtin it is a pending LDS write that the
  // waitcnt pass cannot tell apart from the other stage (one array, dynamic index), so every fragment read waited for
  // vmcnt(0); with two statically selected LDS objects (loop unrolled by two) that went away, but the two copies of the
  // MFMA stream got two accumulator register sets and 256 v_accvgpr moves per chunk.  Hidden, it costs nothing: the
  // explicit vmcnt(0) in front of the barrier that publishes the stage is the only wait it needs.
  auto load_weight_piece = [&](int q, uint32_t wdst_lds) __attribute__((always_inline)) {
    const int v = tid + q * NT;                    // (v < WV is wave-uniform: WV is a multiple of 64)
    if (WV % NT == 0 || v < WV) {
      const int t = v / (MT * 64), rem = v - t * (MT * 64);
      const cv_f16x8* g = cwsrc + (long)t * ctiles * 64 + rem;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(wdst_lds + (uint32_t)(q * NT + wv * 64) * 16u);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
  };
  auto store_chunk = [&](_Float16* sdst) __attribute__((always_inline)) {
#pragma clang loop unroll(full)
    for (int q = 0; q < NPS; q++)
      if (s_off[q] >= 0) *reinterpret_cast<uint4*>(sdst + s_off[q]) = g_ok[q] ? ps[q] : make_uint4(0, 0, 0, 0);
  };

  // B-fragment base offsets (halves) of this lane's pixels for tap (0,0)
  int boff[UT];
#pragma unroll
  for (int u = 0; u < UT; u++) {
    const int row = 2 * UT * pg + 2 * u + (j >> 4), col = j & 15;
    boff[u] = row * RSH + col * CV_SP + 8 * h;
  }
  // One chunk: 9 taps x (UT B fragments + MT A fragments -> UT MT MFMAs), software-pipelined by one tap, with the NEXT
  // chunk's loads (LDS-DMA of the weights, slab pieces into registers) issued one per MFMA in the same stream: with one
  // wave per SIMD nothing else hides an instruction's issue or latency, so everything rides in the shadow of the MFMAs.
  auto compute = [&](const _Float16* sl, const cv_f16x8* w, auto more, uint32_t wdst) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more)::value;
    cv_f16x8 bf[2][UT], af[2][MTW];
#pragma unroll
    for (int u = 0; u < UT; u++) bf[0][u] = *reinterpret_cast<const cv_f16x8*>(sl + boff[u]);
#pragma unroll
    for (int m = 0; m < MTW; m++) af[0][m] = w[(cg * MTW + m) * 64 + lane];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < T) {
        const int toff = ((t + 1) / KS) * RSH + ((t + 1) % KS) * CV_SP;
#pragma unroll
        for (int u = 0; u < UT; u++) bf[nxt][u] = *reinterpret_cast<const cv_f16x8*>(sl + boff[u] + toff);
#pragma unroll
        for (int m = 0; m < MTW; m++) af[nxt][m] = w[((t + 1) * MT + cg * MTW + m) * 64 + lane];
      }
      // this tap's share of the next chunk's loads
      constexpr int TW = T < CV_WTAPS ? T : CV_WTAPS;                            // weights: spread over the first TW taps
      const int tw0 = t < TW ? t : TW, tw1 = t + 1 < TW ? t + 1 : TW;
      const int w0 = tw0 * NPW / TW, w1 = tw1 * NPW / TW;
      const int s0 = t < NPS ? t : NPS, s1 = (T == 1) ? NPS : (t + 1 < NPS ? t + 1 : NPS);   // slab: first taps (longest time to land)
      if (MORE) {
#pragma unroll
        for (int q = w0; q < w1; q++) load_weight_piece(q, wdst);
#pragma unroll
        for (int q = s0; q < s1; q++) load_slab_piece(q);
      }
#pragma unroll
      for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int u = 0; u < UT; u++)
          acc[m][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][m], bf[cur][u], acc[m][u], 0, 0, 0);
      // issue order inside the tap: MFMA, one memory instruction, MFMA, ...  (round 5, measured: 1, 3 or all 6 of the next tap's
      // fragment reads behind each MFMA instead of 2 -- 512-535 us on the 448 -> 256 gate in every arm, inside the run-to-run spread)
      const int nds = (t + 1 < T) ? UT + MTW : 0, nvm = MORE ? (w1 - w0) + (s1 - s0) : 0;
#pragma unroll
      for (int i = 0; i < MTW * UT; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA
        if (2 * i < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS reads (up to 2 per MFMA)
        if (2 * i + 1 < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < nvm) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM
      }
    }
  };
  const uint32_t wl_lds0 = (uint32_t)(uintptr_t)wl0;
  chunk_source(0);
#pragma clang loop unroll(full)
  for (int q = 0; q < NPW; q++) load_weight_piece(q, wl_lds0);
#pragma clang loop unroll(full)
  for (int q = 0; q < NPS; q++) load_slab_piece(q);
  store_chunk(slab0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < nchunk; c++) {
    const int b = c & 1;
    // (the last chunk prefetches itself again into the idle stage instead of branching around the loads)
    chunk_source(c + 1 < nchunk ? c + 1 : c);
    compute(slab0 + b * SLAB_H, wl0 + b * WV, cv_bool<true>(), wl_lds0 + (uint32_t)((b ^ 1) * WV * 16));
    __builtin_amdgcn_sched_barrier(0);
    store_chunk(slab0 + (b ^ 1) * SLAB_H);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of the next stage has landed
    __syncthreads();
  }

  // ---- epilogue: bias + activation in registers, then through a wave-private LDS tile so that the global stores are
  // whole rows of the tile's couts (64 MT contiguous bytes per pixel) instead of 8-byte pieces 2 couts-rows apart: the
  // direct stores were 28 % of the kernel's time (32 store instructions of 64 scattered pieces per wave) ----
  const float* bp = a.bias ? a.bias + (long)n * a.bias_nstride : nullptr;
  const bool full = a.vec && (cz * MT + MT) * 32 <= a.CO;        // workgroup-uniform
  if (full) {
    _Float16* et = reinterpret_cast<_Float16*>(lds_raw) + wv * 32 * ERS;
    const int cobase = (cz * MT + cg * MTW) * 32;       // first cout of this wave
    const bool bvec = bp && (reinterpret_cast<uintptr_t>(bp) & 15) == 0;
#pragma clang loop unroll(full)
    for (int u = 0; u < UT; u++) {
#pragma clang loop unroll(full)
      for (int m = 0; m < MTW; m++) {
#pragma clang loop unroll(full)
        for (int g = 0; g < 4; g++) {
          const int co = cobase + m * 32 + 8 * g + 4 * h;
          float4 b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (bvec) b4 = *reinterpret_cast<const float4*>(bp + co);
          else if (bp) b4 = make_float4(bp[co], bp[co + 1], bp[co + 2], bp[co + 3]);
          cv_f16x4 o = {(_Float16)cv_act(acc[m][u][4 * g] + b4.x, a.act), (_Float16)cv_act(acc[m][u][4 * g + 1] + b4.y, a.act),
                        (_Float16)cv_act(acc[m][u][4 * g + 2] + b4.z, a.act), (_Float16)cv_act(acc[m][u][4 * g + 3] + b4.w, a.act)};
          *reinterpret_cast<cv_f16x4*>(et + j * ERS + m * 32 + 8 * g + 4 * h) = o;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the accumulator reads of later tiles from being hoisted (spills)
      }
      // read back: MTW 8 lanes per pixel row (8 bytes each), 64 / (8 MTW) pixels per instruction
      constexpr int LPR = MTW * 8, PPI = 64 / LPR;
#pragma clang loop unroll(full)
      for (int i = 0; i < 32 / PPI; i++) {
        const int pj = i * PPI + lane / LPR, l = lane % LPR;
        cv_f16x4 v = *reinterpret_cast<const cv_f16x4*>(et + pj * ERS + 4 * l);
        const int y = y0 + 2 * UT * pg + 2 * u + (pj >> 4), x = x0 + (pj & 15);
        if (y < a.H && x < a.W) {
          const long pix = ((long)n * a.H + y) * a.W + x;
          const int co = cobase + 4 * l;
          if (a.fuse == NS_CONV_FUSE_MUL_HI) {          // r * h next to z (gru.py:29-30): couts >= CO/2 are multiplied
            if (co >= (a.CO >> 1)) {
              const cv_f16x4 f = *reinterpret_cast<const cv_f16x4*>(a.e0 + pix * a.e0s + co - (a.CO >> 1));
#pragma unroll
              for (int e = 0; e < 4; e++) v[e] = (_Float16)((float)v[e] * (float)f[e]);
            }
          } else if (a.fuse == NS_CONV_FUSE_GRU) {      // (1 - z) h + z q (gru.py:31-33)
            const cv_f16x4 z = *reinterpret_cast<const cv_f16x4*>(a.e0 + pix * a.e0s + co);
            const cv_f16x4 hh = *reinterpret_cast<const cv_f16x4*>(a.e1 + pix * a.e1s + co);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (_Float16)((float)hh[e] + (float)z[e] * ((float)v[e] - (float)hh[e]));
          }
          *reinterpret_cast<cv_f16x4*>(a.out + pix * a.ostride + a.ooff + co) = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // ragged cout tile or unaligned slice: direct stores
#pragma clang loop unroll(full)
  for (int u = 0; u < UT; u++) {
    const int y = y0 + 2 * UT * pg + 2 * u + (j >> 4), x = x0 + (j & 15);
    const bool pv = y < a.H && x < a.W;
    _Float16* op = a.out + (((long)n * a.H + y) * a.W + x) * a.ostride + a.ooff;
#pragma clang loop unroll(full)
    for (int m = 0; m < MTW; m++) {
#pragma clang loop unroll(full)
      for (int g = 0; g < 4; g++) {
        const int co = (cz * MT + cg * MTW + m) * 32 + 8 * g + 4 * h;
        float v0 = acc[m][u][4 * g], v1 = acc[m][u][4 * g + 1], v2 = acc[m][u][4 * g + 2], v3 = acc[m][u][4 * g + 3];
        if (pv && co + 3 < a.CO && a.vec) {
          if (bp) {
            v0 += bp[co]; v1 += bp[co + 1]; v2 += bp[co + 2]; v3 += bp[co + 3];
          }
          cv_f16x4 o = {(_Float16)cv_act(v0, a.act), (_Float16)cv_act(v1, a.act), (_Float16)cv_act(v2, a.act),
                        (_Float16)cv_act(v3, a.act)};
          *reinterpret_cast<cv_f16x4*>(op + co) = o;
        } else if (pv && co < a.CO) {  // ragged tail of a cout count that is not a multiple of 4, or unaligned slice
          op[co] = (_Float16)cv_act(v0 + (bp ? bp[co] : 0.0f), a.act);
          if (co + 1 < a.CO) op[co + 1] = (_Float16)cv_act(v1 + (bp ? bp[co + 1] : 0.0f), a.act);
          if (co + 2 < a.CO) op[co + 2] = (_Float16)cv_act(v2 + (bp ? bp[co + 2] : 0.0f), a.act);
          if (co + 3 < a.CO) op[co + 3] = (_Float16)cv_act(v3 + (bp ? bp[co + 3] : 0.0f), a.act);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// cout tile of the workgroup that ns_conv_nhwc_f16 uses for `cout` output channels
static int cv_cout_tile(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

extern "C" int ns_conv_packed_cout(int cout) {
  const int t = cv_cout_tile(cout);
  return (cout + t - 1) / t * t;
}

template <int KS, int MT, int UT, int CG>
static void cv_launch(ConvArgs a, hipStream_t st) {
  a.tiles_y = ns_cdiv(a.H, 8 * UT);
  a.tiles_x = ns_cdiv(a.W, CV_TC);
  const long tiles = (long)a.tiles_x * a.tiles_y * a.N, nz = a.COP / (32 * MT);
  dim3 grid((unsigned)((tiles + 7) / 8 * 8 * nz));
  hipLaunchKernelGGL((conv_nhwc_kernel<KS, MT, UT, CG>), grid, dim3(256 * CG), 0, st, a);
}

static int cv_run(const void* const* src_host, const int* src_channels_host, const int* src_strides_host, int nsrc, int N, int H,
                  int W, const void* wpacked, int ksize, int cout, const float* bias, long bias_nstride, int act, void* out,
                  int out_stride, int out_offset, int fuse, const void* e0, int e0_stride, const void* e1, int e1_stride,
                  void* stream) {
  if (N == 0) return NS_OK;
  NS_REQUIRE(src_host && src_channels_host && wpacked && out, "ns_conv_nhwc_f16: null pointer");
  NS_REQUIRE(nsrc >= 1 && nsrc <= CV_MAXSRC, "ns_conv_nhwc_f16: %d sources (1..%d)", nsrc, CV_MAXSRC);
  NS_REQUIRE(N > 0 && H > 0 && W > 0 && cout > 0, "ns_conv_nhwc_f16: bad shape");
  NS_REQUIRE(ksize == 1 || ksize == 3, "ns_conv_nhwc_f16: kernel size %d unsupported (1 or 3)", ksize);
  NS_REQUIRE(act >= NS_ACT_NONE && act <= NS_ACT_TANH, "ns_conv_nhwc_f16: activation %d unknown", act);
  NS_REQUIRE(out_offset >= 0 && out_offset + cout <= out_stride, "ns_conv_nhwc_f16: output slice [%d, %d) leaves the row of %d",
             out_offset, out_offset + cout, out_stride);
  ConvArgs a;
  a.nsrc = nsrc;
  a.src_start[0] = 0;
  for (int s = 0; s < CV_MAXSRC; s++) {
    a.src[s] = nullptr;
    a.src_ch[s] = 0;
    a.src_stride[s] = 0;
    a.src_start[s + 1] = a.src_start[s];
    if (s < nsrc) {
      NS_REQUIRE(src_host[s] && src_channels_host[s] > 0 && src_channels_host[s] % 16 == 0,
                 "ns_conv_nhwc_f16: source %d needs a multiple of 16 channels (got %d)", s, src_channels_host[s]);
      const int stride = src_strides_host ? src_strides_host[s] : src_channels_host[s];
      NS_REQUIRE(stride >= src_channels_host[s] && stride % 8 == 0 && ((uintptr_t)src_host[s] % 16) == 0,
                 "ns_conv_nhwc_f16: source %d: pixel stride %d / base address must keep 16-byte alignment", s, stride);
      a.src[s] = (const _Float16*)src_host[s];
      a.src_ch[s] = src_channels_host[s];
      a.src_stride[s] = stride;
      a.src_start[s + 1] = a.src_start[s] + src_channels_host[s];
    }
  }
  a.N = N; a.H = H; a.W = W;
  a.CI = a.src_start[nsrc];
  a.CO = cout;
  a.COP = ns_conv_packed_cout(cout);
  a.wp = (const cv_f16x8*)wpacked;
  a.bias = bias;
  a.bias_nstride = bias_nstride;
  a.out = (_Float16*)out;
  a.ostride = out_stride;
  a.ooff = out_offset;
  a.act = act;
  a.vec = (out_stride % 4 == 0 && out_offset % 4 == 0 && ((uintptr_t)out % 8) == 0) ? 1 : 0;
  a.fuse = fuse;
  a.e0 = (const _Float16*)e0;
  a.e1 = (const _Float16*)e1;
  a.e0s = e0_stride;
  a.e1s = e1_stride;
  if (fuse != NS_CONV_FUSE_NONE) {
    NS_REQUIRE(fuse == NS_CONV_FUSE_MUL_HI || fuse == NS_CONV_FUSE_GRU, "ns_conv_nhwc_f16_fused: mode %d unknown", fuse);
    // the fused step lives on the whole-row store path: full 4-aligned cout tiles only
    NS_REQUIRE(a.vec && cout % cv_cout_tile(cout) == 0, "ns_conv_nhwc_f16_fused: needs an aligned output slice and cout a multiple of %d",
               cv_cout_tile(cout));
    NS_REQUIRE(e0 && e0_stride % 4 == 0 && ((uintptr_t)e0 % 8) == 0, "ns_conv_nhwc_f16_fused: operand 0 missing or misaligned");
    if (fuse == NS_CONV_FUSE_GRU)
      NS_REQUIRE(e1 && e1_stride % 4 == 0 && ((uintptr_t)e1 % 8) == 0, "ns_conv_nhwc_f16_fused: operand 1 missing or misaligned");
    else
      NS_REQUIRE((cout / 2) % 4 == 0, "ns_conv_nhwc_f16_fused: cout/2 must be a multiple of 4");
  }
  a.tiles_y = 0;
  int mt = cv_cout_tile(cout) / 32;
  NS_REQUIRE((long)N * ns_cdiv(H, 16) * ns_cdiv(W, 16) * (ns_conv_packed_cout(cout) / 32) < (1L << 31), "ns_conv_nhwc_f16: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  const char* cg_env = ns_variant_env("NS_CONV_CG");        // 1 | 2 waves per SIMD for the 128-cout tile (experiments)
  const bool two = cg_env ? atoi(cg_env) == 2 : true;
  // 3x3, 128-cout tile, images taller than one 16-row tile: 32-row tiles (4 column tiles per wave, 128 accumulator
  // registers, still 2 waves per SIMD) -- twice the MFMAs per barrier and per LDS-DMA: 865 / 968 vs 791 / 872 TFLOP/s on
  // the two ConvGRU gates; the 1x1 launches are load-bound and lose (61 vs 49 us).  NS_CONV_UT=2|4 overrides (tests).
  const char* ut_env = ns_variant_env("NS_CONV_UT");
  bool tall = H > 16 && (ut_env ? atoi(ut_env) == 4 : ksize == 3);
  // Few images (the motion filter's single edge, the encoders' N = 1): a 60x80 map is 10 of the 32-row x 128-cout tiles -- 10
  // workgroups on 256 CUs, each walking the whole K loop alone (36 us for 448 -> 256).  Shrink the tile until the launch
  // has ~200 workgroups: first 16-row tiles, then 64- and 32-cout tiles (the packed weight layout does not depend on the
  // cout tile: [chunk][tap][COP / 32][lane]).  The slab is then staged once per cout tile, which is latency well spent
  // here and bandwidth wasted for E = 48 -- that case never gets here.  NS_CONV_MT = 1|2|4 forces the cout tile (tests).
  const char* mt_env = ns_variant_env("NS_CONV_MT");
  auto wgs = [&](int m, bool t) { return (long)N * ns_cdiv(H, t ? 32 : 16) * ns_cdiv(W, CV_TC) * (a.COP / (32 * m)); };
  if (mt_env) {
    const int f = atoi(mt_env);
    if ((f == 1 || f == 2 || f == 4) && f <= mt) mt = f;
  } else {
    const long want = 200;
    if (!ut_env && tall && wgs(mt, true) < want) tall = false;
    while (mt > 1 && wgs(mt, tall && mt == 4) < want) mt >>= 1;
  }
  // (the product library holds the seven tilings its own dispatch reaches; the 32-row 1x1 tile and the one-wave-per-SIMD
  //  128-cout tiles exist for NS_CONV_UT / NS_CONV_CG in the variants library only)
  if (mt == 4 && two && tall) {
    if (ksize == 3) cv_launch<3, 4, 4, 2>(a, st);
#ifdef NS_TEST_VARIANTS
    else cv_launch<1, 4, 4, 2>(a, st);
#else
    else cv_launch<1, 4, 2, 2>(a, st);      // (unreachable: tall implies ksize == 3 without NS_CONV_UT)
#endif
  } else if (ksize == 3) {
    if (mt == 4 && two) cv_launch<3, 4, 2, 2>(a, st);
#ifdef NS_TEST_VARIANTS
    else if (mt == 4) cv_launch<3, 4, 2, 1>(a, st);
#endif
    else if (mt == 2) cv_launch<3, 2, 2, 1>(a, st);
    else cv_launch<3, 1, 2, 1>(a, st);
  } else {
    if (mt == 4 && two) cv_launch<1, 4, 2, 2>(a, st);
#ifdef NS_TEST_VARIANTS
    else if (mt == 4) cv_launch<1, 4, 2, 1>(a, st);
#endif
    else if (mt == 2) cv_launch<1, 2, 2, 1>(a, st);
    else cv_launch<1, 1, 2, 1>(a, st);
  }
  NS_CHECK_LAUNCH("conv_nhwc_kernel");
  return NS_OK;
}

extern "C" int ns_conv_nhwc_f16(const void* const* src_host, const int* src_channels_host, const int* src_strides_host, int nsrc,
                                int N, int H, int W, const void* wpacked, int ksize, int cout, const float* bias,
                                long bias_nstride, int act, void* out, int out_stride, int out_offset, void* stream) {
  return cv_run(src_host, src_channels_host, src_strides_host, nsrc, N, H, W, wpacked, ksize, cout, bias, bias_nstride, act, out,
                out_stride, out_offset, NS_CONV_FUSE_NONE, nullptr, 0, nullptr, 0, stream);
}

extern "C" int ns_conv_nhwc_f16_fused(const void* const* src_host, const int* src_channels_host, const int* src_strides_host,
                                      int nsrc, int N, int H, int W, const void* wpacked, int ksize, int cout, const float* bias,
                                      long bias_nstride, int act, void* out, int out_stride, int out_offset, int fuse,
                                      const void* e0, int e0_stride, const void* e1, int e1_stride, void* stream) {
  return cv_run(src_host, src_channels_host, src_strides_host, nsrc, N, H, W, wpacked, ksize, cout, bias, bias_nstride, act, out,
                out_stride, out_offset, fuse, e0, e0_stride, e1, e1_stride, stream);
}

// ---------------------------------------------------------------------------------------------
// im2col of the flow encoder's 7x7 convolution (networks/droid_net.py:96: Conv2d(4, 128, 7, padding=3) over the motion
// features [E,4,ht,wd] f32): out[e,y,x, k] = flow[e, ci, y+ky-3, x+kx-3] for k = (ci*7 + ky)*7 + kx < 196 (the order of
// weight.reshape(128, 196)), zero outside the image and for the 12 pad channels -> the convolution becomes a 1x1
// ns_conv_nhwc_f16 over 208 channels on the MFMA units.  (MIOpen's immediate mode picks a naive kernel for this shape in
// channels-last: 10.8 ms per call at E=48.)  One thread per 16-byte piece: 26 pieces per pixel, stores fully coalesced.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_im2col_kernel(const float* __restrict__ flow, _Float16* __restrict__ out, long npix,
                                                          int ht, int wd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * 26) return;
  const long pix = i / 26;
  const int piece = (int)(i - pix * 26);
  const int hw = ht * wd;
  const long e = pix / hw;
  const int p = (int)(pix - e * hw), y = p / wd, x = p - y * wd;
  const float* f = flow + e * 4 * hw;
  cv_f16x8 v;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int k = piece * 8 + q;
    const int ci = k / 49, r = k - ci * 49, ky = r / 7, kx = r - ky * 7;
    const int yy = y + ky - 3, xx = x + kx - 3;
    const bool ok = k < 196 && yy >= 0 && yy < ht && xx >= 0 && xx < wd;
    v[q] = ok ? (_Float16)f[ci * hw + yy * wd + xx] : (_Float16)0.0f;
  }
  *reinterpret_cast<cv_f16x8*>(out + pix * 208 + piece * 8) = v;
}

extern "C" int ns_flow_im2col(const float* flow, void* out, int E, int ht, int wd, void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(flow && out, "ns_flow_im2col: null pointer");
  NS_REQUIRE(E > 0 && ht > 0 && wd > 0, "ns_flow_im2col: bad shape");
  const long npix = (long)E * ht * wd;
  hipLaunchKernelGGL(flow_im2col_kernel, dim3(ns_cdiv(npix * 26, 256)), dim3(256), 0, (hipStream_t)stream, flow, (_Float16*)out,
                     npix, ht, wd);
  NS_CHECK_LAUNCH("flow_im2col_kernel");
  return NS_OK;
}

// ---------------------------------------------------------------------------------------------
// Layout glue of the update operator.
//
// planes_to_nhwc: the lookup's output [E, C, HW] f16 (C = 196) -> channels-last [E, HW, CP] (CP = 208, pad channels zero)
// through an LDS tile of 64 pixels: global reads are 128-byte runs of one channel plane (two pixels per lane), global
// writes 16-byte pieces of 416-byte pixel rows.  (torch: permute + pad = two passes, 128 us for E=48; here one.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void planes_to_nhwc_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int C,
                                                             int CP, int HW) {
  extern __shared__ __attribute__((aligned(16))) _Float16 tile[];   // [64][CP + 8]
  const int RS = CP + 8;
  const int e = blockIdx.y, p0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const _Float16* sp = src + (long)e * C * HW;
  // zero the pad channels
  for (int i = tid; i < 64 * (CP - C); i += 256) tile[(i / (CP - C)) * RS + C + i % (CP - C)] = (_Float16)0.0f;
  if ((HW & 1) == 0) {           // two pixels per lane: a wave instruction covers two channel rows of 64 pixels
    const int half = lane >> 5, px = (lane & 31) * 2;
    for (int c = wv * 2 + half; c < C; c += 8) {
      uint32_t w = 0;
      if (p0 + px < HW) w = *reinterpret_cast<const uint32_t*>(sp + (long)c * HW + p0 + px);
      tile[px * RS + c] = __builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
      tile[(px + 1) * RS + c] = __builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    }
  } else {
    for (int c = wv; c < C; c += 4) tile[lane * RS + c] = (p0 + lane < HW) ? sp[(long)c * HW + p0 + lane] : (_Float16)0.0f;
  }
  __syncthreads();
  const int ppr = CP / 8;        // 16-byte pieces per pixel row
  for (int i = tid; i < 64 * ppr; i += 256) {
    const int px = i / ppr, piece = i - px * ppr;
    if (p0 + px < HW)
      *reinterpret_cast<uint4*>(dst + ((long)e * HW + p0 + px) * CP + piece * 8) = *reinterpret_cast<const uint4*>(tile + px * RS + piece * 8);
  }
}

extern "C" int ns_planes_to_nhwc_f16(const void* src, void* dst, int E, int C, int CP, int HW, void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(src && dst, "ns_planes_to_nhwc_f16: null pointer");
  NS_REQUIRE(E > 0 && C > 0 && HW > 0 && CP >= C && CP % 8 == 0 && CP <= 1024, "ns_planes_to_nhwc_f16: bad shape (C %d, CP %d)", C, CP);
  NS_REQUIRE(E <= 65535, "ns_planes_to_nhwc_f16: too many images");
  hipLaunchKernelGGL(planes_to_nhwc_kernel, dim3(ns_cdiv(HW, 64), E), dim3(256), 64 * (CP + 8) * sizeof(_Float16), (hipStream_t)stream,
                     (const _Float16*)src, (_Float16*)dst, C, CP, HW);
  NS_CHECK_LAUNCH("planes_to_nhwc_kernel");
  return NS_OK;
}

// ---------------------------------------------------------------------------------------------
// GraphAgg's scatter-mean (networks/droid_net.py:64-70: torch_scatter.scatter_mean over the source keyframe of every edge):
// out[k, p, c] = mean over the edges e in group k of src[e, p, c].  Groups as CSR (starts [K+1], members [E], built on
// the host from the edge list).  One thread per (group, pixel, 8-channel piece): every input element is read once, f32 sums.
// (torch: zeros + index_add_ + divide = 125 us + 2 passes for E=48.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_mean_kernel(const _Float16* __restrict__ src, int sstride, const int* __restrict__ starts,
                                                         const int* __restrict__ members, _Float16* __restrict__ out, int K, long HW,
                                                         int C8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)K * HW * C8) return;
  const int piece = (int)(i % C8);
  const long kp = i / C8, p = kp % HW;
  const int k = (int)(kp / HW);
  const int e0 = starts[k], e1 = starts[k + 1];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = e0; m < e1; m++) {
    const cv_f16x8 v = *reinterpret_cast<const cv_f16x8*>(src + ((long)members[m] * HW + p) * sstride + piece * 8);
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] += (float)v[q];
  }
  const float inv = e1 > e0 ? 1.0f / (float)(e1 - e0) : 0.0f;
  cv_f16x8 o;
#pragma unroll
  for (int q = 0; q < 8; q++) o[q] = (_Float16)(acc[q] * inv);
  *reinterpret_cast<cv_f16x8*>(out + kp * (C8 * 8) + piece * 8) = o;
}

extern "C" int ns_group_mean_nhwc_f16(const void* src, int src_stride, const int* starts, const int* members, void* out, int K,
                                      int HW, int channels, void* stream) {
  if (K == 0) return NS_OK;
  NS_REQUIRE(src && starts && members && out, "ns_group_mean_nhwc_f16: null pointer");
  NS_REQUIRE(K > 0 && HW > 0 && channels > 0 && channels % 8 == 0 && src_stride >= channels && src_stride % 8 == 0 &&
                 ((uintptr_t)src % 16) == 0,
             "ns_group_mean_nhwc_f16: bad shape / alignment (channels %d, stride %d)", channels, src_stride);
  const long total = (long)K * HW * (channels / 8);
  hipLaunchKernelGGL(group_mean_kernel, dim3(ns_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, src_stride,
                     starts, members, (_Float16*)out, K, (long)HW, channels / 8);
  NS_CHECK_LAUNCH("group_mean_kernel");
  return NS_OK;
}

// ---------------------------------------------------------------------------------------------
// ConvGRU global context (networks/modules/gru.py:25-33): glo = mean over the pixels of sigmoid(w(net)) * net, a per-edge vector
// [E,128]; the three conv*_glo 1x1 convolutions of it are glo @ W [128,384] + b = per-image biases of the gate convolutions
// (nerfslam/update_op.py).  torch ran this as an elementwise product (a third [E,HW,128] tensor written and read back), a
// reduction and a hipBLASLt GEMM whose 42 us are launch latency: ~90 us per update for 100 kflop of matrix product.
//   glo_partial: grid (P, E); thread = (pixel group, 16-byte piece of the 128 channels); f32 products and sums;
//                partial[e][p][c] = sum over part p's pixels
//   glo_finish : grid E, 384 threads: glo[c] = sum_p partial / HW (fixed order), out[e][o] = b[o] + sum_c glo[c] W[c][o]
// ---------------------------------------------------------------------------------------------
#define GLO_C 128
__global__ __launch_bounds__(256) void glo_partial_kernel(const _Float16* __restrict__ wg, const _Float16* __restrict__ net,
                                                          float* __restrict__ partial, int HW, int P) {
  __shared__ float red[16][GLO_C];
  const int tid = threadIdx.x, piece = tid & 15, g = tid >> 4;     // 16 pieces x 16 pixel groups
  const int p = blockIdx.x, e = blockIdx.y;
  const int chunk = (HW + P - 1) / P;
  const int lo = p * chunk, hi = min(HW, lo + chunk);
  const long base = (long)e * HW * GLO_C + piece * 8;
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; q++) s[q] = 0.0f;
  for (int pix = lo + g; pix < hi; pix += 32) {                    // two pixels per iteration: four 16-byte loads in flight
    cv_f16x8 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int px = pix + 16 * u;
      const bool ok = px < hi;
      a[u] = ok ? *reinterpret_cast<const cv_f16x8*>(wg + base + (long)px * GLO_C) : (cv_f16x8)(_Float16)0;
      b[u] = ok ? *reinterpret_cast<const cv_f16x8*>(net + base + (long)px * GLO_C) : (cv_f16x8)(_Float16)0;
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int q = 0; q < 8; q++) s[q] += (float)a[u][q] * (float)b[u][q];
  }
#pragma unroll
  for (int q = 0; q < 8; q++) red[g][piece * 8 + q] = s[q];
  __syncthreads();
  if (tid < GLO_C) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][tid];
    partial[((long)e * P + p) * GLO_C + tid] = t;
  }
}

__global__ __launch_bounds__(384) void glo_finish_kernel(const float* __restrict__ partial, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out, int P, float inv_hw,
                                                         int nout) {
  __shared__ float glo[GLO_C];
  const int e = blockIdx.x, tid = threadIdx.x;
  if (tid < GLO_C) {
    float t = 0.0f;
    for (int p = 0; p < P; p++) t += partial[((long)e * P + p) * GLO_C + tid];
    glo[tid] = t * inv_hw;
  }
  __syncthreads();
  for (int o = tid; o < nout; o += 384) {
    float acc = bias ? bias[o] : 0.0f;
#pragma unroll 8
    for (int c = 0; c < GLO_C; c++) acc += glo[c] * W[(long)c * nout + o];
    out[(long)e * nout + o] = acc;
  }
}

extern "C" int ns_gru_glo_parts(int HW) {
  const int p = (HW + 511) / 512;
  return p < 1 ? 1 : (p > 32 ? 32 : p);
}

extern "C" int ns_gru_glo_bias(const void* wg, const void* net, const float* W, const float* bias, float* partial, float* out, int E,
                               int HW, int nout, void* stream) {
  if (E == 0) return NS_OK;
  NS_REQUIRE(wg && net && W && partial && out, "ns_gru_glo_bias: null pointer");
  NS_REQUIRE(E > 0 && E <= 65535 && HW > 0 && nout > 0, "ns_gru_glo_bias: bad shape");
  NS_REQUIRE(((uintptr_t)wg % 16) == 0 && ((uintptr_t)net % 16) == 0, "ns_gru_glo_bias: 16-byte alignment");
  const int P = ns_gru_glo_parts(HW);
  hipLaunchKernelGGL(glo_partial_kernel, dim3(P, E), dim3(256), 0, (hipStream_t)stream, (const _Float16*)wg, (const _Float16*)net,
                     partial, HW, P);
  NS_CHECK_LAUNCH("glo_partial_kernel");
  hipLaunchKernelGGL(glo_finish_kernel, dim3(E), dim3(384), 0, (hipStream_t)stream, (const float*)partial, W, bias, out, P,
                     1.0f / (float)HW, nout);
  NS_CHECK_LAUNCH("glo_finish_kernel");
  return NS_OK;
}
