// corr_build.hip -- correlation-volume pyramid construction for gfx950.
//
// CorrBlock.__init__ of the reference (networks/modules/corr.py:23-38) builds the level-0 volume with
// torch.matmul and then runs three F.avg_pool2d(2,2) passes over the (h2,w2) plane of every
// (edge, pixel) slice.  ns_corr_pool2x2 is one such pass as a streaming HBM-bound kernel:
// f16 in, f32 accumulate in (row,col) order, one rounding (ATen's avg_pool2d for Half uses a float
// accumulator and casts once).
#include "common.h"

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));

// fast path: w % 4 == 0.  One lane -> two adjacent outputs (8 B from each of two rows, 4 B out).
__global__ __launch_bounds__(256) void corr_pool2x2_vec_kernel(const _Float16* __restrict__ in,
                                                               _Float16* __restrict__ out, long npairs, int ho,
                                                               int wo2, int w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npairs) return;
  const int xp = (int)(idx % wo2);
  const long t = idx / wo2;
  const int y = (int)(t % ho);
  const long s = t / ho;
  const long hin = (long)ho * 2;
  const _Float16* r0 = in + (s * hin + 2 * y) * w + 4 * xp;
  const h4_t a = *reinterpret_cast<const h4_t*>(r0);
  const h4_t b = *reinterpret_cast<const h4_t*>(r0 + w);
  float s0 = 0.0f, s1 = 0.0f;
  s0 += (float)a[0];
  s0 += (float)a[1];
  s0 += (float)b[0];
  s0 += (float)b[1];
  s1 += (float)a[2];
  s1 += (float)a[3];
  s1 += (float)b[2];
  s1 += (float)b[3];
  h2v_t o = {(_Float16)(s0 / 4.0f), (_Float16)(s1 / 4.0f)};
  *reinterpret_cast<h2v_t*>(out + (s * ho + y) * (long)(wo2 * 2) + 2 * xp) = o;
}

// generic path (any h, w; floor semantics of avg_pool2d: trailing odd row/column dropped)
__global__ __launch_bounds__(256) void corr_pool2x2_kernel(const _Float16* __restrict__ in,
                                                           _Float16* __restrict__ out, long nout, int h, int w,
                                                           int ho, int wo) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nout) return;
  const int x = (int)(idx % wo);
  const long t = idx / wo;
  const int y = (int)(t % ho);
  const long s = t / ho;
  const _Float16* r0 = in + (s * h + 2 * y) * w + 2 * x;
  float acc = 0.0f;
  acc += (float)r0[0];
  acc += (float)r0[1];
  acc += (float)r0[w];
  acc += (float)r0[w + 1];
  out[idx] = (_Float16)(acc / 4.0f);
}

extern "C" int ns_corr_pool2x2(const void* in, void* out, long nslices, int h, int w, void* stream) {
  NS_REQUIRE(in && out, "ns_corr_pool2x2: null pointer");
  NS_REQUIRE(nslices >= 0 && h > 0 && w > 0, "ns_corr_pool2x2: bad shape");
  const int ho = h / 2, wo = w / 2;
  if (nslices == 0 || ho == 0 || wo == 0) return NS_OK;
  if (w % 4 == 0 && h % 2 == 0) {
    const long npairs = nslices * ho * (wo / 2);
    hipLaunchKernelGGL(corr_pool2x2_vec_kernel, dim3(ns_cdiv(npairs, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)in, (_Float16*)out, npairs, ho, wo / 2, w);
  } else {
    const long nout = nslices * ho * wo;
    hipLaunchKernelGGL(corr_pool2x2_kernel, dim3(ns_cdiv(nout, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)in, (_Float16*)out, nout, h, w, ho, wo);
  }
  NS_CHECK_LAUNCH("corr_pool2x2_kernel");
  return NS_OK;
}
