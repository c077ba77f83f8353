// ngp_adam.h -- what the NeRF trainer's kernels in ngp.hip and ngp_mlp.hip share: the device control block of the
// graph-captured step and the ONE definition of the Adam update.
#pragma once
#include <hip/hip_runtime.h>

// Device control block of the graph-captured training step (nerfslam/ngp.py): the values a step needs from the
// previous one live in device memory, so that a step is a fixed sequence of launches with fixed arguments.
//   ctl[0] optimiser steps completed   ctl[1] rays of the current batch   ctl[2] ray-sampling seed   ctl[3] training views
// A kernel given `ctl` reads its ray count / step / view count from it (bounded by the by-value argument, which then
// is the CAPACITY its grid was sized for); ctl == nullptr keeps the by-value behaviour.
#define NS_CTL_STEP 0
#define NS_CTL_RAYS 1
#define NS_CTL_SEED 2
#define NS_CTL_VIEWS 3
#define NS_CTL_C1 4  // float bits: 1 - beta1^(ctl[0] + 1), 1 - beta2^(ctl[0] + 1) of the model's Adam (written by ns_ngp_step_advance)
#define NS_CTL_C2 5

// One Adam update of one parameter (tiny-cuda-nn semantics; the caller skips zero-gradient entries when there is no weight
// decay).  ONE definition, inlined into the streaming pass (ngp_adam_kernel) and into the fused flushes of the table
// gradient (ngp_enc_faccum_kernel / ngp_enc_dense_reduce_kernel): the same expression tree, hence the same contractions
// and the same bits from either path (tests/test_ngp_gpu.py::test_fused_table_gradient_adam_is_bit_identical).
__device__ __forceinline__ float adam_apply(float p, float g, float l2, float& m1, float& m2, float c1, float c2, float lr,
                                            float beta1, float beta2, float eps) {
  // every operation pinned (no contraction left to the compiler: the two call sites contracted `beta * m + (1 - beta) * g`
  // differently and their parameters drifted apart by an ulp from the second step on)
  g = __fmaf_rn(l2, p, g);
  const float a = __fmaf_rn(beta1, m1, __fmul_rn(1.0f - beta1, g));
  const float b = __fmaf_rn(beta2, m2, __fmul_rn(__fmul_rn(1.0f - beta2, g), g));
  m1 = a;
  m2 = b;
  const float num = __fmul_rn(lr, __fdiv_rn(a, c1));
  const float den = __fadd_rn(__fsqrt_rn(__fdiv_rn(b, c2)), eps);
  return __fsub_rn(p, __fdiv_rn(num, den));
}

