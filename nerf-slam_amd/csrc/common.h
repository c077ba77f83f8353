// common.h -- shared host/device helpers for libnerfslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nerfslam_hip.h"

#define NS_WAVE 64

void ns_set_error(const char* fmt, ...);

#define NS_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      ns_set_error(__VA_ARGS__);     \
      return NS_EINVAL;              \
    }                                \
  } while (0)

// Call after every launch: does not synchronise, only picks up launch-configuration errors.
#define NS_CHECK_LAUNCH(name)                                                       \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      ns_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));           \
      return NS_ELAUNCH;                                                            \
    }                                                                               \
  } while (0)

static inline int ns_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// wave64 / block reductions.  DPP row shifts inside 16-lane rows, then row broadcasts: six
// full-rate VALU adds, no LDS traffic.  Result is valid in lane 63.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
  // __shfl_xor lowers to DPP / ds_swizzle / permlane on gfx950; the butterfly leaves the total in
  // every lane, which the callers rely on (any lane may publish it).
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
