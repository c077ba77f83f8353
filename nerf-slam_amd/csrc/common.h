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
// A/B switches and comparison kernels.  The sources build TWO libraries (Makefile):
//   libnerfslam_hip.so            the product: ns_variant_env() is the constant nullptr, every switch folds to its default and
//                                 the superseded / comparison kernels (everything inside `#ifdef NS_TEST_VARIANTS`) are not in it;
//   libnerfslam_hip_variants.so   -DNS_TEST_VARIANTS: the same entry points plus the comparison kernels, selected through NS_*
//                                 environment variables -- honoured only together with the master switch NS_VARIANTS.  The tests
//                                 and tools that compare kernels load THIS library (nerfslam/_lib.py: lib() while NS_VARIANTS is
//                                 set); bench.py and the product never do.
// ---------------------------------------------------------------------------------------------
#include <stdlib.h>
#ifdef NS_TEST_VARIANTS
static inline const char* ns_variant_env(const char* name) {
  return getenv("NS_VARIANTS") != nullptr ? getenv(name) : nullptr;     // (not cached: tests switch it on and off in one process)
}
#else
static inline constexpr const char* ns_variant_env(const char*) { return nullptr; }
#endif

// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which the compiler
// implements as s_waitcnt vmcnt(0) lgkmcnt(0): every global load AND every global store of the wave has to be acknowledged
// before the barrier -- a full memory round trip (~2-3 us on a busy chip) wherever a kernel has loads in flight for later use
// or has just stored results nobody in the workgroup reads.  Use this where the two sides of the barrier communicate through
// LDS alone; global loads issued before it stay in flight behind it (the compiler still waits for each where it is used).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// wave64 reductions on the VALU: six DPP-modified adds (quad swaps, half-row / row mirrors, then
// the row broadcasts), no LDS crossbar traffic and no waitcnt per step -- `__shfl_xor` lowers to
// ds_bpermute_b32 on gfx950, whose ~100-cycle latency per step made the 27- and 42-value epilogues
// of the BA kernels cost more than their pixel loops.  The total ends in lane 63; wave_sum()
// broadcasts it with one readlane so every lane (and any later scalar use) sees the same value.
// The order of the additions is fixed, so results are bit-reproducible.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror      -> every lane holds its 16-lane row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63 holds the wave total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// integer form of wave_sum (same six DPP steps, v_add_u32): exact and order independent
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}

__device__ __forceinline__ int wave_sum_i(int v) {
  v = dpp_add_i<0xB1, 0xf>(v);
  v = dpp_add_i<0x4E, 0xf>(v);
  v = dpp_add_i<0x141, 0xf>(v);
  v = dpp_add_i<0x140, 0xf>(v);
  v = dpp_add_i<0x142, 0xa>(v);
  v = dpp_add_i<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
