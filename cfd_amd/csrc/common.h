// Shared helpers for the gfx950 kernels of libcfd_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/cfd_amd.h"

namespace cfd {

// ---- error reporting ---------------------------------------------------------------
char *last_error_buf();  // thread-local, 512 bytes (api.hip)
int tuning(int key);     // cfd_set_tuning value, -1 = heuristic (api.hip)
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes), once per (kernel, CURRENT device): HIP keeps the attribute
// per device function, so a process that drives several GPUs needs it on each; thread-safe (api.hip).  Returns CFD_OK
// or CFD_EHIP with the error text set.
int set_max_lds(const void *fn, int bytes);

#define CFD_FAIL(code, ...)                                   \
  do {                                                        \
    snprintf(cfd::last_error_buf(), 512, __VA_ARGS__);        \
    return (code);                                            \
  } while (0)

#define CFD_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) CFD_FAIL(CFD_EINVAL, __VA_ARGS__);           \
  } while (0)

#define CFD_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) CFD_FAIL(CFD_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

#define CFD_LAUNCH_CHECK()                                                                        \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess) CFD_FAIL(CFD_EHIP, "kernel launch failed: %s", hipGetErrorString(e_));  \
  } while (0)

// ---- dtypes ------------------------------------------------------------------------
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

static inline size_t dtype_size(int dt) { return dt == CFD_BF16 ? 2 : 4; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // RNE, NaN-safe (v_cvt_pk_bf16_f32)

// activation in fp32
// NONE / RELU / LEAKY share one select form, v > 0 ? v : v * ns with ns = 1 / 0 / slope (a scalar, loop-invariant
// choice; NaN-preserving), so a run-time `act` costs one uniform branch (sigmoid) instead of a switch per element
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == CFD_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  const float ns = act == CFD_ACT_LEAKY ? slope : (act == CFD_ACT_RELU ? 0.f : 1.f);
  return v > 0.f ? v : v * ns;
}

// NONE / RELU / LEAKY as one branch-free form v > 0 ? v : ns * v with ns = 1 / 0 / slope: the GEMM epilogues run
// it on 128 accumulators per lane, where a per-element switch on a run-time `act` compiled to ~700 scalar branches per
// wave (a third of the 7.5 K tiles' worth of time an output tile's epilogue took).
__device__ __forceinline__ float act_neg_scale(int act, float slope) {
  return act == CFD_ACT_LEAKY ? slope : (act == CFD_ACT_RELU ? 0.f : 1.f);
}
__device__ __forceinline__ float apply_act_linear(float v, float ns) { return v > 0.f ? v : v * ns; }

// activation of a small register vector with the kind decided ONCE (one uniform branch per vector, none per element)
template <int N>
__device__ __forceinline__ void apply_act_vec(float (&o)[N], const float (&f)[N], float ns, bool sig) {
  if (sig) {
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = 1.f / (1.f + __expf(-f[i]));
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = apply_act_linear(f[i], ns);
  }
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace cfd
