// Common device/host helpers for libds2hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "ds2hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

int ds2_set_error(const char* fmt, ...);

#define DS2_REQUIRE(cond, ...)                                    \
  do {                                                            \
    if (!(cond)) return ds2_set_error(__VA_ARGS__);               \
  } while (0)

#define DS2_LAUNCH_CHECK(name)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) return ds2_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

#define DS2_HIP(call)                                                                \
  do {                                                                               \
    hipError_t e__ = (call);                                                         \
    if (e__ != hipSuccess) return ds2_set_error("%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Environment switches.  OPERATIONAL ones (DS2_RNN_PERSISTENT, DS2_RNN_SPIN_LIMIT, DS2_RNN_REARM_CALLS, DS2_RNN_XCD_LOCAL, DS2_F32_RNN) are read
// with getenv.  TUNING / A-B switches of experiments — most of which lost (DS2_GEMM_RING, DS2_GEMM_WAVES, DS2_GEMM_TILE, DS2_GEMM_DBG, ...) —
// are honoured only when DS2_EXPERIMENTAL=1 is set as well, so that a stray variable cannot change what the product runs.
#include <stdlib.h>
static inline const char* ds2_exp_getenv(const char* name) {
  const char* on = getenv("DS2_EXPERIMENTAL");
  return (on && on[0] == '1') ? getenv(name) : nullptr;
}
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// wave64 reductions (all 64 lanes participate)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Gate non-linearities on the hardware exp unit (v_exp_f32) and the hardware reciprocal (v_rcp_f32, 1 ulp; `__frcp_rn` expands to the
// 10-instruction IEEE division sequence — three of them per GRU gate set sat on the recurrence's critical path): ~2e-7 absolute error,
// far inside the 1e-3 parity bar.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float ax = fminf(fabsf(x), 15.0f);                 // tanh(15) == 1 in fp32; avoids exp overflow
  const float e = __expf(2.0f * ax);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  return copysignf(t, x);
}

// ---- cross-file internals (not part of the C ABI) ------------------------------------------------------------
// norm.hip: out0[c] = sum_k part[k][c][0], out1[c] = sum_k part[k][c][1] over `chunks` fp32 partial rows, combined in fp64
int ds2i_col_finalize_sums(const float* part, int chunks, int H, float* out0, float* out1, hipStream_t s);
// norm.hip: the elementwise half of BatchNorm1d backward, dX = gamma rstd (dY - s0/M - xhat s1/M), from already reduced column sums s0 / s1
int ds2i_bn1d_bwd_apply(const float* dY, int lddy, const float* X, int ldx, float* dX, int lddx, int M, int H, const float* mean, const float* var,
                        const float* gamma, const float* s0, const float* s1, float eps, hipStream_t s);
int ds2i_bn1d_bwd_apply_xbf(const float* dY, int lddy, const void* X_bf16, int ldx, float* dX, int lddx, int M, int H, const float* mean, const float* var,
                            const float* gamma, const float* s0, const float* s1, float eps, hipStream_t stream);
