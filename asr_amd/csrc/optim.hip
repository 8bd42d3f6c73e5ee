// Fused flat-buffer AdamW (decoupled weight decay), one launch for all parameters.
// Same update as torch.optim.AdamW single-tensor path (trainers/__main__.py:41-47; hyper-parameters
// asr_deepspeech/config.yml:41-47):  p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM-bound: 16 B read + 12 B write per parameter... (p,g,m,v in; p,m,v out) = 28 B/param.
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float gscale, const int* __restrict__ apply) {
  if (apply && *apply <= 0) return;                     // device-side gate (1 = apply; 0 / -1: the step was found invalid after this launch was enqueued)
  const long long n4 = n / 4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    pp *= (1.f - lr * wd);
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    f32x4 den;
    den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
    den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
    pp -= (lr / bc1) * mm / den;
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail
  const long long tail0 = n4 * 4;
  for (long long i = tail0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pp = p[i], gg = g[i] * gscale, mm = m[i], vv = v[i];
    pp *= (1.f - lr * wd);
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n, float s) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= s;
}

// fp32 mode's split products for the conv stack (engine.F32_CONV): r = x - float(bf16(x)), the part of x a bf16 operand drops; feeding r
// through the bf16 mode's own cast / pack kernels yields the "lo" operand of the three-term product
__global__ __launch_bounds__(256) void bf16_residual_kernel(const float* __restrict__ x, float* __restrict__ r, long long n4, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = v.x - (float)(__bf16)v.x; o.y = v.y - (float)(__bf16)v.y; o.z = v.z - (float)(__bf16)v.z; o.w = v.w - (float)(__bf16)v.w;
    reinterpret_cast<float4*>(r)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const long long i = n4 * 4 + threadIdx.x;
    r[i] = x[i] - (float)(__bf16)x[i];
  }
}

// out = a + b + c (out may be a; c may be NULL: out = a + b): the partial results of a split product
__global__ __launch_bounds__(256) void sum3_kernel(const float* a, const float* __restrict__ b, const float* __restrict__ c, float* out,
                                                   long long n4, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
    const float4 w = c ? reinterpret_cast<const float4*>(c)[i] : float4{0.f, 0.f, 0.f, 0.f};
    float4 o;
    o.x = u.x + v.x + w.x; o.y = u.y + v.y + w.y; o.z = u.z + v.z + w.z; o.w = u.w + v.w + w.w;
    reinterpret_cast<float4*>(out)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const long long i = n4 * 4 + threadIdx.x;
    out[i] = a[i] + b[i] + (c ? c[i] : 0.f);
  }
}

}  // namespace

// step is 1-based.  grad_scale multiplies g on the fly (e.g. 1/world_size after an all-reduce SUM).
// apply_flag: NULL, or a device int the kernel reads when it RUNS: <= 0 = leave everything untouched.  It lets the host enqueue the update
// before it knows whether the step is valid (finite loss on every rank, no starved recurrence launch: ds2_rnn_step_gate), i.e. without
// a host synchronisation between backward and the optimizer.
extern "C" int ds2_adamw_gated_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step, float grad_scale, const int* apply_flag, void* stream) {
  DS2_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "ds2_adamw_f32: bad args");
  DS2_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
              "ds2_adamw_f32: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, (float)sqrt(bc2), grad_scale, apply_flag);
  DS2_LAUNCH_CHECK("adamw_kernel");
  return 0;
}

extern "C" int ds2_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, float grad_scale, void* stream) {
  return ds2_adamw_gated_f32(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}

namespace {
__global__ void add_i64_kernel(long long* __restrict__ x, int n, long long v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += v;
}
}  // namespace

// x[0..n) += v for a small int64 vector: every BatchNorm's num_batches_tracked in one launch (they are views of one buffer: asr_amd/params.py)
extern "C" int ds2_add_i64(long long* x, int n, long long v, void* stream) {
  DS2_REQUIRE(x && n >= 0, "ds2_add_i64: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_i64_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, n, v);
  DS2_LAUNCH_CHECK("add_i64_kernel");
  return 0;
}

extern "C" int ds2_scale_f32(float* x, long long n, float s, void* stream) {
  DS2_REQUIRE(x && n >= 0, "ds2_scale_f32: bad args");
  if (n == 0) return 0;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, s);
  DS2_LAUNCH_CHECK("scale_kernel");
  return 0;
}

extern "C" int ds2_bf16_residual_f32(const float* x, float* r, long long n, void* stream) {
  DS2_REQUIRE(x && r && n >= 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)r % 16) == 0, "ds2_bf16_residual_f32: bad args");
  if (n == 0) return 0;
  long long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipLaunchKernelGGL(bf16_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, r, n / 4, n);
  DS2_LAUNCH_CHECK("bf16_residual_kernel");
  return 0;
}

extern "C" int ds2_sum3_f32(const float* a, const float* b, const float* c, float* out, long long n, void* stream) {
  DS2_REQUIRE(a && b && out && n >= 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) % 16) == 0, "ds2_sum3_f32: bad args");
  if (n == 0) return 0;
  long long blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipLaunchKernelGGL(sum3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n / 4, n);
  DS2_LAUNCH_CHECK("sum3_kernel");
  return 0;
}
