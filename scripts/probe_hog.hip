// A stand-in for RCCL's channel kernels: N workgroups of 512 threads that stream HBM — 16-byte loads from one
// buffer, 16-byte stores to another, a private slice per workgroup — until a wall-clock deadline (s_memrealtime, 100 MHz).  RCCL's ring /
// tree kernels are exactly this shape on a GPU: one resident workgroup per channel that copies and reduces chunks for the length of the
// collective, holding its CU's registers / LDS and a share of the memory system.  scripts/r6_dp_window.py launches it where the
// data-parallel schedule issues the big all-reduce.     hipcc --offload-arch=gfx950 -O3 -shared -fPIC probe_hog.hip -o libhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void hog_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long vec_per_wg, long long ticks, unsigned long long* moved) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const f32x4* s = src + (long long)blockIdx.x * vec_per_wg;
  f32x4* d = dst + (long long)blockIdx.x * vec_per_wg;
  unsigned long long n = 0;
  for (;;) {
    for (long long i = threadIdx.x; i < vec_per_wg; i += 512 * 4) {        // four independent 16-byte streams per thread in flight
      f32x4 a = __builtin_nontemporal_load(s + i);
      f32x4 b = i + 512 < vec_per_wg ? __builtin_nontemporal_load(s + i + 512) : a;
      f32x4 c = i + 1024 < vec_per_wg ? __builtin_nontemporal_load(s + i + 1024) : a;
      f32x4 e = i + 1536 < vec_per_wg ? __builtin_nontemporal_load(s + i + 1536) : a;
      __builtin_nontemporal_store(a + b, d + i);                            // (a sum: the reduce of a reduce-scatter step)
      if (i + 512 < vec_per_wg) __builtin_nontemporal_store(b + c, d + i + 512);
      if (i + 1024 < vec_per_wg) __builtin_nontemporal_store(c + e, d + i + 1024);
      if (i + 1536 < vec_per_wg) __builtin_nontemporal_store(e + a, d + i + 1536);
      n += 4;
      if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) >= ticks) goto done;
    }
  }
done:
  if (threadIdx.x == 0) atomicAdd(moved, n * 512ull * 32ull);                // bytes read + written by this workgroup (approximately)
}

// launches `wgs` workgroups for `usec` microseconds on `stream`; src / dst: device buffers of at least wgs * bytes_per_wg bytes
extern "C" int hog_launch(int wgs, double usec, const void* src, void* dst, long long bytes_per_wg, unsigned long long* moved, void* stream) {
  hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(512), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes_per_wg / 16, (long long)(usec * 100.0), moved);
  return (int)hipGetLastError();
}
