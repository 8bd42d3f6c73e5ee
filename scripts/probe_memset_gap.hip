// What does a hipMemsetAsync between two kernels cost on the stream's timeline, against a fill KERNEL in the same place?
// (the persistent recurrence launchers reset their exchange buffer — ~2 MB of 0xff — in front of every launch: 10 per train step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void work(float* p, int n, int reps) {        // ~20 us of streaming work
  for (int r = 0; r < reps; ++r)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void fill16(u32x4* p, long long n16, unsigned v) {
  const u32x4 x = {v, v, v, v};
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = x;
}
int main() {
  const int n = 8 << 20; float* buf; char* xb;
  const size_t xbytes = 2 * 1024 * 1024 + 64;
  CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMalloc(&xb, xbytes + 64)); CK(hipMemset(buf, 0, n * sizeof(float)));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(work, dim3(2048), dim3(256), 0, s, buf, n, 1);
        if (mode == 1) CK(hipMemsetAsync(xb, 0xff, xbytes, s));
        if (mode == 2) hipLaunchKernelGGL(fill16, dim3(256), dim3(256), 0, s, (u32x4*)xb, (long long)(xbytes / 16), 0xffffffffu);
        if (mode == 3) CK(hipMemsetAsync(xb, 0, 2 * 1024 * 1024, s));
        hipLaunchKernelGGL(work, dim3(2048), dim3(256), 0, s, buf, n, 1);
      }
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) printf("%-44s %8.2f us per (kernel, X, kernel) triple\n", mode == 0 ? "nothing between" : mode == 1 ? "hipMemsetAsync 0xff, 2 MB + 64 B" : mode == 2 ? "fill kernel, same bytes" : "hipMemsetAsync 0, 2 MB", ms * 1000 / iters);
    }
  }
  return 0;
}
