// LDS access forms of the persistent recurrence's reduction (rnn.hip: red[2][NW][6][64] f32x4): cycles per instruction with 8 waves of one workgroup
// issuing together, and what the bank-conflict counter says (run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE for the latter).
//   0  ds_read2st64_b32, the two dwords of a lane 256 B x k apart (same bank)         = what the compiler makes of the 24 scalar reads today
//   1  two ds_read_b32 at the same addresses
//   2  ds_read2_b32 with the second dword one row of 1040 B (+ 4 banks) further        = a padded row pitch
//   3  ds_write_b128, lane-linear 16 B (the partial-sum write)
//   4  ds_read_b128, lane-linear 16 B
//   5  ds_write_b16 lane-linear, 6 ds_write_b64 lane-linear 8 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  // the gate-math read address of rnn_fwd_persistent_kernel: (wave & 3) * 64 + jl * 4 + (lane >> 4) floats
  const unsigned rd = (unsigned)(((wave & 3) * 64 + (lane & 15) * 4 + (lane >> 4)) * 4);
  const unsigned lin16 = (unsigned)(wave * 6144 + lane * 16);
  float acc = 0.f;
  f32x4 v4 = {1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if constexpr (MODE == 0) {
        f32x2 r;
        asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(rd + u * 1024) : "memory");
        acc += r.x + r.y;
      } else if constexpr (MODE == 1) {
        float a, b;
        asm volatile("ds_read_b32 %0, %2 offset:1024\n\tds_read_b32 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(rd + u * 1024) : "memory");
        acc += a + b;
      } else if constexpr (MODE == 2) {
        f32x2 r;
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(rd + u * 1040) : "memory");
        acc += r.x + r.y;
      } else if constexpr (MODE == 3) {
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lin16 + (u % 6) * 1024), "v"(v4) : "memory");
      } else if constexpr (MODE == 4) {
        f32x4 r;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lin16 + (u % 6) * 1024) : "memory");
        acc += r.x + r.w;
      } else if constexpr (MODE == 5) {
        f32x2 v2 = {1.f, 2.f};
        asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"((unsigned)(wave * 6144 + lane * 8 + (u % 6) * 1024)), "v"(v2) : "memory");
      } else if constexpr (MODE == 6) {
        // rnn_bwd_ksplit_kernel's staging write: As[g][row = 2 wave + hrow][unit], 80-byte rows, bf16
        const int q4 = lane & 3, hrow = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
        const unsigned ad = (unsigned)((2 * wave + hrow) * 80 + (b5 * 16 + 4 * q4 + 2 * b4 + b3) * 2 + (u % 4) * 1280);
        asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(ad), "v"(1.0f) : "memory");
      } else if constexpr (MODE == 7) {
        // its operand read: row lane & 15 (80-byte pitch), 16-byte chunk lane >> 4
        f32x4 r;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)((lane & 15) * 80 + (lane >> 4) * 16 + (u % 4) * 1280)) : "memory");
        acc += r.x + r.w;
      } else {
        // the same write with the two bf16 of a dword in ONE lane pair's ... (candidate): lanes differing in b3 write different dwords
        const int q4 = lane & 3, hrow = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
        const unsigned ad = (unsigned)((2 * wave + hrow) * 80 + (b5 * 16 + 4 * q4 + 2 * b3 + b4) * 2 + (u % 4) * 1280);
        asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(ad), "v"(1.0f) : "memory");
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}
// same, but the 12 instructions issued back to back with ONE wait (throughput form: what the kernel does)
template <int MODE>
__global__ __launch_bounds__(512) void probe_tp(long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  const unsigned rd = (unsigned)(((wave & 3) * 64 + (lane & 15) * 4 + (lane >> 4)) * 4);
  float acc = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x2 r[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if constexpr (MODE == 0) asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:8" : "=v"(r[u]) : "v"(rd + u * 1024) : "memory");
      else if constexpr (MODE == 1) asm volatile("ds_read_b32 %0, %2 offset:1024\n\tds_read_b32 %1, %2 offset:2048" : "=&v"(r[u].x), "=&v"(r[u].y) : "v"(rd + u * 1024) : "memory");
      else asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:4" : "=v"(r[u]) : "v"(rd + u * 1040) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 12; ++u) { asm volatile("" : "+v"(r[u])); acc += r[u].x + r[u].y; }
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
  long long* out; float* sink;
  CK(hipMalloc(&out, 8 * 256 * sizeof(long long))); CK(hipMalloc(&sink, 64));
  const int iters = 2000;
  long long h[8];
  const char* names[9] = {"ds_read2st64_b32 (same bank pair)", "2 x ds_read_b32 (same addresses)", "ds_read2_b32, rows 1040 B apart", "ds_write_b128 lane-linear",
                          "ds_read_b128 lane-linear", "ds_write_b64 lane-linear", "ds_write_b16 (K-split staging)", "ds_read_b128 80 B rows (K-split)",
                          "ds_write_b16, b3 / b4 swapped"};
#define RUN(K, M, label)                                                                                        \
  do {                                                                                                          \
    hipLaunchKernelGGL((K<M>), dim3(256), dim3(512), 65536, 0, out, sink, iters);                               \
    CK(hipDeviceSynchronize());                                                                                 \
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));                                                    \
    long long mx = 0; for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;                                   \
    printf("%-10s %-36s %8.1f clocks per 12-instruction group and wave set (8 waves together)\n", label, names[M], (double)mx / iters);   \
  } while (0)
  RUN(probe, 0, "latency"); RUN(probe, 1, "latency"); RUN(probe, 2, "latency"); RUN(probe, 3, "latency"); RUN(probe, 4, "latency"); RUN(probe, 5, "latency"); RUN(probe, 6, "latency"); RUN(probe, 7, "latency"); RUN(probe, 8, "latency");
  RUN(probe_tp, 0, "throughput"); RUN(probe_tp, 1, "throughput"); RUN(probe_tp, 2, "throughput");
  return 0;
}
