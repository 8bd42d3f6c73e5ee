// Round 5 probe: what the matrix pipe sustains on RANDOM bf16 operands (the chip is power-limited there), by MFMA shape / waves per SIMD /
// accumulator file, with and without an LDS fragment-read stream beside it.  No result is checked; accumulators are written out so that
// nothing is optimised away.   build: hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_mfma_power scripts/probe_mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V = 0: 32x32x16, 8 accumulator tiles (128 x 64 per wave);  V = 1: 16x16x32, 32 accumulator tiles;  V = 2: 32x32x16, 16 tiles (128 x 128)
template <int V, int THREADS, bool LDSRD>
__global__ __launch_bounds__(THREADS) void k(const f32x4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  constexpr int NA = V == 1 ? 8 : 4, NB = V == 1 ? 4 : (V == 2 ? 4 : 2);
  f32x4 fa[NA], fb[NB];
  for (int i = 0; i < NA; ++i) fa[i] = src[(blockIdx.x * 7 + i) * 64 + lane];
  for (int i = 0; i < NB; ++i) fb[i] = src[(blockIdx.x * 5 + 16 + i) * 64 + lane];
  if (LDSRD) {
    for (int i = threadIdx.x; i < 32768 / 16; i += THREADS) reinterpret_cast<f32x4*>(lds)[i] = src[i + lane];
    __syncthreads();
  }
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + ((threadIdx.x >> 6) * 64 + lane) * 16 % 16384;
  if constexpr (V == 1) {
    f32x4 acc[NA][NB];
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            if (LDSRD && ((i * NB + j) % 3) == 0) {
              f32x4 t;
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(la), "n"(((i * NB + j) / 3) * 1024));
              asm volatile("" ::"v"(t));
            }
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
  } else {
    f32x16 acc[NA][NB];
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < (V == 2 ? 2 : 4); ++kk) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            if (LDSRD && (i * NB + j) < (V == 2 ? 8 : 6)) {
              f32x4 t;
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(la), "n"((i * NB + j) * 1024));
              asm volatile("" ::"v"(t));
            }
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
  }
}

template <int V, int THREADS, bool LDSRD>
void run(const char* name, const f32x4* src, float* out, int iters, double flop_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * (THREADS == 512 ? 1 : 1);
  hipFuncSetAttribute((const void*)k<V, THREADS, LDSRD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, THREADS, LDSRD>), dim3(grid), dim3(THREADS), LDSRD ? 65536 : 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = flop_per_iter_per_wave * iters * (THREADS / 64) * grid;
  printf("%-58s %8.1f us  %7.0f TF/s\n", name, ms * 1e3, fl / ms / 1e9);
}

int main(int argc, char** argv) {
  const int zero = argc > 1 ? atoi(argv[1]) : 0;
  const size_t n = 1 << 20;
  std::vector<unsigned short> h(n * 8);
  srand(1);
  for (auto& x : h) {                                    // random bf16 in [-2, 2) with random mantissas (or zeros)
    unsigned short v = (unsigned short)(((rand() & 1) << 15) | ((0x3e + (rand() % 3)) << 8 >> 1 << 1) | (rand() & 0x1ff));
    x = zero ? 0 : v;
  }
  f32x4* src; float* out;
  hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
  const int iters = 4000;
  const double f32 = 2.0 * 32 * 32 * 16, f16 = 2.0 * 16 * 16 * 32;
  printf("operands: %s\n", zero ? "zeros" : "random");
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 512, false>("32x32x16, 8 waves/CU, 8 acc tiles                        ", src, out, iters, f32 * 32);
    run<1, 512, false>("16x16x32, 8 waves/CU, 32 acc tiles                       ", src, out, iters, f16 * 64);
    run<0, 256, false>("32x32x16, 4 waves/CU, 8 acc tiles                        ", src, out, iters, f32 * 32);
    run<2, 256, false>("32x32x16, 4 waves/CU, 16 acc tiles (128 x 128 per wave)  ", src, out, iters, f32 * 32);
    run<1, 256, false>("16x16x32, 4 waves/CU, 32 acc tiles                       ", src, out, iters, f16 * 64);
    run<0, 512, true>("32x32x16, 8 waves/CU + 6 ds_read_b128 per 8 MFMA         ", src, out, iters, f32 * 32);
    run<1, 512, true>("16x16x32, 8 waves/CU + 11 ds_read_b128 per 32 MFMA       ", src, out, iters, f16 * 64);
    run<2, 256, true>("32x32x16, 4 waves/CU, 16 tiles + 8 ds_read_b128 per 16   ", src, out, iters, f32 * 32);
  }
  return 0;
}
