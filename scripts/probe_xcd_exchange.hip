// Measurement probe (not product code): tagged-payload exchange when all workgroups of a group sit on ONE XCD (workgroup id % 8, the observed
// round-robin placement) and exchange through that XCD's L2: plain 16-byte stores, loads that bypass only the per-CU L1 (sc0), versus the
// system-coherent (sc1) stores and loads the product uses.  8 groups x 32 workgroups, 1 KB piece per workgroup and step, every workgroup
// gathers its group's 32 KB.  Prints us/step, the number of stale/wrong chunks, and whether every group really was on one XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NG = 8, NSL = 32, PIECE = 1024;
template <int MODE> __device__ __forceinline__ void st16(void* p, u32x4 v) {
  if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ u32x4 ld16(const void* p) {
  u32x4 v;
  if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE, int SPREAD>
__global__ __launch_bounds__(512) void xchg(char* buf, int steps, unsigned* err, unsigned* xcc) {
  extern __shared__ char lds[];
  const int wg = blockIdx.x, group = SPREAD ? wg / NSL : wg % NG, slice = SPREAD ? wg % NSL : wg / NG;   // SPREAD: a group's members on all 8 XCDs
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[wg] = id & 0xf; }
  unsigned bad = 0, acc = 0;
  for (int s = 0; s < steps; ++s) {
    char* pb = buf + (((size_t)(s & 3) * NG + group) * NSL) * PIECE;
    if (wave == 0) st16<MODE>(pb + slice * PIECE + lane * 16, u32x4{(unsigned)s, (unsigned)slice, (unsigned)lane, 7u});
    u32x4 v[4];
    int spins = 0;
    while (true) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ld16<MODE>(pb + (wave * 4 + i) * PIECE + lane * 16);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
      bool ok = true;
#pragma unroll
      for (int i = 0; i < 4; ++i) ok = ok && v[i].x == (unsigned)s;
      if (__ballot(ok) == ~0ull) break;
      if (++spins > (1 << 18)) { bad += 1000000; break; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { bad += (v[i].y != (unsigned)(wave * 4 + i)) || (v[i].z != (unsigned)lane); acc += v[i].w; }
    *reinterpret_cast<u32x4*>(lds + wave * 1024 + lane * 16) = v[0];
    __syncthreads();
    if (bad >= 1000000) break;
  }
  if (bad) atomicAdd(err, bad);
  if (acc == 12345u) err[1] = acc;
}
template <int MODE, int SPREAD> void run(const char* name, char* buf, unsigned* err, unsigned* xcc, int steps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)xchg<MODE, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(buf, 0xee, (size_t)4 * NG * NSL * PIECE));
    CHECK(hipMemset(err, 0, 8));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((xchg<MODE, SPREAD>), dim3(NG * NSL), dim3(512), 100 * 1024, 0, buf, steps, err, xcc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr[2], hx[NG * NSL];
    CHECK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    int split = 0;
    for (int g = 0; g < NG; ++g) for (int s2 = 1; s2 < NSL; ++s2) split += SPREAD ? (hx[g * NSL + s2] != hx[g * NSL]) : (hx[s2 * NG + g] != hx[g]);
    printf("%s rep %d: %.3f us/step, stale/wrong or starved %u, workgroups not on their group's XCD %d\n", name, rep, ms * 1e3 / steps, herr[0], split);
  }
}
int main() {
  char* buf; unsigned *err, *xcc;
  CHECK(hipMalloc(&buf, (size_t)4 * NG * NSL * PIECE)); CHECK(hipMalloc(&err, 8)); CHECK(hipMalloc(&xcc, NG * NSL * 4));
  run<0, 0>("sc1 store + sc1 load, group on ONE XCD      ", buf, err, xcc, 2000);
  run<0, 1>("sc1 store + sc1 load, group over all 8 XCDs ", buf, err, xcc, 2000);
  return 0;
}
