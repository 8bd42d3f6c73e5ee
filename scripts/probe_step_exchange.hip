// Measurement probe (not product code): what would ONE time step of a persistent recurrent kernel pay for exchanging
// h_t between workgroups inside a launch, compared with the kernel boundary the product uses (rnn.hip: one launch per step)?
//
// Geometry = the C3 recurrence: 256 workgroups (2 directions x 2 batch tiles x 64 hidden slices), one per CU (forced by
// 100 KB of dynamic LDS, the size of a resident bf16 W_hh slice).  Per step every workgroup
//   produces its 1 KB piece of h_t (32 rows x 16 units bf16)  -> 16-byte sc1 (agent-scope, write-through) stores,
//   publishes a per-producer flag (step number)               -> drained sc1 store,
//   waits for the 64 flags of its (direction, batch tile) group -> one wave polls 64 flags, bounded spin,
//   gathers the group's 64 KB of h_t                            -> 16-byte sc1 loads, 8 in flight per lane,
// and checks every 16-byte chunk carries the expected step tag (stale cross-XCD L2 lines would show up here).
// Prints us/step for (a) flags only, (b) flags + payload.   Build: hipcc --offload-arch=gfx950 -O3 (scripts/build_probes.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NSL = 64;            // producers per group
constexpr int NGROUPS = 4;         // 2 directions x 2 batch tiles
constexpr int PIECE = 1024;        // bytes per producer per step
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// buf: [2 parity][NGROUPS][NSL][PIECE] ; flags: [NGROUPS][NSL] (u32, monotonically increasing step + 1)
__global__ __launch_bounds__(512) void exchange_kernel(char* buf, unsigned* flags, int steps, int with_payload, unsigned* err, unsigned* abort_flag,
                                                       unsigned long long* sink) {
  extern __shared__ char lds[];
  const int wg = blockIdx.x;
  const int group = wg / NSL, slice = wg % NSL;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
  unsigned long long acc = 0;
  unsigned bad = 0;
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    char* pb = buf + (((size_t)(s & 1) * NGROUPS + group) * NSL) * PIECE;
    if (with_payload == 2) {
      // tag-polling variant: no flags, no store drain — the producer fires its tagged 1 KB piece and every consumer wave re-reads its
      // 8 pieces until each 16-byte chunk carries this step's tag (the payload IS the flag)
      if (wave == 0) {
        u32x4 v = {(unsigned)s, (unsigned)slice, (unsigned)lane, 0x5eed0000u + (unsigned)group};
        store16_sc1(pb + slice * PIECE + lane * 16, v);
      }
      u32x4 v[8];
      int spins = 0;
      while (true) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = load16_sc1(pb + (wave * 8 + i) * PIECE + lane * 16);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok = ok && (v[i].x == (unsigned)s);
        if (__ballot(ok) == ~0ull) break;
        if (++spins > (1 << 16)) { if (lane == 0) s_abort = 1; break; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bad += (v[i].y != (unsigned)(wave * 8 + i)) || (v[i].z != (unsigned)lane);
        acc += v[i].w;
      }
      *reinterpret_cast<u32x4*>(lds + (wave * 8) * PIECE + lane * 16) = v[0];
      __syncthreads();          // (the consumer's MFMA phase would start here; also keeps a fast wave from lapping the buffer parity)
      if (s_abort) break;
      continue;
    }
    if (wave == 0) {
      // produce: one wave-store of 1 KB, tagged with the step
      u32x4 v = {(unsigned)s, (unsigned)slice, (unsigned)lane, 0x5eed0000u + (unsigned)group};
      store16_sc1(pb + slice * PIECE + lane * 16, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&flags[group * NSL + slice], (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // wait for the whole group
      int spins = 0;
      while (true) {
        const unsigned f = __hip_atomic_load(&flags[group * NSL + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(f >= (unsigned)(s + 1)) == ~0ull) break;
        if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          if (lane == 0) { s_abort = 1; __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_abort) break;
    if (with_payload) {
      // gather 64 KB: wave w reads pieces w*8 .. w*8+7 (1 KB each, one wave-load per piece), 8 loads in flight per lane
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = load16_sc1(pb + (wave * 8 + i) * PIECE + lane * 16);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bad += (v[i].x != (unsigned)s) || (v[i].y != (unsigned)(wave * 8 + i)) || (v[i].z != (unsigned)lane);
        acc += v[i].w;
      }
      // stand-in for "h_t is now in LDS for the MFMA phase"
      *reinterpret_cast<u32x4*>(lds + (wave * 8) * PIECE + lane * 16) = v[0];
    }
    __syncthreads();
  }
  if (bad) atomicAdd(err, bad);
  if (acc == 0x123456789abcull) sink[0] = acc;
}

// the product's alternative: one (empty-bodied) launch per step with the same grid
__global__ __launch_bounds__(512) void boundary_kernel(char* buf, int s) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0 && buf == nullptr) lds[0] = (char)s;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2000;
  const int nwg = NGROUPS * NSL;
  char* buf;
  unsigned *flags, *err, *abort_flag;
  unsigned long long* sink;
  CHECK(hipMalloc(&buf, (size_t)2 * NGROUPS * NSL * PIECE));
  CHECK(hipMalloc(&flags, NGROUPS * NSL * sizeof(unsigned)));
  CHECK(hipMalloc(&err, 4));
  CHECK(hipMalloc(&abort_flag, 4));
  CHECK(hipMalloc(&sink, 8));
  const int lds_bytes = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)exchange_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CHECK(hipFuncSetAttribute((const void*)boundary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int with_payload = 0; with_payload < 3; ++with_payload) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(flags, 0, NGROUPS * NSL * sizeof(unsigned)));
      CHECK(hipMemset(err, 0, 4));
      CHECK(hipMemset(abort_flag, 0, 4));
      CHECK(hipMemset(buf, 0xee, (size_t)2 * NGROUPS * NSL * PIECE));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(exchange_kernel, dim3(nwg), dim3(512), lds_bytes, 0, buf, flags, steps, with_payload, err, abort_flag, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned h_err, h_abort;
      CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&h_abort, abort_flag, 4, hipMemcpyDeviceToHost));
      printf("in-launch exchange  payload=%d rep=%d: %.3f us/step  (stale/wrong chunks %u, aborted %u)\n", with_payload, rep, ms * 1e3 / steps, h_err,
             h_abort);
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(boundary_kernel, dim3(nwg), dim3(512), lds_bytes, 0, buf, s);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel boundary (empty 256-WG launch per step) rep=%d: %.3f us/step\n", rep, ms * 1e3 / steps);
  }
  return 0;
}
