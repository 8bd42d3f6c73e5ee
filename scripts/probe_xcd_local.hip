// Measurement probe (not product code): the persistent recurrence's per-step exchange (rnn.hip: tagged-payload polling, four round-robin
// buffers, sentinel reset two steps ahead, one workgroup barrier per step) when a group's 32 workgroups sit on ONE XCD and the exchange is
// served by THAT XCD's L2: plain (L2-retained) stores + L1-bypassing (sc1) loads, against the product's sc1 (write-through, L2-dropping)
// stores.  Roles are taken either from the workgroup id (id % 8 = the observed round-robin placement) or from a census of the REAL XCD ids
// (s_getreg HW_REG_XCC_ID + one atomic per workgroup): the second form does not assume anything about placement.
//   geometry F: forward  — every workgroup publishes 1 KB per step, every wave gathers 4 chunks  (32 KB per workgroup)
//   geometry B: backward — every workgroup publishes 3 KB per step, every wave gathers 12 chunks (96 KB per workgroup)
// Prints us/step, stale/wrong chunks, starved waves, and how many workgroups were not on their group's XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NG = 8, NSL = 32;
constexpr unsigned SENT = 0xffffffffu;

// 16-byte exchange accesses as compiler-visible buffer operations (aux bit 4 = sc1): the compiler tracks the asynchronous loads itself
// (an inline-asm load's destination is unprotected until one's own wait: copies of the still-empty registers were observed)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
template <int SM> __device__ __forceinline__ void st16(rsrc_t r, unsigned off, u32x4 v) {
  if (SM == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
  else if (SM == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 1);
}
template <int LM> __device__ __forceinline__ u32x4 ld16(rsrc_t r, unsigned off) {
  return LM == 0 ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17);
}

// ROLE 0: group = wg % 8 ; 1: census of real XCC ids ; 2: group = wg / 32 (spread over all XCDs)
template <int SM, int LM, int ROLE, int PIECES, int POLL>
__global__ __launch_bounds__(512) void xchg(char* buf, int steps, unsigned* err, unsigned* xcc_out, unsigned* census) {
  constexpr int CH = 4 * PIECES;                     // chunks per wave
  constexpr int NCH = NSL * PIECES;                  // chunks per group buffer
  __shared__ int s_role[2];
  __shared__ u32x4 sink[8][64];
  const int wg = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned myxcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
  myxcc &= 0xf;
  int group, slice;
  if (ROLE == 1) {
    if (threadIdx.x == 0) {
      const int slot = (int)atomicAdd(&census[myxcc], 1u);
      atomicAdd(&census[8], 1u);
      int spins = 0;
      while (__hip_atomic_load(&census[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)gridDim.x && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
      bool even = true;
      for (int x = 0; x < 8; ++x) even = even && __hip_atomic_load(&census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)NSL;
      s_role[0] = even ? (int)myxcc : wg % NG;
      s_role[1] = even ? slot : wg / NG;
      if (!even && wg == 0) err[2] = 1;              // census uneven: fell back to id-based roles
    }
    __syncthreads();
    group = s_role[0]; slice = s_role[1];
  } else if (ROLE == 0) { group = wg % NG; slice = wg / NG; }
  else { group = wg / NSL; slice = wg % NSL; }
  if (threadIdx.x == 0) xcc_out[wg] = myxcc | (group << 8);
  const size_t bufbytes = (size_t)NG * NCH * 1024;
  const unsigned gbase = (unsigned)group * NCH * 1024;
  const rsrc_t rs = make_rsrc(buf, (unsigned)(4 * bufbytes));
  // publishing lanes: as in the product, a wave owns 4 rows x 16 units of the workgroup's 16 x 32 tile = 8 x 16 B per piece
  const bool pub_lane = lane < 8 * PIECES;
  const int pl = lane & 7, piece = lane >> 3;
  const int lane_idx = ((wave >> 2) * 2 + (pl >> 2)) * 16 + (wave & 3) * 4 + (pl & 3);
  const unsigned pub_off = gbase + ((unsigned)(piece * NSL + slice) * 64 + lane_idx) * 16;
  unsigned bad = 0, starved = 0;
  for (int s = 0; s < steps; ++s) {
    u32x4 v[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) v[k] = u32x4{0u, 0u, 0u, 0u};
    if (s > 0) {
      const unsigned xin = gbase + (unsigned)((s - 1) & 3) * (unsigned)bufbytes;
      unsigned pend = (1u << CH) - 1u;
      int spins = 0;
      // POLL 0: every iteration re-reads all pending chunks.  1: first wait for ONE probe chunk (the wave's last), then read the rest.
      // 2: first attempt reads everything; after a miss only one pending chunk is polled until it lands, then all pending again.
      unsigned mask = POLL == 1 ? (1u << (CH - 1)) : ~0u;
      while (pend) {
        asm volatile("" ::: "memory");                     // the poll must really re-read
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (mask & pend & (1u << k)) v[k] = ld16<LM>(rs, xin + ((unsigned)(wave + 8 * k) * 64 + lane) * 16);
        unsigned got = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (mask & pend & (1u << k)) {
            const bool ok = v[k].x != SENT && v[k].y != SENT && v[k].z != SENT && v[k].w != SENT;
            if (__ballot(ok) == ~0ull) got |= 1u << k;
          }
        got = __builtin_amdgcn_readfirstlane(got);
        pend &= ~got;
        if (POLL == 1) { if (got) mask = ~0u; }
        else if (POLL == 2) { if (mask == ~0u) { if (pend) mask = pend & (~pend + 1u); } else if (got) mask = ~0u; }
        if (pend && ++spins > (1 << 18)) { starved = 1; break; }
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) bad += (v[k].x != (unsigned)(s - 1)) || (v[k].y != (unsigned)(wave + 8 * k)) || (v[k].z != (unsigned)lane);
      sink[wave][lane] = v[0];
    }
    __syncthreads();
    if (starved) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pub_lane) {
      st16<SM>(rs, (unsigned)(s & 3) * (unsigned)bufbytes + pub_off, u32x4{(unsigned)s, (unsigned)(piece * NSL + slice), (unsigned)lane_idx, 7u});
      st16<SM>(rs, (unsigned)((s + 2) & 3) * (unsigned)bufbytes + pub_off, u32x4{SENT, SENT, SENT, SENT});
    }
  }
  if (bad) atomicAdd(&err[0], bad);
  if (starved && lane == 0) atomicAdd(&err[1], 1u);
}

template <int SM, int LM, int ROLE, int PIECES, int POLL = 0> void run(const char* name, char* buf, unsigned* err, unsigned* xcc, unsigned* census, int steps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t total = (size_t)4 * NG * NSL * PIECES * 1024;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(buf, 0xff, total));
    CHECK(hipMemset(err, 0, 16));
    CHECK(hipMemset(census, 0, 64));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((xchg<SM, LM, ROLE, PIECES, POLL>), dim3(NG * NSL), dim3(512), 0, 0, buf, steps, err, xcc, census);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr[4], hx[NG * NSL];
    CHECK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    int first[NG], split = 0;
    for (int g = 0; g < NG; ++g) first[g] = -1;
    for (int w = 0; w < NG * NSL; ++w) {
      const int g = hx[w] >> 8, x = hx[w] & 0xff;
      if (first[g] < 0) first[g] = x;
      else split += x != first[g];
    }
    printf("%s rep %d: %.3f us/step, stale/wrong %u, starved waves %u, census fallback %u, workgroups off their group's XCD %d\n", name, rep,
           ms * 1e3 / steps, herr[0], herr[1], herr[2], split);
  }
}
int main() {
  char* buf; unsigned *err, *xcc, *census;
  CHECK(hipMalloc(&buf, (size_t)4 * NG * NSL * 3 * 1024)); CHECK(hipMalloc(&err, 16)); CHECK(hipMalloc(&xcc, NG * NSL * 4)); CHECK(hipMalloc(&census, 64));
  const int steps = 2000;
  run<0, 0, 0, 1, 0>("F sc1 st, poll all (product)        ", buf, err, xcc, census, steps);
  run<1, 0, 1, 1, 0>("F plain st, census, poll all        ", buf, err, xcc, census, steps);
  run<1, 0, 1, 1, 1>("F plain st, census, probe-first     ", buf, err, xcc, census, steps);
  run<1, 0, 1, 1, 2>("F plain st, census, one-after-miss  ", buf, err, xcc, census, steps);
  run<0, 0, 0, 3, 0>("B sc1 st, poll all (product)        ", buf, err, xcc, census, steps);
  run<0, 0, 0, 3, 1>("B sc1 st, probe-first               ", buf, err, xcc, census, steps);
  run<0, 0, 0, 3, 2>("B sc1 st, one-after-miss            ", buf, err, xcc, census, steps);
  run<1, 0, 1, 3, 0>("B plain st, census, poll all        ", buf, err, xcc, census, steps);
  run<1, 0, 1, 3, 1>("B plain st, census, probe-first     ", buf, err, xcc, census, steps);
  run<1, 0, 1, 3, 2>("B plain st, census, one-after-miss  ", buf, err, xcc, census, steps);
  return 0;
}
