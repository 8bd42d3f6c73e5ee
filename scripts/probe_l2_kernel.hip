// device side of scripts/probe_l2_residency.cpp: every workgroup streams its slice of a weight-like buffer (what a recurrent step does
// with W_hh); built as a stand-alone code object (--genco) so that the host can dispatch it through its own AQL queue
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(512) void read_w(const u32x4* __restrict__ w, int vec_per_wg, int nslices, unsigned* __restrict__ sink) {
  const u32x4* p = w + (size_t)(blockIdx.x % nslices) * vec_per_wg;     // two workgroups share a slice, like the two batch tiles
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int i = threadIdx.x; i < vec_per_wg; i += 512 * 4) {
    u32x4 a = p[i], b = p[min(i + 512, vec_per_wg - 1)], c = p[min(i + 1024, vec_per_wg - 1)], d = p[min(i + 1536, vec_per_wg - 1)];
    acc ^= a ^ b ^ c ^ d;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[blockIdx.x] = acc.x;   // never true for the test pattern: keeps the loads alive
}

// the same plus the step's all-to-all: every workgroup also reads 64 KB that the PREVIOUS launch wrote (1 KB per workgroup, 64
// producers per consumer group, ping-pong buffers) and writes its own 1 KB piece for the next launch
extern "C" __global__ __launch_bounds__(512) void read_w_exchange(const u32x4* __restrict__ w, int vec_per_wg, int nslices, unsigned* __restrict__ sink,
                                                                 u32x4* __restrict__ xbuf, int step, int fresh_vec) {
  const u32x4* p = w + (size_t)(blockIdx.x % nslices) * vec_per_wg;
  const int group = blockIdx.x / 64;                                   // 4 groups of 64 workgroups (direction x batch tile)
  const u32x4* xin = xbuf + ((size_t)((step + 1) & 1) * 4 + group) * 4096;   // 64 KB = 4096 x 16 B written one launch ago
  u32x4* xout = xbuf + ((size_t)(step & 1) * 4 + group) * 4096 + (blockIdx.x % 64) * 64;
  u32x4 acc = {0u, 0u, 0u, 0u};
  // mode (fresh_vec >> 20): 0 = every workgroup reads the fresh data first | 1 = only one workgroup per (XCD, group) reads it at all |
  // 2 = that one reads it first, the other seven of its XCD read W first and the fresh data afterwards (from the XCD's L2, if the
  // fetcher's lines have landed by then)
  const int mode_all = fresh_vec >> 20, nfresh = fresh_vec & 0xfffff;
  const int mode = mode_all & 7;
  const bool fetcher = (blockIdx.x % 64) < 8;                       // workgroup ids go round-robin over the 8 XCDs
  // all (up to 8) loads of a thread issued before the first use — a plain `for (...) acc ^= xin[i]` loop with a run-time trip count
  // is NOT unrolled and pays one fabric round trip per iteration (mode bit 3 keeps that serialised form for comparison)
  auto read_fresh = [&]() {
    if (mode_all & 8) { for (int i = threadIdx.x; i < nfresh; i += 512) acc ^= xin[i]; return; }
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int idx = threadIdx.x + j * 512; v[j] = xin[min(idx, nfresh - 1)]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (threadIdx.x + j * 512 < nfresh) acc ^= v[j];
  };
  auto read_w_all = [&]() {
    for (int i = threadIdx.x; i < vec_per_wg; i += 512 * 4) {
      u32x4 a = p[i], b = p[min(i + 512, vec_per_wg - 1)], c = p[min(i + 1024, vec_per_wg - 1)], d = p[min(i + 1536, vec_per_wg - 1)];
      acc ^= a ^ b ^ c ^ d;
    }
  };
  if (mode == 4) {
    // as mode 3 without the wait, the other seven eighths requested in rotated order (nearest neighbour's part last requested first)
    const int k = (blockIdx.x % 64) / 8, part = nfresh / 8;
    for (int i = threadIdx.x; i < part; i += 512) acc ^= xin[k * part + i];
    read_w_all();
    for (int j = 1; j < 8; ++j) {
      const int kk = (k + j) & 7;
      for (int i = threadIdx.x; i < part; i += 512) acc ^= xin[kk * part + i];
    }
  } else if (mode == 3) {
    // cooperative fetch: the 8 workgroups of one (XCD, group) each pull a different eighth of the fresh data through the fabric
    // (into their shared L2), stream W, then read the whole 64 KB — by then mostly L2 hits
    const int k = (blockIdx.x % 64) / 8, part = nfresh / 8;
    for (int i = threadIdx.x; i < part; i += 512) acc ^= xin[k * part + i];
    read_w_all();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_fresh();
  } else if (mode == 0 || fetcher) { read_fresh(); read_w_all(); }
  else if (mode == 1) { read_w_all(); }
  else { read_w_all(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); read_fresh(); }
  if (threadIdx.x < 64) xout[threadIdx.x] = u32x4{(unsigned)step, blockIdx.x, threadIdx.x, acc.x & 1u};
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[blockIdx.x] = acc.x;
}
