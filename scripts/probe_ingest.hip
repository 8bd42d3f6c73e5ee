// Round 5 probe: how fast ONE CU can take operand bytes from the L2 / from HBM, by instruction form — the bound of the 256 x 256 GEMM tile
// (64 KB of operands per 64-deep k-tile and CU).  256 workgroups x 512 threads; every workgroup streams `span` bytes of its own region over and
// over (span = 64 KB: L2-resident; span = 16 MB: streaming), 1 KiB per wave-instruction.
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_ingest scripts/probe_ingest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// MODE 0: global_load_lds dwordx4, rows of 128 B (8 rows per instruction); 1: the same with rows of 64 B (16 rows per instruction, row pitch 2 KB)
// MODE 2: global_load_dwordx4 -> VGPR, consumed by an add; 3: ... -> VGPR -> ds_write_b128
// MIX (MODE 4 / 5): of every 8 pieces MIXHOT come from the workgroup's L2-resident 64 KB, the others stream through its 16 MB (HBM):
// does a share of HBM-latency pieces in the in-order return queue pull the whole stream down to the HBM latency?
template <int MIXHOT, int DEPTH>
__global__ __launch_bounds__(512) void kmix(const char* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* hot = src + (long long)blockIdx.x * (16LL << 20);
  const char* cold = hot + 65536;
  const long long loff = (long long)(lane >> 3) * 128 + (lane & 7) * 16;
  long long ph = 0, pc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const bool is_hot = (d % 8) < MIXHOT;
      const char* g = is_hot ? hot + (ph * 8 + wave) * 1024 + loff : cold + (pc * 8 + wave) * 1024 + loff;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(lds + ((it & 1) * DEPTH + d) * 8192 + wave * 1024), 16, 0, 0);
      if (is_hot) { if (++ph >= 8) ph = 0; } else { if (++pc >= 2040) pc = 0; }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 accv = *reinterpret_cast<f32x4*>(lds + threadIdx.x * 16);
  out[blockIdx.x * 512 + threadIdx.x] = accv[0] + accv[1] + accv[2] + accv[3];
}
template <int MIXHOT, int DEPTH>
void runmix(const char* name, const char* src, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096 / DEPTH * 2;
  (void)hipFuncSetAttribute((const void*)kmix<MIXHOT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DEPTH * 8192);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kmix<MIXHOT, DEPTH>), dim3(256), dim3(512), 2 * DEPTH * 8192, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_per_cu = (double)iters * DEPTH * 8 * 1024;
  printf("%-64s %7.1f GB/s per CU  (%5.2f TB/s chip, HBM share %4.2f TB/s)\n", name, bytes_per_cu / best / 1e6, bytes_per_cu * 256 / best / 1e9,
         bytes_per_cu * 256 / best / 1e9 * (8 - MIXHOT) / 8);
}


// SHARED cold stream (the GEMM's situation): the 4 workgroups g, g + 8, g + 16, g + 24 of a group (same XCD under round-robin dispatch) stream
// the SAME 16 MB in step — every line is fetched from HBM once and requested by all four while it is still in flight, so all four wait
// for it.  SPLIT = false: every workgroup requests the pieces in the same order.  SPLIT = true: of every 4 consecutive pieces a workgroup
// requests ITS OWN one FAR ahead (AHEAD iterations) and the other three at the normal distance — by then its siblings' far requests have
// brought them into the L2: each workgroup waits for HBM on a quarter of the bytes only.
template <bool SPLIT, int DEPTH, int AHEAD>
__global__ __launch_bounds__(512) void kshare(const char* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5), me = (blockIdx.x >> 3) & 3;      // 64 groups of 4
  const char* base = src + (long long)grp * (16LL << 20);
  const long long loff = (long long)(lane >> 3) * 128 + (lane & 7) * 16;
  const long long npieces = (16LL << 20) / 8192;                                   // 8 KB per (iteration step): 8 waves x 1 KB
  long long p = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      long long pp = p;
      if (SPLIT) pp = ((int)(p & 3) == me) ? p + 4 * AHEAD : p;                     // own quarter of every 4 pieces: far ahead
      if (pp >= npieces) pp -= npieces;
      const char* g = base + (pp * 8 + wave) * 1024 + loff;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(lds + ((it & 1) * DEPTH + d) * 8192 + wave * 1024), 16, 0, 0);
      if (++p >= npieces) p = 0;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 accv = *reinterpret_cast<f32x4*>(lds + threadIdx.x * 16);
  out[blockIdx.x * 512 + threadIdx.x] = accv[0] + accv[1] + accv[2] + accv[3];
}
template <bool SPLIT, int DEPTH, int AHEAD>
void runshare(const char* name, const char* src, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096 / DEPTH * 2;
  (void)hipFuncSetAttribute((const void*)kshare<SPLIT, DEPTH, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DEPTH * 8192);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kshare<SPLIT, DEPTH, AHEAD>), dim3(256), dim3(512), 2 * DEPTH * 8192, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_per_cu = (double)iters * DEPTH * 8 * 1024;
  printf("%-76s %7.1f GB/s per CU  (HBM side: %5.2f TB/s)\n", name, bytes_per_cu / best / 1e6, bytes_per_cu * 64 / best / 1e9);
}


// the GEMM's A-operand pattern under sharing: 4 workgroups of a group stream the same (256 rows x K) panel, k-slice by k-slice; a wave
// instruction takes ROWS rows x (1024 / ROWS) bytes, row pitch 12288 B (dX: K = 6144 bf16); consecutive k-slices continue along the rows.
template <int ROWS>
__global__ __launch_bounds__(512) void kpanel(const char* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int SEG = 1024 / ROWS, LPR = SEG / 16;                       // bytes per row per instruction, lanes per row
  constexpr int PITCH = 12288, PIECES = 256 / ROWS / 8;                  // wave-instructions per wave per k-slice
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5);
  const char* base = src + (long long)grp * (16LL << 20);                // 5 panels of 256 x 12288 B in 16 MB
  const long long loff = (long long)(lane / LPR) * PITCH + (lane % LPR) * 16;
  int ks = 0, panel = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int piece = d % PIECES;
      const char* g = base + (long long)panel * 256 * PITCH + (long long)((wave * PIECES + piece) * ROWS) * PITCH + (long long)ks * SEG + loff;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(lds + ((it & 1) * 8 + d) * 8192 + wave * 1024), 16, 0, 0);
      if (piece == PIECES - 1) { if (++ks >= PITCH / SEG) { ks = 0; if (++panel >= 5) panel = 0; } }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 accv = *reinterpret_cast<f32x4*>(lds + threadIdx.x * 16);
  out[blockIdx.x * 512 + threadIdx.x] = accv[0] + accv[1] + accv[2] + accv[3];
}
template <int ROWS>
void runpanel(const char* name, const char* src, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1024;
  (void)hipFuncSetAttribute((const void*)kpanel<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 8192);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kpanel<ROWS>), dim3(256), dim3(512), 2 * 8 * 8192, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_per_cu = (double)iters * 8 * 8 * 1024;
  printf("%-76s %7.1f GB/s per CU  (HBM side: %5.2f TB/s)\n", name, bytes_per_cu / best / 1e6, bytes_per_cu * 64 / best / 1e9);
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, float* __restrict__ out, long long span, int iters, long long pitch) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long long)blockIdx.x * span;
  // per-lane source offset inside a 1 KiB piece
  long long loff;
  if (MODE == 1) loff = (long long)(lane >> 2) * pitch + (lane & 3) * 16;      // 16 rows x 64 B
  else loff = (long long)(lane >> 3) * pitch + (lane & 7) * 16;                // 8 rows x 128 B
  const long long piece_bytes = (MODE == 1 ? 16 : 8) * pitch;                  // source distance between consecutive pieces of one wave
  const long long npieces = span / piece_bytes / 8;                             // pieces per wave per sweep
  f32x4 accv = {0, 0, 0, 0};
  long long p = 0;
  if (MODE <= 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = base + (p * 8 + wave) * piece_bytes + loff;
        __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(lds + ((it & 1) * DEPTH + d) * 8192 + wave * 1024), 16, 0, 0);
        if (++p >= npieces) p = 0;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");   // keep DEPTH .. 2 DEPTH in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    accv = *reinterpret_cast<f32x4*>(lds + threadIdx.x * 16);
  } else {
    f32x4 r[2][DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[0][d] = r[1][d] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {      // consume the set loaded two half-iterations ago, then refill it
          if (MODE == 3) *reinterpret_cast<f32x4*>(lds + (h * DEPTH + d) * 8192 + threadIdx.x * 16) = r[h][d];
          else accv += r[h][d];
          const char* g = base + (p * 8 + wave) * piece_bytes + loff;
          r[h][d] = *reinterpret_cast<const f32x4*>(g);
          if (++p >= npieces) p = 0;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) accv += r[0][d] + r[1][d];
    if (MODE == 3) accv += *reinterpret_cast<f32x4*>(lds + threadIdx.x * 16);
  }
  out[blockIdx.x * 512 + threadIdx.x] = accv[0] + accv[1] + accv[2] + accv[3];
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, float* out, long long span, long long pitch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096 / DEPTH * 2;
  (void)hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DEPTH * 8192);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(512), 2 * DEPTH * 8192, 0, src, out, span, iters, pitch);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_per_cu = (double)iters * DEPTH * 8 * 1024;
  printf("%-64s span %6lld KB pitch %5lld: %7.1f GB/s per CU  (%5.2f TB/s chip)\n", name, span >> 10, pitch, bytes_per_cu / best / 1e6, bytes_per_cu * 256 / best / 1e9);
}

int main() {
  const long long total = 256LL * (16 << 20);
  char* src; float* out;
  (void)hipMalloc(&src, total); (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemset(src, 1, total);
  for (long long span : {65536LL, 16LL << 20}) {
    run<0, 4>("global_load_lds x4, 8 rows x 128 B, 4-8 in flight per wave", src, out, span, 128);
    run<0, 8>("global_load_lds x4, 8 rows x 128 B, 8-16 in flight per wave", src, out, span, 128);
    run<1, 8>("global_load_lds x4, 16 rows x 64 B, 8-16 in flight per wave", src, out, span, 64);
    run<0, 8>("global_load_lds x4, 8 rows x 128 B of 2 KB-pitch rows, 8-16 in flight", src, out, span, 2048);
    run<1, 8>("global_load_lds x4, 16 rows x 64 B of 2 KB-pitch rows, 8-16 in flight", src, out, span, 2048);
    run<2, 4>("global_load_dwordx4 -> VGPR, 4-8 in flight per wave", src, out, span, 128);
    run<2, 8>("global_load_dwordx4 -> VGPR, 8-16 in flight per wave", src, out, span, 128);
    run<2, 8>("global_load_dwordx4 -> VGPR, 2 KB-pitch rows, 8-16 in flight", src, out, span, 2048);
    run<3, 8>("global_load_dwordx4 -> VGPR -> ds_write_b128, 8-16 in flight", src, out, span, 128);
  }
  runmix<7, 8>("glds mix 7/8 L2-resident + 1/8 HBM stream, 8-16 in flight", src, out);
  runmix<6, 8>("glds mix 6/8 L2-resident + 2/8 HBM stream, 8-16 in flight", src, out);
  runmix<4, 8>("glds mix 4/8 L2-resident + 4/8 HBM stream, 8-16 in flight", src, out);
  runpanel<16>("shared 256-row panel, pitch 12 KB: 16 rows x 64 B per instruction (ring slices)", src, out);
  runpanel<8>("shared 256-row panel, pitch 12 KB: 8 rows x 128 B per instruction (64-deep k-tiles)", src, out);
  runpanel<4>("shared 256-row panel, pitch 12 KB: 4 rows x 256 B per instruction", src, out);
  runpanel<2>("shared 256-row panel, pitch 12 KB: 2 rows x 512 B per instruction", src, out);
  runshare<false, 8, 0>("4 workgroups share one HBM stream, same request order", src, out);
  runshare<true, 8, 4>("4 workgroups share one HBM stream, own quarter 4 iterations (256 KB) ahead", src, out);
  runshare<true, 8, 16>("4 workgroups share one HBM stream, own quarter 16 iterations (1 MB) ahead", src, out);
  runshare<true, 8, 64>("4 workgroups share one HBM stream, own quarter 64 iterations (4 MB) ahead", src, out);
  return 0;
}
