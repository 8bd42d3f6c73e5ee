// fp32 GEMM on the CDNA4 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf-chain
// numerics at the 157 TF/s rate).  Replaces, for the DeepSpeech2 train step, the aten::addmm / mm
// calls inside aten::gru/lstm (input projections, blocks.py:76-78,88) and nn.Linear
// (deepspeech.py:105) plus their autograd backward (dX, dW).
//
// Row-major everywhere.  C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
//   transA == 0: A stored (M,K) lda   | transA == 1: A stored (K,M) lda
//   transB == 0: B stored (K,N) ldb   | transB == 1: B stored (N,K) ldb   ("NT": weights as stored)
// Batched (grid.z / splitk) and deterministic split-K (partials in workspace + ordered reduce).
//
// Tiling: 128x128x16 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32
// tiles (64 accumulator VGPRs).  Operand tiles are staged global -> registers -> LDS k-major
// ([k][m] / [k][n], row pitch 132 floats) so that MFMA fragment reads (lane = (m&31) + 32*(k&1))
// are unit-stride ds_read_b32 — conflict-free.  Global loads for tile kt+1 are in flight while
// tile kt is multiplied (2 LDS buffers, one barrier per K-tile).
#include <type_traits>
#include "common.h"
#include "permlane.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;  // LDT*4 bytes multiple of 16

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K, lda, ldb, ldc;
  long long sA, sB, sC;       // batch strides (elements)
  int splitk;                 // >= 1
  int kchunk;                 // K range per split (multiple of BK)
  int accumulate;             // C += (only when splitk == 1)
  float* partial;             // [batch][splitk][M][N] when splitk > 1
};

// Load 4 consecutive elements along the contiguous dimension with bounds/zero fill.
// `vec` => 16B-aligned fast path allowed.
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int valid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
  }
  return v;
}

// KCONT == true : operand stored (rows = m-or-n index, cols = k) -> tile 128 rows x 16 k
// KCONT == false: operand stored (rows = k, cols = m-or-n index) -> tile 16 k x 128
template <bool KCONT>
struct TileLoader {
  // two float4 per thread.  FULL: the whole 128 x 16 tile is in range and 16-byte aligned -> unconditional
  // vector loads, no control flow (keeps the loads of tile k+1 in flight across the MFMAs of tile k).
  template <bool FULL>
  __device__ static void load(const float* __restrict__ base, int ld, int r0, int rmax, int k0, int kmax,
                              bool vec, f32x4 (&reg)[2]) {
    const int tid = threadIdx.x;
    if (KCONT) {
      const int kq = tid & 3;
      const int k = k0 + 4 * kq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r0 + (tid >> 2) + 64 * i;
        const float* p = base + (long long)r * ld + k;
        if (FULL) reg[i] = *reinterpret_cast<const f32x4*>(p);
        else reg[i] = load4(p, (r < rmax) ? (kmax - k) : 0, vec);
      }
    } else {
      const int mq = tid & 31;
      const int m = r0 + 4 * mq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = k0 + (tid >> 5) + 8 * i;
        const float* p = base + (long long)k * ld + m;
        if (FULL) reg[i] = *reinterpret_cast<const f32x4*>(p);
        else reg[i] = load4(p, (k < kmax) ? (rmax - m) : 0, vec);
      }
    }
  }
  __device__ static void store(float* __restrict__ lds, const f32x4 (&reg)[2]) {
    const int tid = threadIdx.x;
    if (KCONT) {
      const int kq = tid & 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 2) + 64 * i;
        lds[(4 * kq + 0) * LDT + r] = reg[i].x;
        lds[(4 * kq + 1) * LDT + r] = reg[i].y;
        lds[(4 * kq + 2) * LDT + r] = reg[i].z;
        lds[(4 * kq + 3) * LDT + r] = reg[i].w;
      }
    } else {
      const int mq = tid & 31;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = (tid >> 5) + 8 * i;
        *reinterpret_cast<f32x4*>(&lds[k * LDT + 4 * mq]) = reg[i];
      }
    }
  }
};

template <bool TA, bool TB, bool FULL>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, const float* __restrict__ A, const float* __restrict__ B, int m0,
                                              int n0, int kbeg, int kend, int nkt, int vecA, int vecB, float* __restrict__ lds0,
                                              f32x16 (&acc)[2][2]) {
  // lds0: [2 buffers][A,B][BK*LDT]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 ra[2], rb[2];
  TileLoader<!TA>::template load<FULL>(A, g.lda, m0, g.M, kbeg, kend, vecA, ra);
  TileLoader<TB>::template load<FULL>(B, g.ldb, n0, g.N, kbeg, kend, vecB, rb);
  TileLoader<!TA>::store(lds0, ra);
  TileLoader<TB>::store(lds0 + BK * LDT, rb);
  __syncthreads();
  const int arow = wm * 64 + (lane & 31);
  const int brow = wn * 64 + (lane & 31);
  const int khalf = lane >> 5;
  // edge tiles only (FULL == false): a 32 x 32 MFMA tile that lies completely outside C is not computed (wave-uniform).  The fc block
  // runs N = 29 classes through this 128-wide tile: 14 of the 16 MFMAs per k-step of a block only fed columns that are never stored.
  const bool ua0 = FULL || m0 + wm * 64 < g.M, ua1 = FULL || m0 + wm * 64 + 32 < g.M;
  const bool ub0 = FULL || n0 + wn * 64 < g.N, ub1 = FULL || n0 + wn * 64 + 32 < g.N;
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) {
      const int k0 = kbeg + (kt + 1) * BK;
      TileLoader<!TA>::template load<FULL>(A, g.lda, m0, g.M, k0, kend, vecA, ra);
      TileLoader<TB>::template load<FULL>(B, g.ldb, n0, g.N, k0, kend, vecB, rb);
    }
    const float* As = lds0 + (cur * 2 + 0) * (BK * LDT);
    const float* Bs = lds0 + (cur * 2 + 1) * (BK * LDT);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int k = 2 * ks + khalf;
      const float a0 = As[k * LDT + arow];
      const float a1 = As[k * LDT + arow + 32];
      const float b0 = Bs[k * LDT + brow];
      const float b1 = Bs[k * LDT + brow + 32];
      if (FULL) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      } else {
        if (ua0 && ub0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        if (ua0 && ub1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        if (ua1 && ub0) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        if (ua1 && ub1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) {
      TileLoader<!TA>::store(lds0 + ((cur ^ 1) * 2 + 0) * (BK * LDT), ra);
      TileLoader<TB>::store(lds0 + ((cur ^ 1) * 2 + 1) * (BK * LDT), rb);
    }
    __syncthreads();
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g, int vecA, int vecB) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK * LDT];  // [buf][A/B]
  const int z = blockIdx.z;
  const int zb = z / g.splitk, zs = z % g.splitk;
  const float* A = g.A + (long long)zb * g.sA;
  const float* B = g.B + (long long)zb * g.sB;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = zs * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A operand: transA==0 -> stored (M,K): K-contiguous. transA==1 -> stored (K,M): M-contiguous.
  // B operand: transB==1 -> stored (N,K): K-contiguous. transB==0 -> stored (K,N): N-contiguous.
  const bool full = vecA && vecB && (m0 + BM <= g.M) && (n0 + BN <= g.N) && ((kend - kbeg) % BK == 0);   // block-uniform
  float* lds0 = &lds[0][0][0];
  if (full) gemm_mainloop<TA, TB, true>(g, A, B, m0, n0, kbeg, kend, nkt, vecA, vecB, lds0, acc);
  else gemm_mainloop<TA, TB, false>(g, A, B, m0, n0, kbeg, kend, nkt, vecA, vecB, lds0, acc);
  const int khalf = lane >> 5;

  // epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C;
  long long ldc;
  const bool partial = g.splitk > 1;
  if (partial) {
    C = g.partial + ((long long)zb * g.splitk + zs) * (long long)g.M * g.N;
    ldc = g.N;
  } else {
    C = g.C + (long long)zb * g.sC;
    ldc = g.ldc;
  }
  // bias loads before the first store: a load between stores waits for all earlier stores (one in-order vmcnt)
  float bvj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    bvj[j] = (!partial && g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= g.N) continue;
      const float bv = bvj[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          float* p = C + (long long)row * ldc + col;
          if (!partial && g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Skinny NT product C[M, N <= 32] = A[M, K] B[N, K]^T (+ bias): the fc layer's logits (nn.Linear 1024 -> 29 classes, deepspeech.py:105).
// The 128 x 128 tile above runs it at one k-tile of 16 per ~1.45 us (a load -> LDS -> barrier chain with one tile in flight, two of the four
// waves without a column to compute): 93 us for 131 MB of A.  Here a wave owns 32 rows for the whole K:
//   B          all of it in LDS as Bs[k][32] (zero columns past N): lane (n, h) reads Bs[2 j + h][n] for MFMA j — one contiguous 256 B per wave
//   A          straight from global into the MFMA layout: lane (m = lane & 31, h = lane >> 5) loads A[m][8 c + 4 h .. + 3]; two
//              v_permlane32_swap turn the lower half's (k0 k1 k2 k3) and the upper half's (k4 .. k7) into the four operand registers
//              [k0 | k1], [k2 | k3], [k4 | k5], [k6 | k7] (lower | upper lanes); 16 such loads in flight per lane
//   arithmetic the same v_mfma_f32_32x32x2_f32 chain in the same k order as gemm_f32_kernel (one accumulator per element, k ascending, even k
//              from the lower lanes): BIT-IDENTICAL results
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int SK_INFLIGHT = 16;
__global__ __launch_bounds__(256) void gemm_f32_skinny_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                 float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                                 int K) {
  extern __shared__ __attribute__((aligned(16))) float Bs[];     // [K][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * 128 + wave * 32;
  const bool rows = r0 < M;                                       // (wave-uniform; a wave without rows still helps to fill Bs)
  const int m = lane & 31, h = lane >> 5;
  const float* pa = A + (long long)min(r0 + m, M - 1) * lda + 4 * h;
  const float* pb = Bs + h * 32 + m;
  const int nc = K >> 3;                                          // K % 128 == 0: whole groups of SK_INFLIGHT chunks, no tail
  // The A stream is issued and awaited by hand (left to itself the compiler rotates the registers and waits for each load right behind its
  // issue): the loads return in order, so when chunk i is due at most SK_INFLIGHT - 1 younger loads are outstanding — vmcnt(15); the register
  // set of a chunk is re-loaded (16 chunks ahead) behind its four MFMAs.  Behind the last group the re-loads repeat its chunks.  The first
  // group goes out BEFORE the B fill: HBM latency and the fill's L2 round trips overlap.
  f32x4 q[SK_INFLIGHT];
#pragma unroll
  for (int i = 0; i < SK_INFLIGHT; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(q[i]) : "v"(pa + 8 * i) : "memory");
  {
    const int n = tid & 31, kq = tid >> 5;                        // 8 k-quads per pass, 8 passes in flight
    for (int k0 = kq * 4; k0 < K; k0 += 256) {
      f32x4 w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k4 = k0 + 32 * j;
        w[j] = (n < N && k4 < K) ? *reinterpret_cast<const f32x4*>(B + (long long)n * ldb + k4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k4 = k0 + 32 * j;
        if (k4 < K) { Bs[(k4 + 0) * 32 + n] = w[j].x; Bs[(k4 + 1) * 32 + n] = w[j].y; Bs[(k4 + 2) * 32 + n] = w[j].z; Bs[(k4 + 3) * 32 + n] = w[j].w; }
      }
    }
  }
  __syncthreads();
  if (!rows) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bc[4] = {pb[0], pb[64], pb[128], pb[192]};
  for (int c0 = 0; c0 < nc; c0 += SK_INFLIGHT) {
    const float* pn = pa + 8 * (c0 + SK_INFLIGHT < nc ? c0 + SK_INFLIGHT : c0);
    const float* b = pb + c0 * 256;                               // 8 k-rows of 32 per chunk
#pragma unroll
    for (int i = 0; i < SK_INFLIGHT; ++i) {
      // B values of the NEXT chunk (the last chunk of all re-reads itself): requested before this chunk's MFMAs
      const float* bn = b + ((c0 + i + 1 < nc) ? (i + 1) * 256 : i * 256);
      const float bn0 = bn[0], bn1 = bn[64], bn2 = bn[128], bn3 = bn[192];
      asm volatile("s_waitcnt vmcnt(15)" : "+v"(q[i]) : : "memory");
      const u32pair xy = permlane32_swap(q[i].x, q[i].y);         // a = [k0 | k1], b = [k4 | k5]
      const u32pair zw = permlane32_swap(q[i].z, q[i].w);         // a = [k2 | k3], b = [k6 | k7]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xy.a, bc[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zw.a, bc[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xy.b, bc[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zw.b, bc[3], acc, 0, 0, 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(q[i]) : "v"(pn + 8 * i) : "memory");
      bc[0] = bn0; bc[1] = bn1; bc[2] = bn2; bc[3] = bn3;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the trailing re-loads: nothing may land in a register that has been re-used)
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (m < N) {
    const float bv = bias ? bias[m] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < M) C[(long long)row * ldc + m] = acc[r] + bv;
    }
  }
}

// Skinny-K NN product C[M, N] = A[M, K <= 32] B[K, N]: the fc layer's input gradient dXn = dLogits W (K = 29 classes; autograd of nn.Linear,
// deepspeech.py:105).  The tile kernel reaches 2.2 TB/s on the 131 MB it writes (A rows of 29 floats take its scalar edge-load path, and a
// workgroup is two k-tiles of prologue for 64 stores per lane).  Here B sits in LDS as it lies in memory ([k][N]); a wave keeps the A fragments
// of its 32 rows in 15 registers (lane (m, h): A[m][2 j + h]) and walks the N / 32 column tiles: 15 ds_read_b32 + 15 MFMAs + 16 row stores
// (two full 128-byte lines each) per tile.  Same MFMA chain in the same k order (the tile kernel's sixteenth MFMA adds exact zeros): bit-identical.
__global__ __launch_bounds__(256) void gemm_f32_skinny_k_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                                int K) {
  extern __shared__ __attribute__((aligned(16))) float Bk[];     // [32][N], rows K .. 31 zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e0 = tid * 4; e0 < 32 * N; e0 += 8 * 1024) {           // 8 loads in flight per thread
    f32x4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 1024, k = e / N, n = e - k * N;
      w[u] = (e < 32 * N && k < K) ? *reinterpret_cast<const f32x4*>(B + (long long)k * ldb + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u * 1024 < 32 * N) *reinterpret_cast<f32x4*>(Bk + e0 + u * 1024) = w[u];
  }
  const int r0 = blockIdx.x * 128 + wave * 32;
  const int m = lane & 31, h = lane >> 5;
  float a[15];
  {
    const float* pa = A + (long long)min(r0 + m, M - 1) * lda;
#pragma unroll
    for (int j = 0; j < 15; ++j) a[j] = (2 * j + h < K) ? pa[2 * j + h] : 0.f;
  }
  __syncthreads();
  if (r0 >= M) return;
  const float* pb = Bk + h * N + m;
  // (no bias in this kernel: a load inside the tile loop would make every tile wait for the previous tile's stores — one in-order vmcnt)
  auto run = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    float bc[15], bn[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) bc[j] = pb[2 * j * N];
    for (int n0 = 0; n0 < N; n0 += 32) {
      const int nn = n0 + 32 < N ? n0 + 32 : n0;                  // the next tile's B values, requested before this tile's MFMAs
#pragma unroll
      for (int j = 0; j < 15; ++j) bn[j] = pb[2 * j * N + nn];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 15; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bc[j], acc, 0, 0, 0);      // (k >= K: exact zeros on both sides)
      float* pc = C + (long long)(r0 + 4 * h) * ldc + n0 + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (FULL || r0 + 4 * h + dr < M) pc[(long long)dr * ldc] = acc[r] + 0.f;     // (+ 0.f: the tile kernel's "+ bias" with no bias)
      }
#pragma unroll
      for (int j = 0; j < 15; ++j) bc[j] = bn[j];
    }
  };
  if (r0 + 32 <= M) run(std::true_type{}); else run(std::false_type{});
}

// Skinny-M TN product, split along K: partial[zs][M <= 32][N] = A[kbeg .. kend, M]^T B[kbeg .. kend, N] — the fc layer's weight gradient
// dW = dLogits^T Xn (29 classes x H, K = T*B; autograd of nn.Linear, deepspeech.py:105).  The tile kernel takes the 29-float rows of dLogits
// through its scalar edge loads, one 16-deep k-tile per HBM round trip.  Here one wave = (K slice, 32 columns): per MFMA one dword per lane of
// each operand straight from global in the MFMA layout (A: lane (m, h) <- A[k0 + 2 j + h][m], 29 consecutive floats per half, L2-resident;
// B: 128 contiguous bytes per half, streamed once), 16 MFMAs = 32 loads in flight per wave, counted by hand (vmcnt(30) at every MFMA).  Same
// slices, same k order inside a slice, the same ordered reduction behind it: bit-identical to the tile kernel's split-K result.
constexpr int SM_AHEAD = 16;
__global__ __launch_bounds__(256) void gemm_f32_skinny_m_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                float* __restrict__ part, int M, int N, int K, int kchunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int zs = blockIdx.y, n0 = blockIdx.x * 128 + wave * 32;
  const int kbeg = zs * kchunk, kend = min(K, kbeg + kchunk);
  const int m = lane & 31, h = lane >> 5;
  const bool mok = m < M;
  const int nj = (kend - kbeg + 1) >> 1;                          // MFMAs of this slice (wave-uniform), a multiple of SM_AHEAD is NOT required
  // element addresses of MFMA j: k = kbeg + 2 j + h, clamped to the last row of the slice (the value is then replaced by zero)
  const float* pa0 = A + (mok ? m : 0);
  const float* pb0 = B + n0 + m;
  auto ka = [&](int j) { const int k = kbeg + 2 * j + h; return k < kend ? k : kend - 1; };
  float av[SM_AHEAD], bv[SM_AHEAD];
#pragma unroll
  for (int i = 0; i < SM_AHEAD; ++i) {
    const long long k = ka(i < nj ? i : nj - 1);
    asm volatile("global_load_dword %0, %1, off" : "=&v"(av[i]) : "v"(pa0 + k * lda) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=&v"(bv[i]) : "v"(pb0 + k * ldb) : "memory");
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j0 = 0; j0 < nj; j0 += SM_AHEAD) {
#pragma unroll
    for (int i = 0; i < SM_AHEAD; ++i) {
      const int j = j0 + i;
      asm volatile("s_waitcnt vmcnt(30)" : "+v"(av[i]), "+v"(bv[i]) : : "memory");
      const bool real = j < nj && kbeg + 2 * j + h < kend;        // (past the slice: exact zeros, as the tile kernel's zero-filled k-tile)
      const float a = (real && mok) ? av[i] : 0.f, b = real ? bv[i] : 0.f;
      if (j < nj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      const int jn = j + SM_AHEAD < nj ? j + SM_AHEAD : nj - 1;   // (behind the end: re-reads of the last pair, unused)
      const long long k = ka(jn);
      asm volatile("global_load_dword %0, %1, off" : "=&v"(av[i]) : "v"(pa0 + k * lda) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=&v"(bv[i]) : "v"(pb0 + k * ldb) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float* out = part + (long long)zs * M * N + n0 + m;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < M) out[(long long)row * N] = acc[r];
  }
}

// ordered (deterministic) reduction of split-K partials
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias,
                                     int M, int N, int ldc, long long sC, int splitk, int accumulate) {
  const long long zb = blockIdx.y;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int row = idx / N, col = idx % N;
  const float* p = part + zb * splitk * (long long)M * N + idx;
  float s = 0.f;
  // same order of additions, the loads of 16 slabs in flight together (the fc weight gradient sums 118 slabs with 116 workgroups on the chip:
  // one dependent load + add per trip was 29 us for 14 MB)
  const long long slab = (long long)M * N;
  int k = 0;
  for (; k + 16 <= splitk; k += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(k + u) * slab];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; k < splitk; ++k) s += p[k * slab];
  if (bias) s += bias[col];
  float* c = C + zb * sC + (long long)row * ldc + col;
  if (accumulate) s += *c;
  *c = s;
}

}  // namespace

extern "C" size_t ds2_gemm_f32_workspace_bytes(int M, int N, int batch, int splitk) {
  if (splitk <= 1) return 0;
  return (size_t)batch * splitk * (size_t)M * N * sizeof(float);
}

extern "C" int ds2_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, long long strideA,
                            const float* B, int ldb, long long strideB, float* C, int ldc, long long strideC,
                            const float* bias, int accumulate, int batch, int splitk, void* workspace,
                            size_t workspace_bytes, void* stream) {
  DS2_REQUIRE(M > 0 && N > 0 && K >= 0 && batch >= 1, "ds2_gemm_f32: bad dims M=%d N=%d K=%d batch=%d", M, N, K, batch);
  DS2_REQUIRE(A && B && C, "ds2_gemm_f32: null pointer");
  if (splitk < 1) splitk = 1;
  int kchunk = ceil_div(ceil_div(K, splitk), BK) * BK;
  if (kchunk == 0) kchunk = BK;
  splitk = ceil_div(K, kchunk);
  if (splitk < 1) splitk = 1;
  if (splitk > 1)
    DS2_REQUIRE(workspace && workspace_bytes >= ds2_gemm_f32_workspace_bytes(M, N, batch, splitk),
                "ds2_gemm_f32: split-K workspace too small");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.splitk = splitk; g.kchunk = kchunk; g.accumulate = accumulate; g.partial = (float*)workspace;
  const int vecA = ((lda % 4) == 0) && (((uintptr_t)A % 16) == 0) && ((strideA % 4) == 0);
  const int vecB = ((ldb % 4) == 0) && (((uintptr_t)B % 16) == 0) && ((strideB % 4) == 0);
  dim3 grid(ceil_div(N, BN), ceil_div(M, BM), batch * splitk), block(256);
  hipStream_t s = (hipStream_t)stream;
  // skinny NT (the fc logits): gemm_f32_skinny_nt_kernel, bit-identical to the tile kernel.  DS2_GEMM_SKINNY=0: the tile kernel (A/B switch)
  static const char* skinny_env = ds2_exp_getenv("DS2_GEMM_SKINNY");
  if (!transA && transB && N <= 32 && M >= 1024 && K >= 128 && (K % 128) == 0 && (size_t)K * 128 <= 160 * 1024 && batch == 1 && splitk == 1 && !accumulate &&
      vecA && vecB && !(skinny_env && skinny_env[0] == '0')) {
    static bool attr = false;
    if (!attr) {
      DS2_HIP(hipFuncSetAttribute((const void*)gemm_f32_skinny_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(gemm_f32_skinny_nt_kernel, dim3(ceil_div(M, 128)), dim3(256), (size_t)K * 128, s, A, lda, B, ldb, C, ldc, bias, M, N, K);
    DS2_LAUNCH_CHECK("gemm_f32_skinny_nt_kernel");
    return 0;
  }
  // skinny-K NN (the fc layer's input gradient): gemm_f32_skinny_k_kernel, bit-identical to the tile kernel
  // (one workgroup per CU — W fills the LDS —, 128 rows each: worth it from ~3/4 of the chip's CUs upwards; c2's 16000 rows stay on the tile kernel)
  if (!transA && !transB && K >= 1 && K <= 30 && (N % 32) == 0 && (size_t)N * 128 <= 160 * 1024 && M >= 20000 && batch == 1 && splitk == 1 && !accumulate &&
      !bias && vecB && !(skinny_env && skinny_env[0] == '0')) {
    static bool attr = false;
    if (!attr) {
      DS2_HIP(hipFuncSetAttribute((const void*)gemm_f32_skinny_k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(gemm_f32_skinny_k_kernel, dim3(ceil_div(M, 128)), dim3(256), (size_t)N * 128, s, A, lda, B, ldb, C, ldc, bias, M, N, K);
    DS2_LAUNCH_CHECK("gemm_f32_skinny_k_kernel");
    return 0;
  }
  // skinny-M split-K TN (the fc layer's weight gradient): gemm_f32_skinny_m_kernel writes the same slabs as the tile kernel, bit for bit
  const bool skinny_m = transA && !transB && M <= 32 && (N % 128) == 0 && splitk > 1 && batch == 1 && kchunk >= 64 && !(skinny_env && skinny_env[0] == '0');
  if (skinny_m) {
    hipLaunchKernelGGL(gemm_f32_skinny_m_kernel, dim3(N / 128, splitk), dim3(256), 0, s, A, lda, B, ldb, (float*)workspace, M, N, K, kchunk);
    DS2_LAUNCH_CHECK("gemm_f32_skinny_m_kernel");
  } else if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g, vecA, vecB);
  else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g, vecA, vecB);
  else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g, vecA, vecB);
  else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g, vecA, vecB);
  if (!skinny_m) DS2_LAUNCH_CHECK("gemm_f32_kernel");
  if (splitk > 1) {
    dim3 rg(ceil_div(M * N, 256), batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, s, (const float*)workspace, C, bias, M, N, ldc, strideC,
                       splitk, accumulate);
    DS2_LAUNCH_CHECK("splitk_reduce_kernel");
  }
  return 0;
}
