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
#include "common.h"

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

// ordered (deterministic) reduction of split-K partials
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias,
                                     int M, int N, int ldc, long long sC, int splitk, int accumulate) {
  const long long zb = blockIdx.y;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int row = idx / N, col = idx % N;
  const float* p = part + zb * splitk * (long long)M * N + idx;
  float s = 0.f;
  for (int k = 0; k < splitk; ++k) s += p[(long long)k * M * N];
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
  if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g, vecA, vecB);
  else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g, vecA, vecB);
  else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g, vecA, vecB);
  else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g, vecA, vecB);
  DS2_LAUNCH_CHECK("gemm_f32_kernel");
  if (splitk > 1) {
    dim3 rg(ceil_div(M * N, 256), batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, s, (const float*)workspace, C, bias, M, N, ldc, strideC,
                       splitk, accumulate);
    DS2_LAUNCH_CHECK("splitk_reduce_kernel");
  }
  return 0;
}
