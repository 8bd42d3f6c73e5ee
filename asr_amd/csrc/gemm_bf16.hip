// bf16-operand / fp32-accumulate GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16, ~2.5 PF/s dense)
// for the dense input-to-hidden GEMMs of the RNN gates — the one place the north-star asks for MFMA —
// plus the fp32 -> bf16 cast / transposing-cast passes that feed it.
//
// ONE kernel form: "NT",  C[M,N] (fp32) (+)= A[M,K] (bf16) * B[N,K]^T (bf16) (+ bias[N]); both operands K-contiguous,
// so every staging load is a 16-byte run and every MFMA fragment a single ds_read_b128.  The other two GEMM
// forms of the backward pass are brought to NT by a transposing cast of one operand (an HBM-bound pass that
// costs ~1% of the GEMM it enables):
//     forward   Gx  = Xn W_ih^T        A = bf16(Xn)        B = bf16(W_ih)
//     dXn       = dGx W_ih             A = bf16(dGx)       B = bf16(W_ih)^T   (transposing cast of the weights)
//     dW_ih     = dGx^T Xn             A = bf16(dGx)^T     B = bf16(Xn)^T     (transposing casts, K = T*B)
//
// Two kernels.  Large problems: 256x256x64 tiles staged by LDS-DMA into a double-buffered, XOR-swizzled LDS image (below).
// Small ones: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles x 4 k-steps:
// LDS tile rows are 64 bf16 = 128 B + 16 B pad (pitch 144 B): ds_write_b128 by 8-lane groups and
// ds_read_b128 by the MFMA's 16-lane groups are both bank-conflict-free.  Global loads of tile k+1 are issued
// before the MFMAs of tile k and parked in registers; out-of-range rows / k-segments read a zero page
// (pointer select before the load, no control flow in the loop).
#include "common.h"
#include <type_traits>
#include <cstdint>
#include <stdlib.h>
#include <string.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TN_MAX_PROBLEMS = 16;            // products per grouped TN launch (gemm_bf16_tn_glds_kernel<true>, gemm_tn_group.h)
constexpr int PITCH = 144;                       // bytes per LDS tile row (128 + 16 pad)
constexpr int TILE_BYTES = BM * PITCH;           // 18432

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // global address space zero page
__device__ __attribute__((aligned(16))) float g_sink16[4];                          // where store lanes outside the matrix write (gemm_nt_ring.h)

struct BArgs {
  const __bf16* A; const __bf16* B; float* C; const float* bias;
  int M, N, K, lda, ldb, ldc;
  long long sA, sB, sC;
  int splitk, kchunk, accumulate;
  int nt_store;        // final C written with non-temporal stores (streamed out once)
  int super_rows;      // 256 x 256 NT kernels: row tiles per super-row of the XCD-aware tile walk (0: row-major)
  float* partial;
};

// 4 x 16-byte loads per thread per operand tile: row = (tid >> 3) + 32*i, k-segment = tid & 7 (8 bf16 each)
__device__ __forceinline__ void load_tile(const __bf16* __restrict__ base, int ld, int r0, int rmax, int k0, int kmax, f32x4 (&reg)[4]) {
  const int seg = threadIdx.x & 7;
  const int k = k0 + seg * 8;
  const bool kok = k + 8 <= kmax;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + (threadIdx.x >> 3) + 32 * i;
    const bool ok = kok && r < rmax;
    const void* p = ok ? (const void*)(base + (long long)r * ld + k) : (const void*)g_zero16;
    reg[i] = *reinterpret_cast<const f32x4*>(p);
  }
}
__device__ __forceinline__ void store_tile(char* __restrict__ lds, const f32x4 (&reg)[4]) {
  const int seg = threadIdx.x & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (threadIdx.x >> 3) + 32 * i;
    *reinterpret_cast<f32x4*>(lds + r * PITCH + seg * 16) = reg[i];
  }
}

__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(BArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];   // [A | B]
  const int z = blockIdx.z;
  const int zb = z / g.splitk, zs = z % g.splitk;
  const __bf16* A = g.A + (long long)zb * g.sA;
  const __bf16* B = g.B + (long long)zb * g.sB;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = zs * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[4], rb[4];
  load_tile(A, g.lda, m0, g.M, kbeg, kend, ra);
  load_tile(B, g.ldb, n0, g.N, kbeg, kend, rb);
  char* As = lds;
  char* Bs = lds + TILE_BYTES;
  const char* afrag = As + (wm * 64 + l31) * PITCH + half * 16;   // + mi*32*PITCH + kk*32
  const char* bfrag = Bs + (wn * 64 + l31) * PITCH + half * 16;

  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();                    // previous tile's fragment reads are done
    store_tile(As, ra);
    store_tile(Bs, rb);
    __syncthreads();
    if (kt + 1 < nkt) {
      const int k0 = kbeg + (kt + 1) * BK;
      load_tile(A, g.lda, m0, g.M, k0, kend, ra);
      load_tile(B, g.ldb, n0, g.N, k0, kend, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(afrag + kk * 32);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(afrag + 32 * PITCH + kk * 32);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bfrag + kk * 32);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bfrag + 32 * PITCH + kk * 32);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

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
  float bvj[2];                                   // bias loads before the first store (see the 256-tile kernel)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + l31;
    bvj[j] = (!partial && g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = bvj[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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

// ---- 256 x 256 x 64, LDS-DMA staged, double-buffered: one barrier per k-tile -------------------------------------
// 8 waves (2 x 4), each wave 128 x 64 = 4 x 2 MFMA 32x32x16 tiles.  Operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass): one wave-instruction lands 8 rows x 128 B = 1 KiB
// lane-linearly, so the LDS image has 128-B rows with NO padding; bank conflicts of the fragment reads are removed by
// an XOR swizzle applied on the per-lane SOURCE address and again on the read: the 16-byte slot p of row r holds
// k-segment p ^ ((r >> 1) & 7), which puts the 16 rows of every ds_read_b128 lane group on 16 distinct bank quads.
// Main loop: one barrier per k-tile, two tiles of DMA in flight, fragment reads one k-step ahead in the MFMA shadows (see the loop).
constexpr int G_TILE = 256 * 128;                 // bytes of one operand tile (256 rows x 64 bf16)
constexpr int G_LDS = 4 * G_TILE;                 // [2 buffers][A | B] = 128 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

// WM x WN waves; each wave owns a (256/WM) x (256/WN) sub-tile.  2 x 4 (8 waves, 128 x 64 per wave) is the production shape.
// PP (ping-pong, 2 x 4 waves only): the two wave groups (rows 0-127 / 128-255 of the tile = one wave per SIMD each) run the k-steps
// half a step out of phase, held there by two raw barriers per k-step: while one group issues its 8 MFMAs the other reads its next
// fragments and issues DMA, so the matrix pipe of every SIMD always has a wave in its MFMA cluster.
// PERS (2 x 4 waves, no split-K, K a multiple of 128): PERSISTENT form for short reductions.  With K = 1024 a tile is 16 k-tiles = 29 us of
// main loop, and a workgroup that computes ONE tile pays ~12 us around it (dispatch, pointer set-up, the first operand DMA's round trip,
// the write-out of 256 KB): 29 % of the forward projection.  Here the grid is one workgroup per CU and each walks tiles orig, orig + grid,
// ...: the k-tile stream simply continues across the tile boundary — the DMA of the NEXT tile's first two k-tiles goes out in the last two
// k-steps of the current one, its first fragments are read behind the last MFMAs as always, and the epilogue (wave-private LDS patch outside
// the operand buffers, bias fetched at the tile's start) runs while that DMA lands.  Same products, same order: bit-identical results.
constexpr int G_PATCH = 8 * 16 * 40 * 4;          // PERS epilogue: 8 wave-private patches of 16 rows x 40 floats behind the operand buffers
template <int WM, int WN, bool PP = false, bool PERS = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_glds_kernel(BArgs g, int ntx, int nty) {
  static_assert(!PP || (WM == 2 && WN == 4), "ping-pong schedule is written for 2 x 4 waves");
  static_assert(!PERS || (WM == 2 && WN == 4 && !PP), "persistent form: 2 x 4 waves, software-pipelined loop");
  constexpr int NWV = WM * WN;
  constexpr int NI = 8 / WM, NJ = 8 / WN;   // 32 x 32 MFMA tiles per wave along M / N
  constexpr int NP = 32 / NWV;              // 8-row pieces per wave per operand tile
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const int z = blockIdx.z;
  const int zb = z / g.splitk, zs = z % g.splitk;
  const __bf16* A = g.A + (long long)zb * g.sA;
  const __bf16* B = g.B + (long long)zb * g.sB;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so give each XCD one contiguous run of the
  // (row-tile major) tile list: the tiles sharing an A row-tile then hit the same 4 MB L2.  (bijective for any count)
  const int nt = ntx * nty;
  int orig = blockIdx.x;
  // (g.super_rows > 0: the sequence runs through super-rows of that many row tiles column by column, so that the 32 tiles an XCD holds at one
  // time form a compact block — see gemm_nt_ring.h)
  auto tile_origin = [&](int o, int& tm0, int& tn0) {
    const int xcd = o & 7, q8 = nt >> 3, r8 = nt & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
    if (g.super_rows > 0) {
      const int per = g.super_rows * ntx, sr = tile / per, rem = tile - sr * per;
      const int rows = min(g.super_rows, nty - sr * g.super_rows);
      const int tn = rem / rows;
      tm0 = (sr * g.super_rows + rem - tn * rows) * 256; tn0 = tn * 256;
    } else {
      tm0 = (tile / ntx) * 256; tn0 = (tile % ntx) * 256;
    }
  };
  int m0, n0;
  tile_origin(orig, m0, n0);
  const int kbeg = zs * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  // staging: wave w moves pieces w, w+8, w+16, w+24 (8 rows each) of A and of B
  const int prow = lane >> 3;                       // row inside the piece
  const __bf16* srcA[NP];
  const __bf16* srcB[NP];
  bool okA[NP], okB[NP];
  int segk[NP];                                     // k offset (elements) of the 16-byte segment this lane fetches
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = (wave + NWV * i) * 8 + prow;        // row inside the 256-row tile
    segk[i] = (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    okA[i] = m0 + r < g.M;
    okB[i] = n0 + r < g.N;
    srcA[i] = A + (long long)(m0 + r) * g.lda + segk[i];
    srcB[i] = B + (long long)(n0 + r) * g.ldb + segk[i];
  }
  auto stage_part = [&](int buf, int k0, int i) {     // one A piece + one B piece (2 of the wave's 8 DMA instructions)
    char* dA = ldsg + buf * 2 * G_TILE;
    char* dB = dA + G_TILE;
    const bool kok = k0 + segk[i] + 8 <= kend;
    const void* pa = (okA[i] && kok) ? (const void*)(srcA[i] + k0) : (const void*)g_zero16;
    const void* pb = (okB[i] && kok) ? (const void*)(srcB[i] + k0) : (const void*)g_zero16;
    glds16(pa, dA + (wave + NWV * i) * 1024);
    glds16(pb, dB + (wave + NWV * i) * 1024);
  };
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) stage_part(buf, k0, i);
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: row (.. + l31), k-segment kk*2 + half, swizzled with ((row >> 1) & 7) = (l31 >> 1) & 7
  const int sx = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = (((kk * 2 + half) ^ sx) << 4);
  const int arow = (wm * (NI * 32) + l31) * 128;
  const int brow = G_TILE + (wn * (NJ * 32) + l31) * 128;

  if constexpr (PP) {
    stage(0, kbeg);
    // Schedule per k-step (kk):  [read fragments kk | DMA slice | waits]  BARRIER  [8 MFMAs]  BARRIER.  Group B (wm = 1) runs one
    // barrier behind group A, so A's MFMA section always coincides with B's read section and vice versa.
    //  * fragments of a k-step are complete (lgkmcnt 0) before its first barrier, so a buffer is dead once both groups have passed
    //    that barrier of the tile's last k-step -> the DMA of tile kt+1 (into the other buffer) may start in k-step 0 of tile kt;
    //  * every wave waits for its own DMA of tile kt+1 (vmcnt 0) in k-step 3 of tile kt, before its first barrier: when group A
    //    passes the last barrier of tile kt, all eight waves have done so.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // phase offset of group B
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = kt + 1 < nkt;                   // block-uniform
      const char* base = ldsg + (kt & 1) * 2 * G_TILE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a[NI], b[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(base + arow + i * 32 * 128 + koff[kk]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + brow + j * 32 * 128 + koff[kk]);
        if (more) {
          if (kk == 0) { stage_part((kt + 1) & 1, kbeg + (kt + 1) * BK, 0); stage_part((kt + 1) & 1, kbeg + (kt + 1) * BK, 1); }
          if (kk == 1) stage_part((kt + 1) & 1, kbeg + (kt + 1) * BK, 2);
          if (kk == 2) stage_part((kt + 1) & 1, kbeg + (kt + 1) * BK, 3);
          if (kk == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // group A catches the barrier group B is one behind on
  } else {
    // ---- software-pipelined main loop, ONE barrier per k-tile.  The fragments of k-step kk+1 (for kk = 3: of the NEXT tile's
    // k-step 0) are read while the MFMAs of k-step kk issue, so the matrix pipe never waits for an LDS round trip; the barrier sits in
    // front of the last k-step, behind 8 MFMAs that are still executing, and the DMA of tile kt+2 goes out right behind it into the
    // buffer whose last fragment read that barrier has just retired.  Two tiles of DMA are in flight at any time.
    //   order per tile:  R(1) M(0) | R(2) M(1) | R(3) M(2) | wait own DMA + own reads, BARRIER | DMA(kt+2) R(next 0) M(3)
    // LDS-DMA visibility: a tile is read only after every wave's vmcnt(0) AND a barrier (guide: "one barrier after the wait").
    const char* zp = reinterpret_cast<const char*>(g_zero16);
    // rows beyond M / N are CLAMPED to the last valid row instead of zero-filled: they only feed C rows / columns that are never stored
    const char* qA[NP];
    const char* qB[NP];
    auto retarget = [&](int tm0, int tn0) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int r = (wave + NWV * i) * 8 + prow;
        qA[i] = reinterpret_cast<const char*>(A + (long long)min(tm0 + r, g.M - 1) * g.lda + segk[i] + kbeg);
        qB[i] = reinterpret_cast<const char*>(B + (long long)min(tn0 + r, g.N - 1) * g.ldb + segk[i] + kbeg);
      }
    };
    retarget(m0, n0);
    bool more_tiles = false;                            // PERS: another tile follows this one (its first k-tiles are staged from inside this one)
    const int nfull = (g.nt_store & 2) ? 1 : (kend - kbeg) / BK;       // tiles that lie completely inside [kbeg, kend)  (bit 1: timing experiment)
    // piece i (8 rows of A + 8 rows of B per wave) of tile kt -> buffer buf.  Tiles are staged in order, so the pointers just advance.
    auto stage_piece = [&](int buf, int kt, int i) {
      char* dA = ldsg + buf * 2 * G_TILE + (wave + NWV * i) * 1024;
      if (PERS || kt < nfull) {                       // (PERS: K is a multiple of the k-tile; `kt` may run into the next tile)
        glds16(qA[i], dA);
        glds16(qB[i], dA + G_TILE);
        qA[i] += BK * 2;
        qB[i] += BK * 2;
      } else if (!(g.nt_store & 2)) {                 // the K tail: per-segment bounds (the pointers already stand on this tile)
        const bool kok = kbeg + kt * BK + segk[i] + 8 <= kend;
        glds16(kok ? qA[i] : zp, dA);
        glds16(kok ? qB[i] : zp, dA + G_TILE);
      }
    };
    auto stage_tile = [&](int buf, int kt) {
#pragma unroll
      for (int i = 0; i < NP; ++i) stage_piece(buf, kt, i);
    };
    // The fragment reads and their waits are inline asm, and every read is issued in the execution shadow of an MFMA (one read
    // behind each of the first NI + NJ MFMAs of a k-step): a 32x32x16 MFMA occupies the matrix pipe for 32 cycles during which its
    // wave is free to issue other instructions, whereas reads issued in a block in front of the MFMAs cost ~20 % of the loop when the
    // two waves of a SIMD run in step.  Left to the compiler, every wait for an LDS read also becomes lgkmcnt(0) in FRONT of the MFMAs.
    // Each wait names the registers it retires as in/out operands, so the MFMAs that consume them cannot be scheduled above it.
    f32x4 fa[2][NI], fb[2][NJ];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;
    unsigned ra[4], rb[4];                              // byte address of this lane's fragment slot per k-step, buffer 0
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { ra[kk] = lds0 + arow + koff[kk]; rb[kk] = lds0 + brow + koff[kk]; }
    static_assert(NI + NJ <= NI * NJ + 2 && NI <= 4 && NJ <= 2, "interleave below");
    constexpr int NR = NI + NJ;                          // fragment reads per k-step
    // read r of a k-step, in the order the next k-step consumes them: A0, B0 .. B(NJ-1), A1 .. A(NI-1)
#define G_ISA(r_) ((r_) == 0 || (r_) > NJ)
#define G_IDX(r_) ((r_) == 0 ? 0 : (r_) <= NJ ? (r_) - 1 : (r_) - NJ)
#define G_RD1(set, bufoff, kk, r_)                                                                                                    \
  do {                                                                                                                                \
    if ((r_) < NR && G_ISA(r_))                                                                                                       \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][(r_) < NR && G_ISA(r_) ? G_IDX(r_) : 0]) : "v"(ra[kk] + (bufoff)), \
                   "n"(((r_) < NR && G_ISA(r_) ? G_IDX(r_) : 0) * 4096));                                                             \
    else if ((r_) < NR)                                                                                                               \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][(r_) < NR && !G_ISA(r_) ? G_IDX(r_) : 0]) : "v"(rb[kk] + (bufoff)), \
                   "n"(((r_) < NR && !G_ISA(r_) ? G_IDX(r_) : 0) * 4096));                                                            \
  } while (0)
    // every LDS read of this wave has returned (both fragment sets complete); WAITSTR may add vmcnt(0)
#define G_RETIRE_ALL(WAITSTR)                                                                                                         \
  asm volatile(WAITSTR " lgkmcnt(0)"                                                                                                  \
               : "+v"(fa[0][0]), "+v"(fa[0][NI > 1 ? 1 : 0]), "+v"(fa[0][NI > 2 ? 2 : 0]), "+v"(fa[0][NI > 3 ? 3 : 0]),               \
                 "+v"(fb[0][0]), "+v"(fb[0][NJ > 1 ? 1 : 0]),                                                                         \
                 "+v"(fa[1][0]), "+v"(fa[1][NI > 1 ? 1 : 0]), "+v"(fa[1][NI > 2 ? 2 : 0]), "+v"(fa[1][NI > 3 ? 3 : 0]),               \
                 "+v"(fb[1][0]), "+v"(fb[1][NJ > 1 ? 1 : 0])                                                                          \
               :                                                                                                                      \
               : "memory")
    // One k-step: the MFMAs of fragment set `cur` row by row; behind MFMA m goes read m of k-step kk_n (buffer offset off_n) into set
    // `nxt` and — when DMA is true — piece m of tile kt+2.  Before row i a COUNTED wait retires exactly the fragments that row needs
    // (A_i, for i = 0 also every B): of the previous step's NR reads NI-1-i may still be in flight, plus the min(i*NJ, NR) reads this
    // step has issued so far — every fragment gets 6 to 9 MFMA times (190-290 cycles) between its issue and its first use.
#define G_STEP(cur, nxt, off_n, kk_n, DMA)                                                                                            \
  do {                                                                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < NI * NJ; ++m_) {                                                                          \
      if (m_ % NJ == 0)                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(%3)"                                                                                          \
                     : "+v"(fa[cur][m_ / NJ]), "+v"(fb[cur][0]), "+v"(fb[cur][NJ - 1])                                                \
                     : "n"(NI - 1 - m_ / NJ + ((m_ / NJ) * NJ < NR ? (m_ / NJ) * NJ : NR))                                            \
                     : "memory");                                                                                                     \
      acc[m_ / NJ][m_ % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][m_ / NJ]),                   \
                                                                     __builtin_bit_cast(bf16x8, fb[cur][m_ % NJ]), acc[m_ / NJ][m_ % NJ], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
      G_RD1(nxt, off_n, kk_n, m_);                                                                                                    \
      if ((DMA) && m_ < NP && (kt + 2 < nkt || more_tiles)) stage_piece(kt & 1, kt + 2, m_);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
    }                                                                                                                                 \
  } while (0)
    stage_tile(0, 0);
    if (nkt > 1) {
      stage_tile(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");   // tile 0 has landed; tile 1 may still be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) G_RD1(0, 0u, 0, r);
    for (;;) {                                         // PERS: one iteration per tile of this workgroup; else exactly one
      int m0n = 0, n0n = 0;
      f32x4 pbv[NJ];                                   // PERS: the tile's bias values, fetched here so that the epilogue waits for nothing
      if constexpr (PERS) {
        more_tiles = orig + (int)gridDim.x < nt;
        if (more_tiles) tile_origin(orig + (int)gridDim.x, m0n, n0n);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * (NJ * 32) + j * 32 + (lane & 7) * 4;
          pbv[j] = (g.bias && col < g.N) ? *reinterpret_cast<const f32x4*>(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
      for (int kt = 0; kt < nkt; ++kt) {
        const unsigned boff = (kt & 1) * 2 * G_TILE, noff = ((kt + 1) & 1) * 2 * G_TILE;
        __builtin_amdgcn_sched_barrier(0);
        G_STEP(0, 1, boff, 1, false);
        G_STEP(1, 0, boff, 2, false);
        G_STEP(0, 1, boff, 3, false);
        // own DMA of tile kt+1 has landed and own reads of buffer kt&1 are complete; past the barrier that holds for every wave:
        // tile kt+1 may be read, buffer kt&1 may be refilled (the last k-step's MFMAs still run from registers)
        G_RETIRE_ALL("s_waitcnt vmcnt(0)");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PERS) {
          if (more_tiles && kt == nkt - 2) retarget(m0n, n0n);   // from here on the DMA stages the next tile's k-tiles 0 and 1
        }
        G_STEP(1, 0, noff, 0, true);                   // (after the last tile: harmless reads of stale LDS, retired below)
      }
      if constexpr (!PERS) break;
      if constexpr (PERS) {
        // ---- epilogue of this tile (the next tile's operands are landing meanwhile; its k-step-0 fragments are already on their way into
        // fragment set 0): every 32 x 32 accumulator tile through the wave-private patch in two halves of 16 rows, out as 16-byte stores
        constexpr int EP = 40;
        float* patch = reinterpret_cast<float*>(ldsg + G_LDS) + wave * (16 * EP);
        const int prow8 = lane >> 3, pc4 = (lane & 7) * 4;
        float* C = g.C;
        const long long ldc = g.ldc;
        const bool stream_out = (g.nt_store & 1) != 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn * (NJ * 32) + j * 32 + pc4;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
              for (int r = 0; r < 8; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][hh * 8 + r];
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int it = 0; it < 2; ++it) {
                const int rl = it * 8 + prow8;
                const int row = m0 + wm * (NI * 32) + i * 32 + hh * 16 + rl;
                f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * EP + pc4]);
                if (row < g.M && col < g.N) {
                  v += pbv[j];
                  f32x4* pc = reinterpret_cast<f32x4*>(C + (long long)row * ldc + col);
                  if (stream_out) __builtin_nontemporal_store(v, pc);
                  else *pc = v;
                }
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
        if (!more_tiles) break;
        orig += (int)gridDim.x; m0 = m0n; n0 = n0n;
      }
    }
    G_RETIRE_ALL("s_waitcnt");
    if constexpr (PERS) return;
#undef G_RD1
#undef G_ISA
#undef G_IDX
#undef G_RETIRE_ALL
#undef G_STEP
  }

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
  // ---- epilogue, wide form (whenever the C rows are 16-byte addressable): every 32 x 32 accumulator tile goes through a wave-private LDS
  // patch and leaves as 16-byte stores, 8 rows x 128 B per wave-instruction - a quarter of the store instructions of the lane-per-column
  // form below (the address unit spends ~16 cycles on a wave-instruction whatever the width per lane: 256 KB of C per tile take
  // ~16 B/clk with 4-byte lanes).  Patch pitch 40 floats: the two half-waves of a write (rows r, r + 4) land on disjoint bank halves.
  {
    const bool wide = (g.N % 4) == 0 && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && !(g.nt_store & (32 | 64 | 128)) &&
                      (partial || !g.bias || ((uintptr_t)g.bias % 16) == 0);
    if constexpr (NWV == 8) if (wide) {                // (the 16-wave variant sits at its 128-register cap: with this branch compiled in it spills accumulators inside the main loop)
      constexpr int EP = 40;
      __syncthreads();                                  // every wave is done with the operand tiles: the LDS is free
      // everything the epilogue addresses with is derived from an opaque copy of the lane id made HERE: computed from `lane` the compiler
      // hoists it above the main loop, and the 16-wave variant (128 registers) then spills accumulators inside the loop
      int el = lane;
      asm volatile("" : "+v"(el));
      const int l31 = el & 31, half = el >> 5;
      float* patch = reinterpret_cast<float*>(ldsg) + wave * (32 * EP);
      const int prow = el >> 3, pc4 = (el & 7) * 4;
      const bool stream_out = !partial && (g.nt_store & 1);
      f32x4 bv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * (NJ * 32) + j * 32 + pc4;
        bv[j] = (!partial && g.bias && col < g.N) ? *reinterpret_cast<const f32x4*>(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r];
          __builtin_amdgcn_wave_barrier();              // (LDS executes a wave's instructions in order; this only pins the compiler)
          const int col = n0 + wn * (NJ * 32) + j * 32 + pc4;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + prow;
            const int row = m0 + wm * (NI * 32) + i * 32 + rl;
            f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * EP + pc4]);
            if (row < g.M && col < g.N) {
              v += bv[j];
              f32x4* pc = reinterpret_cast<f32x4*>(C + (long long)row * ldc + col);
              if (!partial && g.accumulate) v += *pc;
              if (stream_out) __builtin_nontemporal_store(v, pc);
              else *pc = v;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      return;
    }
  }
  // bias values of this lane's NJ columns, loaded BEFORE the first store: a load between stores would wait (vmcnt is one in-order
  // counter) for every store issued so far — a serialised HBM write round trip per (i, j) tile of the epilogue
  float bvj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn * (NJ * 32) + j * 32 + l31;
    bvj[j] = (!partial && g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * (NJ * 32) + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = bvj[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (NI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && !((g.nt_store & 32) && r != 0)) {      // (bit 5: timing experiment, 1/16 of the stores)
          float v = acc[i][j][r] + bv;
          float* p = C + (long long)row * ldc + col;
          if (g.nt_store & 64) p = C + (long long)((blockIdx.x % 120) * 256 + (row & 255)) * ldc + (col & 255);   // timing experiment: L2-resident target
          if (!partial && g.accumulate) v += *p;
          // final C is streamed out once (788 MB for the forward Gx) -> non-temporal; split-K slabs are re-read right away -> cached
          if (partial || !g.nt_store) *p = v;
          else __builtin_nontemporal_store(v, p);
        }
      }
    }
  }
}

// ---- "TN" form: C[M,N] (fp32) (+)= A[K,M]^T (bf16) * B[K,N] (bf16): BOTH operands row-major with the reduction index on the ROWS -------
// This is the weight-gradient product of the recurrent layers (dW = dGx^T [Xn | h], K = T*B): with it the backward pass needs no
// transposed copy of dGx, d(hn), h or Xn (four HBM passes per layer).  Same 256 x 256 x 64 tile, the same LDS-DMA double buffer and the
// same software pipeline as the NT kernel above; what changes is the LDS image and the fragment read:
//   * an operand tile is 64 k-rows x 256 columns = 64 rows of 512 B; one DMA wave-instruction lands two k-rows (1 KiB);
//   * a 32x32x16 MFMA wants, per lane, 8 consecutive k of ONE column - a strided gather from that image.  ds_read_b64_tr_b16 does the
//     gather in the LDS crossbar: inside a 16-lane group, lane i receives element (i & 3) of the 8-byte datum addressed by lane
//     4j + (i >> 2), j = 0..3 (scripts/probe_tr_read.hip).  With lane p addressing row (p >> 2), columns 4 (p & 3) .. + 3 of a 4 x 16
//     block, lane i ends up with rows 0..3 of column i: two such reads give the 8 k of one MFMA operand;
//   * the four rows of a read sit 512 B apart (same banks): the 16-byte slot s of row r is stored at slot s ^ ((r & 3) << 2) (applied to
//     the DMA source address and to the read address), which spreads a 32-lane read group over all 64 banks exactly once.
// GROUPED (TnSGroup): ONE launch for several products of different shapes sharing the split factor — the weight-gradient products of one
// recurrent layer (dW_ih, the r,z rows and the n rows of dW_hh of both directions): blockIdx.x is a flat work-item index, item ->
// (problem, K slice, tile); every item is one tile over one K slice of ~K / splitk rows, so with 192 tiles x 4 slices = 768 equal items the
// chip runs exactly three full rounds where three separate launches each had their own ramp and tail, and the partial slabs are 4 per tile.
// TERMS: consecutive problems that name the same C are terms of ONE product (the fp32 mode's hi.hi + hi.lo + lo.hi on row-pitched views of
// split operands): each term is an entry of its own, all of them write slabs of the product's slab array (slab0 = term * splitk) and the
// entry of the first term carries the reduce over all nslab = terms * splitk slabs.
struct TnSProb {
  const __bf16* A; const __bf16* B; float* C; float* partial;       // partial: the product's slabs [nslab][M][N] (unused when it has one slab)
  int M, N, K, lda, ldb, ldc, ntx, ntiles, kchunk, first_item;
  int slab0, nslab, to_slab;                                        // this entry's first slab; slabs to reduce (first term only, else 0); write a slab?
  // fused reduce (gemm_tn_w4.h): the product's arrival counters, one per tile (zeroed by the launcher), and how many slabs a tile waits for —
  // the workgroup that stores the LAST slab of a tile adds all of them in slab order and writes C; tick == NULL: the separate reduce launch
  int* tick; int nslab_all;
};
struct TnSGroup {
  TnSProb p[TN_MAX_PROBLEMS];
  int nprob, splitk, nitems, order;
  // epilogue of ONE product in the reduce launch (ds2_gemm_bf16_tn_splitk_group_ep; ep_prob < 0: none):
  //   C[r][c] = (sum of slabs)[r][c] * ep_scale[c] + ep_rowv[r] * ep_shift[c]   — the weight gradient of a projection with a folded BatchNorm
  int ep_prob;
  const float* ep_scale; const float* ep_rowv; const float* ep_shift;
};

template <bool GROUPED>
__global__ __launch_bounds__(512) void gemm_bf16_tn_glds_kernel(BArgs g, int ntx, int nty, TnSGroup grp) {
  constexpr int WN = 4, NWV = 8, NI = 4, NJ = 2, NP = 4;
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const __bf16* A;
  const __bf16* B;
  int pM, pN, pK, plda, pldb, zb, zs, orig, nt, kchunk;
  float* Cfinal; float* Cslab; long long ldcf;
  bool partial;
  if constexpr (GROUPED) {
    // the item's problem: field by field with wave-uniform compares (a dynamically indexed kernel-argument struct would go to scratch)
    // items handed to the workgroups in XCD-sized runs of 32 consecutive (slice, tile) pairs: see gemm_tn_w4.h (same order, same reason)
    int item = blockIdx.x;
    if (grp.order) {
      const int full = grp.nitems & ~255;
      if (item < full) {
        item = (item & ~255) + ((item & 7) << 5) + ((item & 255) >> 3);
      } else {
        const int R = grp.nitems - full, o = item - full, x = o & 7, q8 = R >> 3, r8 = R & 7;
        item = full + (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (o >> 3);
      }
    }
    A = grp.p[0].A; B = grp.p[0].B; Cfinal = grp.p[0].C; Cslab = grp.p[0].partial;
    pM = grp.p[0].M; pN = grp.p[0].N; pK = grp.p[0].K; plda = grp.p[0].lda; pldb = grp.p[0].ldb; ldcf = grp.p[0].ldc; ntx = grp.p[0].ntx;
    nt = grp.p[0].ntiles; kchunk = grp.p[0].kchunk;
    int first = 0, slab0 = grp.p[0].slab0, to_slab = grp.p[0].to_slab;
#pragma unroll
    for (int k = 1; k < TN_MAX_PROBLEMS; ++k)
      if (k < grp.nprob && item >= grp.p[k].first_item) {
        A = grp.p[k].A; B = grp.p[k].B; Cfinal = grp.p[k].C; Cslab = grp.p[k].partial;
        pM = grp.p[k].M; pN = grp.p[k].N; pK = grp.p[k].K; plda = grp.p[k].lda; pldb = grp.p[k].ldb; ldcf = grp.p[k].ldc; ntx = grp.p[k].ntx;
        nt = grp.p[k].ntiles; kchunk = grp.p[k].kchunk; first = grp.p[k].first_item; slab0 = grp.p[k].slab0; to_slab = grp.p[k].to_slab;
      }
    const int local = item - first;
    zb = 0; zs = local / nt; orig = local % nt;
    partial = to_slab != 0;
    Cslab += (long long)(slab0 + zs) * pM * pN;
  } else {
    const int z = blockIdx.z;
    zb = z / g.splitk; zs = z % g.splitk;
    A = g.A + (long long)zb * g.sA;
    B = g.B + (long long)zb * g.sB;
    pM = g.M; pN = g.N; pK = g.K; plda = g.lda; pldb = g.ldb; kchunk = g.kchunk;
    nt = ntx * nty;
    orig = blockIdx.x;
    partial = g.splitk > 1;
    Cfinal = g.C + (long long)zb * g.sC; ldcf = g.ldc;
    Cslab = g.partial + ((long long)zb * g.splitk + zs) * (long long)g.M * g.N;
  }
  int tile = orig;
  if (!(GROUPED && grp.order)) {
    const int xcd = orig & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int m0 = (tile / ntx) * 256, n0 = (tile % ntx) * 256;
  const int kbeg = zs * kchunk;
  const int kend = min(pK, kbeg + kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  // staging: piece p = wave + 8 i holds k-rows 2p, 2p + 1; lane -> (k-row, physical 16-byte slot); columns past M / N are clamped
  // to the last valid segment (they only feed C rows / columns that are never stored), k-rows past kend read the zero page
  const char* zp = reinterpret_cast<const char*>(g_zero16);
  const char* qA[NP];
  const char* qB[NP];
  int rowk[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = (wave + NWV * i) * 2 + (lane >> 5);
    const int gs = (lane & 31) ^ ((r & 3) << 2);
    rowk[i] = r;
    qA[i] = reinterpret_cast<const char*>(A + (long long)(kbeg + r) * plda + min(m0 + gs * 8, pM - 8));
    qB[i] = reinterpret_cast<const char*>(B + (long long)(kbeg + r) * pldb + min(n0 + gs * 8, pN - 8));
  }
  const long long stepA = (long long)plda * (BK * 2), stepB = (long long)pldb * (BK * 2);
  const int nfull = (kend - kbeg) / BK;
  auto stage_piece = [&](int buf, int kt, int i) {
    char* dA = ldsg + buf * 2 * G_TILE + (wave + NWV * i) * 1024;
    if (kt < nfull) {
      glds16(qA[i], dA);
      glds16(qB[i], dA + G_TILE);
      qA[i] += stepA;
      qB[i] += stepB;
    } else {                                          // the K tail (the pointers already stand on this tile)
      const bool kok = kbeg + kt * BK + rowk[i] < kend;
      glds16(kok ? qA[i] : zp, dA);
      glds16(kok ? qB[i] : zp, dA + G_TILE);
    }
  };
  auto stage_tile = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) stage_piece(buf, kt, i);
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses (buffer 0, k-step 0, first of the two reads): lane p of a 16-lane group addresses row (p >> 2) of the
  // 4-row block, columns 4 (p & 3) .. + 3 of the group's 16 columns; k-step kk and the second read are immediate offsets
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;
  const int p16 = lane & 15, q4 = p16 >> 2, g16 = (lane >> 4) & 1;
  const unsigned rowpart = (unsigned)((half * 8 + q4) * 512);
  unsigned va[NI], vb[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = wm * 128 + i * 32 + g16 * 16 + 4 * (p16 & 3);
    va[i] = lds0 + rowpart + ((((col >> 3) ^ (q4 << 2)) << 4) | (((col >> 2) & 1) << 3));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = wn * 64 + j * 32 + g16 * 16 + 4 * (p16 & 3);
    vb[j] = lds0 + G_TILE + rowpart + ((((col >> 3) ^ (q4 << 2)) << 4) | (((col >> 2) & 1) << 3));
  }
  // A fragment is two reads, each landing in one half of the 4-register MFMA operand.  The halves are joined (a register-tuple
  // definition, no instruction) right behind the reads and the waits name the whole operand, as in the NT kernel; the build checks the
  // generated code for this (scripts/check_tn_isa.py): a register copy between a read and its wait would move data that has not landed.
  f32x4 fa[2][NI], fb[2][NJ];
  constexpr int NR = NI + NJ;
#define T_ISA(r_) ((r_) == 0 || (r_) > NJ)
#define T_IDX(r_) ((r_) == 0 ? 0 : (r_) <= NJ ? (r_) - 1 : (r_) - NJ)
#define T_RD2(dst, addr, off)                                                                                                         \
  do {                                                                                                                                \
    f32x2 lo_, hi_;                                                                                                                   \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                        \
                 : "=&v"(lo_), "=&v"(hi_)                                                                                             \
                 : "v"(addr), "n"(off), "n"((off) + 2048));                                                                           \
    dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);                                                                              \
  } while (0)
#define T_RD1(set, bufoff, kk, r_)                                                                                                    \
  do {                                                                                                                                \
    if ((r_) < NR && T_ISA(r_))                                                                                                       \
      T_RD2(fa[set][(r_) < NR && T_ISA(r_) ? T_IDX(r_) : 0], va[(r_) < NR && T_ISA(r_) ? T_IDX(r_) : 0] + (bufoff), (kk) * 8192);     \
    else if ((r_) < NR)                                                                                                               \
      T_RD2(fb[set][(r_) < NR && !T_ISA(r_) ? T_IDX(r_) : 0], vb[(r_) < NR && !T_ISA(r_) ? T_IDX(r_) : 0] + (bufoff), (kk) * 8192);   \
  } while (0)
#define T_RETIRE_ALL(WAITSTR)                                                                                                         \
  asm volatile(WAITSTR " lgkmcnt(0)"                                                                                                  \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]),                      \
                 "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1])                       \
               :                                                                                                                      \
               : "memory")
  // the counted waits of the NT pipeline with every count doubled (a fragment is two reads here); the largest is 12 <= 15
#define T_STEP(cur, nxt, off_n, kk_n, DMA)                                                                                            \
  do {                                                                                                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < NI * NJ; ++m_) {                                                                          \
      if (m_ % NJ == 0)                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(%3)"                                                                                          \
                     : "+v"(fa[cur][m_ / NJ]), "+v"(fb[cur][0]), "+v"(fb[cur][NJ - 1])                                                \
                     : "n"(2 * (NI - 1 - m_ / NJ + ((m_ / NJ) * NJ < NR ? (m_ / NJ) * NJ : NR)))                                      \
                     : "memory");                                                                                                     \
      acc[m_ / NJ][m_ % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][m_ / NJ]),                   \
                                                                     __builtin_bit_cast(bf16x8, fb[cur][m_ % NJ]), acc[m_ / NJ][m_ % NJ], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
      T_RD1(nxt, off_n, kk_n, m_);                                                                                                    \
      if ((DMA) && m_ < NP && kt + 2 < nkt) stage_piece(kt & 1, kt + 2, m_);                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
    }                                                                                                                                 \
  } while (0)
  stage_tile(0, 0);
  if (nkt > 1) {
    stage_tile(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) T_RD1(0, 0u, 0, r);
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned boff = (kt & 1) * 2 * G_TILE, noff = ((kt + 1) & 1) * 2 * G_TILE;
    __builtin_amdgcn_sched_barrier(0);
    T_STEP(0, 1, boff, 1, false);
    T_STEP(1, 0, boff, 2, false);
    T_STEP(0, 1, boff, 3, false);
    T_RETIRE_ALL("s_waitcnt vmcnt(0)");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    T_STEP(1, 0, noff, 0, true);
  }
  T_RETIRE_ALL("s_waitcnt");
#undef T_RD1
#undef T_RD2
#undef T_ISA
#undef T_IDX
#undef T_RETIRE_ALL
#undef T_STEP

  float* C = partial ? Cslab : Cfinal;
  const long long ldc = partial ? (long long)pN : ldcf;
  const bool accumulate = !partial && !GROUPED && g.accumulate;
  // wide epilogue (see the NT kernel): 32 x 32 accumulator tiles through a wave-private LDS patch, 16-byte stores
  if ((ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && !(!GROUPED && (g.nt_store & 128))) {
    constexpr int EP = 40;
    __syncthreads();
    float* patch = reinterpret_cast<float*>(ldsg) + wave * (32 * EP);
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        const int col = n0 + wn * (NJ * 32) + j * 32 + pc4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = it * 8 + prow;
          const int row = m0 + wm * (NI * 32) + i * 32 + rl;
          f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * EP + pc4]);
          if (row < pM && col < pN) {                   // (N % 8 == 0: a 4-column run never straddles the edge)
            f32x4* pc = reinterpret_cast<f32x4*>(C + (long long)row * ldc + col);
            if (accumulate) v += *pc;
            *pc = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * (NJ * 32) + j * 32 + l31;
      if (col >= pN) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (NI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < pM) {
          float v = acc[i][j][r];
          float* p = C + (long long)row * ldc + col;
          if (accumulate) v += *p;
          *p = v;
        }
      }
    }
  }
}

__global__ void splitk_reduce_bf_kernel(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                        int ldc, long long sC, int splitk, int accumulate) {
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

// the slabs of every product of a grouped launch -> C (one launch; fixed summation order: bit-identical from run to run).  Entries with
// nslab == 0 (further terms of a product, or a product written directly) own no elements here.
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(TnSGroup grp, long long total) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    long long local = idx;
    const float* part = nullptr; float* C = nullptr; int M = 0, N = 1, ldc = 0, nslab = 0;
    bool found = false, ep = false;
#pragma unroll
    for (int q = 0; q < TN_MAX_PROBLEMS; ++q) {
      const long long mn = (q < grp.nprob && grp.p[q].nslab > 0) ? (long long)grp.p[q].M * grp.p[q].N / 4 : 0;
      if (!found && local < mn) {
        part = grp.p[q].partial; C = grp.p[q].C; M = grp.p[q].M; N = grp.p[q].N; ldc = grp.p[q].ldc; nslab = grp.p[q].nslab;
        found = true; ep = q == grp.ep_prob;
      }
      if (!found) local -= mn;
    }
    const long long e = local * 4;                      // N % 8 == 0: a float4 never straddles a row
    const int row = (int)(e / N), col = (int)(e % N);
    f32x4 sum = *reinterpret_cast<const f32x4*>(part + e);
    for (int sidx = 1; sidx < nslab; ++sidx) sum += *reinterpret_cast<const f32x4*>(part + (long long)sidx * M * N + e);
    if (ep) sum = sum * *reinterpret_cast<const f32x4*>(grp.ep_scale + col) + *reinterpret_cast<const f32x4*>(grp.ep_shift + col) * grp.ep_rowv[row];
    *reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) = sum;
  }
}

// dst[r][c] = bf16(src[r][c]), c < C ; dst[r][c] = 0 for C <= c < ldd      (ldd % 8 == 0)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int lds_, __bf16* __restrict__ dst, int ldd, int R,
                                                        int Cc, int vec) {
  const int cq = ldd / 4;
  const long long total = (long long)R * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = i / cq, c0 = (i % cq) * 4;
    const float* s = src + (long long)r * lds_ + c0;
    float v[4];
    if (c0 + 4 <= Cc && vec) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(s);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (c0 + j < Cc) ? s[j] : 0.f;
    }
    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(dst + (long long)r * ldd + c0) = o;
  }
}

// dstT[c][r] = bf16(src[r][c]) ; dstT row pitch ldt >= R (ldt % 8 == 0), pad columns R..ldt zero-filled.  Optionally also
// dstR[r][c] = bf16(src[r][c]) (row pitch ldr % 8 == 0, pad columns C..ldr zero) from the SAME read of src: the backward pass
// needs dGx both ways (dX = dGx W, dW = dGx^T X) and the fp32 source is the big operand of both passes.
// 64x64 tiles through LDS: fp32 reads are float4 (256-B runs along c), transposed writes are 16-byte runs along r.
__global__ __launch_bounds__(256) void cast_transpose_bf16_kernel(const float* __restrict__ src, int lds_, __bf16* __restrict__ dstT, int ldt,
                                                                  __bf16* __restrict__ dstR, int ldr, int R, int Cc, int vec,
                                                                  float* __restrict__ colpart) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int rl = pass * 16 + (tid >> 4), cl = (tid & 15) * 4;
    const int r = r0 + rl, c = c0 + cl;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const float* sp = src + (long long)r * lds_ + c;
      if (vec && c + 4 <= Cc) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(sp);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (c + j < Cc) ? sp[j] : 0.f;
      }
      if (dstR && c < ldr) *reinterpret_cast<bf16x4*>(dstR + (long long)r * ldr + c) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rl][cl + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = pass * 32 + (tid >> 3), rl = (tid & 7) * 8;
    const int c = c0 + cl, r = r0 + rl;
    if (c < Cc && r < ldt) {                      // rows >= R were staged as zeros; ldt % 8 == 0 keeps the 8-run inside the pitch
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)tile[rl + j][cl];
      *reinterpret_cast<bf16x8*>(dstT + (long long)c * ldt + r) = o;
    }
  }
  if (colpart && tid < 64 && c0 + tid < Cc && r0 < R) {     // column sums of this 64-row tile: partial [row tile][C][2] (second slot unused)
    float sum = 0.f;
#pragma unroll 16
    for (int r = 0; r < 64; ++r) sum += tile[r][tid];
    float* o = colpart + ((long long)blockIdx.y * Cc + c0 + tid) * 2;
    o[0] = sum;
    o[1] = 0.f;
  }
}

// bf16 (R, C) -> bf16 (C, ldt) transposed (+ optional fp32 column sums): the bf16-mode backward writes dGx in bf16 straight from
// the recurrent epilogue, so the only preparation left for the weight-gradient GEMMs is this half-traffic transpose.
// 64x64 tiles through LDS; 16-byte loads along c, 16-byte stores along r.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const __bf16* __restrict__ src, int lds_, __bf16* __restrict__ dstT, int ldt, int R, int Cc,
                                                             float* __restrict__ colpart) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = pass * 32 + (tid >> 3), cl = (tid & 7) * 8;
    const int r = r0 + rl, c = c0 + cl;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const __bf16* sp = src + (long long)r * lds_ + c;
      if (c + 8 <= Cc) {
        const bf16x8 q = *reinterpret_cast<const bf16x8*>(sp);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c + j < Cc) ? (float)sp[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[rl][cl + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = pass * 32 + (tid >> 3), rl = (tid & 7) * 8;
    const int c = c0 + cl, r = r0 + rl;
    if (c < Cc && r < ldt) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)tile[rl + j][cl];
      *reinterpret_cast<bf16x8*>(dstT + (long long)c * ldt + r) = o;
    }
  }
  if (colpart && tid < 64 && c0 + tid < Cc && r0 < R) {
    float sum = 0.f;
#pragma unroll 16
    for (int r = 0; r < 64; ++r) sum += tile[r][tid];
    float* o = colpart + ((long long)blockIdx.y * Cc + c0 + tid) * 2;
    o[0] = sum;
    o[1] = 0.f;
  }
}

// fp32 (R x C, pitch lds_) -> SPLIT bf16 operand for the fp32 mode's three-term products: x = hi + lo + O(2^-18 x) with hi = bf16(x),
// lo = bf16(x - hi), and  a.b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi  (the dropped lo.lo term is 2^-18 of the product): three bf16 MFMA
// products with fp32 accumulation instead of one fp32-input MFMA product at a sixteenth of the rate.  The three terms are ONE GEMM over a
// reduction index three times as long: the A operand is written as [hi | hi | lo] per row and the B operand as [hi | lo | hi]
// (order 0 / 1), each block Cp = pad8(C) columns wide with zero padding; order 2 = [hi | lo] (operands of TN products, whose terms are
// accumulated launch by launch on row-pitched views of the blocks).
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, int lds_, __bf16* __restrict__ dst, int ldd, int R, int Cc,
                                                         int Cp, int order, int vec) {
  const int cq = Cp / 4;
  const long long total = (long long)R * cq;
  const int b_hi2 = order == 0 ? Cp : (order == 1 ? 2 * Cp : -1);       // second copy of hi (none for order 2)
  const int b_lo = order == 0 ? 2 * Cp : Cp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = i / cq, c0 = (i % cq) * 4;
    const float* sp = src + (long long)r * lds_ + c0;
    float v[4];
    if (c0 + 4 <= Cc && vec) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(sp);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (c0 + j < Cc) ? sp[j] : 0.f;
    }
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (__bf16)v[j];
      lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
    __bf16* d = dst + (long long)r * ldd + c0;
    *reinterpret_cast<bf16x4*>(d) = hi;
    if (b_hi2 >= 0) *reinterpret_cast<bf16x4*>(d + b_hi2) = hi;
    *reinterpret_cast<bf16x4*>(d + b_lo) = lo;
  }
}

#include "gemm_nt_ring.h"
#include "permlane.h"
#include "gemm_nt_w4.h"
#include "gemm_tn_group.h"
#include "gemm_tn_w4.h"

}  // namespace

// CU count of the CURRENT device, and "set this function's dynamic-LDS attribute once per DEVICE": per device id, so that a process that
// drives two GPUs (device_test != device, two threads on two devices) never launches with the other device's grid size or without the
// attribute.  The four-wave kernels (gemm_nt_w4.h / gemm_tn_w4.h: v_mfma_f32_16x16x32_bf16, 128 KB of LDS, one in-order
// vmcnt shared by loads and stores) are gfx950 code: ds2_is_gfx950() gates them in the launchers.
static int ds2_dev_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return dev;
}
static int ds2_cus_current() {
  static int cus[64] = {0};
  const int dev = ds2_dev_index();
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
  }
  return cus[dev];
}
static bool ds2_is_gfx950() {
  static int known[64] = {0};                            // 0 unknown, 1 yes, 2 no
  const int dev = ds2_dev_index();
  if (!known[dev]) {
    hipDeviceProp_t prop;
    known[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 2;
  }
  return known[dev] == 1;
}
#define DS2_ATTR_ONCE(func, bytes)                                                                                          \
  do {                                                                                                                      \
    static bool done_[64] = {false};                                                                                        \
    const int d_ = ds2_dev_index();                                                                                         \
    if (!done_[d_]) { DS2_HIP(hipFuncSetAttribute((const void*)func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); done_[d_] = true; } \
  } while (0)

extern "C" size_t ds2_gemm_bf16_workspace_bytes(int M, int N, int batch, int splitk) {
  if (splitk <= 1) return 0;
  return (size_t)batch * splitk * (size_t)M * N * sizeof(float);
}

// C[M,N] fp32 (+)= A[M,K] bf16 (pitch lda) * B[N,K]^T bf16 (pitch ldb) (+ bias).  K, lda, ldb multiples of 8; 16-byte aligned bases.
extern "C" int ds2_gemm_bf16_nt(int M, int N, int K, const void* A, int lda, long long strideA, const void* B, int ldb, long long strideB,
                                float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch, int splitk,
                                void* workspace, size_t workspace_bytes, void* stream) {
  DS2_REQUIRE(M > 0 && N > 0 && K > 0 && batch >= 1, "ds2_gemm_bf16_nt: bad dims M=%d N=%d K=%d", M, N, K);
  DS2_REQUIRE(A && B && C, "ds2_gemm_bf16_nt: null pointer");
  DS2_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (strideA % 8) == 0 && (strideB % 8) == 0,
              "ds2_gemm_bf16_nt: K, lda, ldb, strides must be multiples of 8 (K=%d lda=%d ldb=%d)", K, lda, ldb);
  DS2_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "ds2_gemm_bf16_nt: operands must be 16-byte aligned");
  if (splitk < 1) splitk = 1;
  int kchunk = ceil_div(ceil_div(K, splitk), BK) * BK;
  splitk = ceil_div(K, kchunk);
  if (splitk > 1)
    DS2_REQUIRE(workspace && workspace_bytes >= ds2_gemm_bf16_workspace_bytes(M, N, batch, splitk), "ds2_gemm_bf16_nt: workspace too small");
  BArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.splitk = splitk; g.kchunk = kchunk; g.accumulate = accumulate; g.partial = (float*)workspace;
  static const char* nt_env = ds2_exp_getenv("DS2_GEMM_NT");       // tuning override, default on
  g.nt_store = nt_env ? (nt_env[0] != '0') : 1;
  // timing experiments only (WRONG RESULTS; scripts/ab_gemm_dbg.sh): 1 = no operand DMA after the first k-tile, 16 = 1/16 of the C
  // stores, 32 = C stores aimed at an L2-resident region
  static const int dbg_bits = ds2_exp_getenv("DS2_GEMM_DBG") ? atoi(ds2_exp_getenv("DS2_GEMM_DBG")) << 1 : 0;
  g.nt_store |= dbg_bits;
  static const char* wide_env = ds2_exp_getenv("DS2_GEMM_WIDE");  // "0": lane-per-column epilogue stores (A/B switch)
  if (wide_env && wide_env[0] == '0') g.nt_store |= 128;
  static const int super_rows = ds2_exp_getenv("DS2_GEMM_SR") ? atoi(ds2_exp_getenv("DS2_GEMM_SR")) : 4;   // row tiles per super-row of the tile walk (0: row-major)
  g.super_rows = batch == 1 ? super_rows : 0;
  hipStream_t s = (hipStream_t)stream;
  // 256 x 256 LDS-DMA kernel whenever its tiles cover at least half the chip; the 128 x 128 kernel for everything smaller
  const long long tiles256 = (long long)ceil_div(N, 256) * ceil_div(M, 256) * batch * splitk;
  static const char* force = ds2_exp_getenv("DS2_GEMM_TILE");      // "128" | "glds": tuning override (scripts/bench_gemm.py)
  const bool use_glds = force ? (force[0] == 'g') : (M >= 256 && N >= 256 && tiles256 >= 128);
  if (use_glds) {
    DS2_ATTR_ONCE((gemm_bf16_nt_glds_kernel<2, 4>), G_LDS);
    const int ntx = ceil_div(N, 256), nty = ceil_div(M, 256);
    // 8 waves (128 x 64 per wave) for every shape (a 16-wave 64 x 64 variant lost to it in round 2 once the epilogue went through LDS and is no
    // longer built).
    static const char* wv = ds2_exp_getenv("DS2_GEMM_WAVES");      // "pp": the ping-pong schedule of the one-tile kernel (tuning override)
    const bool pp = wv && wv[0] == 'p';
    if (pp) {
      DS2_ATTR_ONCE((gemm_bf16_nt_glds_kernel<2, 4, true>), G_LDS);
      hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<2, 4, true>), dim3(ntx * nty, 1, batch * splitk), dim3(512), G_LDS, s, g, ntx, nty);
    } else {
      // persistent form (one workgroup per CU walking several tiles; see the kernel): short reductions with more tiles than CUs, plain write-out
      static const char* pe = ds2_exp_getenv("DS2_GEMM_PERS");     // "0": one workgroup per tile for every shape (A/B switch)
      const int cus = ds2_cus_current();
      const int nkt = K / BK;
      const bool wide = (N % 4) == 0 && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0);
      const bool pers = !(pe && pe[0] == '0') && batch == 1 && splitk == 1 && !accumulate && (K % BK) == 0 && nkt >= 2 && (nkt % 2) == 0 &&
                        nkt <= 128 && wide && !(g.nt_store & ~1) && ntx * nty > cus;
      // EXPERIMENTS of round 5 (gemm_nt_ring.h; profiles/r05_gemm_ring_ab.txt), bit-identical to the production kernel and not faster:
      // DS2_GEMM_RING=1 the software-pipelined loop over a four-slice LDS ring with counted DMA waits, =q the ping-pong form on
      // v_mfma_f32_16x16x32_bf16 with a register-direct epilogue.  Both are bound, like the production kernel, by what a CU can take in
      // through its vector-memory path when part of the operand stream misses the L2 (the operand DMA alone, no MFMA: 350 us on the dX shape).
      static const char* ring_env = ds2_exp_getenv("DS2_GEMM_RING");
      const bool ring = ring_env && (ring_env[0] == '1' || ring_env[0] == 'q') && !(pe && pe[0] == '0') && batch == 1 && splitk == 1 && !accumulate &&
                        (K % 32) == 0 && K >= 128 && wide && !(g.nt_store & ~1) && ntx * nty > cus;
      // four waves x 128 x 128 (gemm_nt_w4.h): the default wherever it applies — K a multiple of the 64-deep k-tile, more tiles than CUs, plain
      // write-out, 32-bit lane offsets inside a tile (DS2_GEMM_W4=0: the 8-wave persistent kernel instead, A/B switch)
      static const char* w4_env = ds2_exp_getenv("DS2_GEMM_W4");
      const bool w4 = ds2_is_gfx950() && !(w4_env && w4_env[0] == '0') && !(ring_env && ring_env[0] != 'w') && !(pe && pe[0] == '0') && batch == 1 && splitk == 1 &&
                      !accumulate && (K % 64) == 0 && K >= 128 && wide && !(g.nt_store & ~1) && ntx * nty > cus &&
                      (long long)lda * 512 < (1ll << 31) && (long long)ldb * 512 < (1ll << 31);
      if (w4) {
        static const int wdbg = ds2_exp_getenv("DS2_W4_DBG") ? atoi(ds2_exp_getenv("DS2_W4_DBG")) : 0;   // timing ablations (WRONG RESULTS; scripts/r5_w4_dbg.sh)
#define DS2_W_V(n) { DS2_ATTR_ONCE((gemm_bf16_nt_w4_kernel<n>), W4_LDS); \
                     hipLaunchKernelGGL((gemm_bf16_nt_w4_kernel<n>), dim3(cus, 1, 1), dim3(256), W4_LDS, s, g, ntx, nty); }
        if (wdbg == 1) DS2_W_V(1) else if (wdbg == 2) DS2_W_V(2) else if (wdbg == 4) DS2_W_V(4) else if (wdbg == 5) DS2_W_V(5) else if (wdbg == 6) DS2_W_V(6) else DS2_W_V(0)
#undef DS2_W_V
      } else if (ring) {
        DS2_ATTR_ONCE((gemm_bf16_nt_ring_kernel<false>), R_RING + G_PATCH);
        DS2_ATTR_ONCE((gemm_bf16_nt_pp16_kernel<false>), R_RING);
        static const int rdbg = ds2_exp_getenv("DS2_RING_DBG") ? atoi(ds2_exp_getenv("DS2_RING_DBG")) : 0;   // timing ablations of the pp16 kernel (WRONG RESULTS)
        if (ring_env[0] == 'q' && rdbg) {
#define DS2_Q_DBG(n) if (rdbg == n) { DS2_HIP(hipFuncSetAttribute((const void*)gemm_bf16_nt_pp16_kernel<false, n>, hipFuncAttributeMaxDynamicSharedMemorySize, R_RING)); \
                       hipLaunchKernelGGL((gemm_bf16_nt_pp16_kernel<false, n>), dim3(cus, 1, 1), dim3(512), R_RING, s, g, ntx, nty); }
          DS2_Q_DBG(1) DS2_Q_DBG(2) DS2_Q_DBG(3) DS2_Q_DBG(8) DS2_Q_DBG(18) DS2_Q_DBG(22) DS2_Q_DBG(26) DS2_Q_DBG(82) DS2_Q_DBG(146)
#undef DS2_Q_DBG
        } else if (ring_env[0] == 'q' && ring_env[1] == '5') {       // five slots: the whole LDS
          DS2_ATTR_ONCE((gemm_bf16_nt_pp16_kernel<false, 0, 5>), 5 * R_SLICE);
          hipLaunchKernelGGL((gemm_bf16_nt_pp16_kernel<false, 0, 5>), dim3(cus, 1, 1), dim3(512), 5 * R_SLICE, s, g, ntx, nty);
        } else if (ring_env[0] == 'q')
          hipLaunchKernelGGL((gemm_bf16_nt_pp16_kernel<false>), dim3(cus, 1, 1), dim3(512), R_RING, s, g, ntx, nty);
        else
          hipLaunchKernelGGL((gemm_bf16_nt_ring_kernel<false>), dim3(cus, 1, 1), dim3(512), R_RING + G_PATCH, s, g, ntx, nty);
      } else if (pers) {
        DS2_ATTR_ONCE((gemm_bf16_nt_glds_kernel<2, 4, false, true>), G_LDS + G_PATCH);
        hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<2, 4, false, true>), dim3(cus, 1, 1), dim3(512), G_LDS + G_PATCH, s, g, ntx, nty);
      } else {
        hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<2, 4>), dim3(ntx * nty, 1, batch * splitk), dim3(512), G_LDS, s, g, ntx, nty);
      }
    }
    DS2_LAUNCH_CHECK("gemm_bf16_nt_glds_kernel");
  } else {
    dim3 grid(ceil_div(N, BN), ceil_div(M, BM), batch * splitk);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, grid, dim3(256), 0, s, g);
    DS2_LAUNCH_CHECK("gemm_bf16_nt_kernel");
  }
  if (splitk > 1) {
    hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3(ceil_div(M * N, 256), batch), dim3(256), 0, s, (const float*)workspace, C, bias, M, N,
                       ldc, strideC, splitk, accumulate);
    DS2_LAUNCH_CHECK("splitk_reduce_bf_kernel");
  }
  return 0;
}

// C[M,N] **bf16** = A[M,K] bf16 * B[N,K]^T bf16 + bias (fp32 accumulation and bias add, one rounding at the store): the x-projections of a
// recurrent layer in the bf16 training mode (aten::addmm inside aten::gru / lstm, blocks.py:76-78, 88), consumed once by ds2_rnn_fwd_x.
// Four-wave 256 x 256 x 64 kernel only (gemm_nt_w4.h, OBF): returns 1 — nothing launched, call ds2_gemm_bf16_nt — where that kernel does not
// apply (K % 64, fewer tiles than CUs, N % 8, alignment), 0 when launched, < 0 on error.
extern "C" int ds2_gemm_bf16_nt_obf16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, void* stream) {
  DS2_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "ds2_gemm_bf16_nt_obf16: bad arguments M=%d N=%d K=%d", M, N, K);
  DS2_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
              "ds2_gemm_bf16_nt_obf16: K, lda, ldb must be multiples of 8 and the operands 16-byte aligned");
  const int cus = ds2_cus_current();
  const int ntx = ceil_div(N, 256), nty = ceil_div(M, 256);
  const bool ok = ds2_is_gfx950() && (K % 64) == 0 && K >= 128 && (N % 8) == 0 && (ldc % 8) == 0 && ((uintptr_t)C % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                  ntx * nty > cus && (long long)lda * 512 < (1ll << 31) && (long long)ldb * 512 < (1ll << 31);
  if (!ok) return 1;
  BArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.C = (float*)C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = g.sB = g.sC = 0;
  g.splitk = 1; g.kchunk = K; g.accumulate = 0; g.partial = nullptr;
  g.nt_store = 1; g.super_rows = 4;
  DS2_ATTR_ONCE((gemm_bf16_nt_w4_kernel<0, true>), W4_LDS);
  hipLaunchKernelGGL((gemm_bf16_nt_w4_kernel<0, true>), dim3(cus, 1, 1), dim3(256), W4_LDS, (hipStream_t)stream, g, ntx, nty);
  DS2_LAUNCH_CHECK("gemm_bf16_nt_w4_kernel<obf16>");
  return 0;
}

// C[M,N] fp32 (+)= A[K,M]^T B[K,N]: A (K rows, pitch lda) and B (K rows, pitch ldb) bf16 row-major.  M, N, lda, ldb and the batch
// strides multiples of 8, 16-byte aligned bases.  batch > 1: independent products at the given element strides (strides may be negative).
extern "C" int ds2_gemm_bf16_tn(int M, int N, int K, const void* A, int lda, long long strideA, const void* B, int ldb, long long strideB,
                                float* C, int ldc, long long strideC, int accumulate, int batch, int splitk, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DS2_REQUIRE(M >= 8 && N >= 8 && K > 0 && batch >= 1, "ds2_gemm_bf16_tn: bad dims M=%d N=%d K=%d", M, N, K);
  DS2_REQUIRE(A && B && C, "ds2_gemm_bf16_tn: null pointer");
  DS2_REQUIRE((M % 8) == 0 && (N % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (strideA % 8) == 0 && (strideB % 8) == 0,
              "ds2_gemm_bf16_tn: M, N, lda, ldb, strides must be multiples of 8 (M=%d N=%d lda=%d ldb=%d)", M, N, lda, ldb);
  DS2_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "ds2_gemm_bf16_tn: operands must be 16-byte aligned");
  if (splitk < 1) splitk = 1;
  int kchunk = ceil_div(ceil_div(K, splitk), BK) * BK;
  splitk = ceil_div(K, kchunk);
  if (splitk > 1)
    DS2_REQUIRE(workspace && workspace_bytes >= ds2_gemm_bf16_workspace_bytes(M, N, batch, splitk), "ds2_gemm_bf16_tn: workspace too small");
  BArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.C = C; g.bias = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.splitk = splitk; g.kchunk = kchunk; g.accumulate = accumulate; g.partial = (float*)workspace; g.nt_store = 0; g.super_rows = 0;
  { static const char* wide_env = ds2_exp_getenv("DS2_GEMM_WIDE"); if (wide_env && wide_env[0] == '0') g.nt_store |= 128; }
  hipStream_t s = (hipStream_t)stream;
  DS2_ATTR_ONCE((gemm_bf16_tn_glds_kernel<false>), G_LDS);
  const int ntx = ceil_div(N, 256), nty = ceil_div(M, 256);
  hipLaunchKernelGGL(gemm_bf16_tn_glds_kernel<false>, dim3(ntx * nty, 1, batch * splitk), dim3(512), G_LDS, s, g, ntx, nty, TnSGroup{});
  DS2_LAUNCH_CHECK("gemm_bf16_tn_glds_kernel");
  if (splitk > 1) {
    hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3(ceil_div(M * N, 256), batch), dim3(256), 0, s, (const float*)workspace, C,
                       (const float*)nullptr, M, N, ldc, strideC, splitk, accumulate);
    DS2_LAUNCH_CHECK("splitk_reduce_bf_kernel");
  }
  return 0;
}

// Several TN products in ONE launch of the 256 x 256 kernel with a common split-K factor (gemm_bf16_tn_glds_kernel<true>) + ONE reduce launch.
extern "C" size_t ds2_gemm_bf16_tn_splitk_group_workspace_bytes(int nprob, const ds2_tn_problem* probs, int splitk) {
  if (!probs) return 0;
  if (splitk < 1) splitk = 1;
  size_t n = 0;
  for (int i = 0; i < nprob; ++i) n += (size_t)splitk * probs[i].M * probs[i].N * sizeof(float);   // (an upper bound: one-slab products use none)
  for (int i = 0; i < nprob; ++i) n += (size_t)ceil_div(probs[i].M, 256) * ceil_div(probs[i].N, 256) * sizeof(int);   // arrival counters of the fused reduce
  return n + 256;
}

static int tn_splitk_group_impl(int nprob, const ds2_tn_problem* probs, int splitk, int ep_index, const float* ep_scale, const float* ep_rowv,
                                const float* ep_shift, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int ds2_gemm_bf16_tn_splitk_group(int nprob, const ds2_tn_problem* probs, int splitk, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  return tn_splitk_group_impl(nprob, probs, splitk, -1, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}
// ... with an epilogue on product `ep_index`, applied by the reduce launch:  C = (A^T B) diag(scale) + rowv (x) shift  (scale / shift: N floats,
// 16-byte aligned; rowv: M floats) — the weight gradient of a projection whose BatchNorm was folded into it (ds2_wih_fold_bf16).  Returns 1,
// nothing launched, when that product would have a single slab (split factor 1: no reduce launch to carry the epilogue): call the plain entry
// and ds2_scale_rank1_f32.
extern "C" int ds2_gemm_bf16_tn_splitk_group_ep(int nprob, const ds2_tn_problem* probs, int splitk, int ep_index, const float* ep_scale,
                                                const float* ep_rowv, const float* ep_shift, void* workspace, size_t workspace_bytes, void* stream) {
  DS2_REQUIRE(ep_index >= 0 && ep_index < nprob && ep_scale && ep_rowv && ep_shift && ((uintptr_t)ep_scale % 16) == 0 && ((uintptr_t)ep_shift % 16) == 0,
              "ds2_gemm_bf16_tn_splitk_group_ep: bad epilogue arguments");
  return tn_splitk_group_impl(nprob, probs, splitk, ep_index, ep_scale, ep_rowv, ep_shift, workspace, workspace_bytes, stream);
}
static int tn_splitk_group_impl(int nprob, const ds2_tn_problem* probs, int splitk, int ep_index, const float* ep_scale, const float* ep_rowv,
                                const float* ep_shift, void* workspace, size_t workspace_bytes, void* stream) {
  DS2_REQUIRE(nprob >= 1 && nprob <= TN_MAX_PROBLEMS && probs, "ds2_gemm_bf16_tn_splitk_group: 1..%d problems", TN_MAX_PROBLEMS);
  if (splitk < 1) splitk = 1;
  int kmin = probs[0].K;
  for (int i = 1; i < nprob; ++i) kmin = probs[i].K < kmin ? probs[i].K : kmin;
  while (splitk > 1 && ceil_div(kmin, ceil_div(ceil_div(kmin, splitk), BK) * BK) < splitk) --splitk;   // every slice of every problem must hold rows
  TnSGroup g;
  int items = 0;
  size_t off = 0;
  long long elems = 0;
  bool any_slab = false;
  for (int i = 0; i < nprob; ++i) {
    const ds2_tn_problem& q = probs[i];
    DS2_REQUIRE(q.A && q.B && q.C && q.M >= 8 && q.N >= 8 && q.K > 0, "ds2_gemm_bf16_tn_splitk_group: problem %d: bad dims M=%d N=%d K=%d", i, q.M, q.N, q.K);
    DS2_REQUIRE((q.M % 8) == 0 && (q.N % 8) == 0 && (q.lda % 8) == 0 && (q.ldb % 8) == 0 && (q.ldc % 4) == 0 && ((uintptr_t)q.C % 16) == 0,
                "ds2_gemm_bf16_tn_splitk_group: problem %d: M, N, lda, ldb must be multiples of 8, ldc of 4, C 16-byte aligned", i);
    DS2_REQUIRE(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0, "ds2_gemm_bf16_tn_splitk_group: problem %d: operands must be 16-byte aligned", i);
    // consecutive problems with the same C = terms of one product
    int term = 0;
    while (i - term - 1 >= 0 && probs[i - term - 1].C == q.C) ++term;
    int terms = term + 1;
    while (i + (terms - term) < nprob && probs[i + (terms - term)].C == q.C) ++terms;
    if (term > 0)
      DS2_REQUIRE(probs[i - term].M == q.M && probs[i - term].N == q.N && probs[i - term].ldc == q.ldc, "ds2_gemm_bf16_tn_splitk_group: problem %d: terms of one product must agree in M, N, ldc", i);
    TnSProb& p = g.p[i];
    p.A = (const __bf16*)q.A; p.B = (const __bf16*)q.B; p.C = q.C;
    p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    p.ntx = ceil_div(q.N, 256);
    p.ntiles = p.ntx * ceil_div(q.M, 256);
    p.kchunk = ceil_div(ceil_div(q.K, splitk), BK) * BK;
    DS2_REQUIRE(ceil_div(q.K, p.kchunk) == splitk, "ds2_gemm_bf16_tn_splitk_group: problem %d: K=%d does not split %d ways", i, q.K, splitk);
    p.first_item = items;
    items += p.ntiles * splitk;
    const int nslab = terms * splitk;
    p.to_slab = nslab > 1;
    p.slab0 = term * splitk;
    p.nslab = (term == 0 && nslab > 1) ? nslab : 0;
    p.tick = nullptr; p.nslab_all = nslab;
    if (term == 0) {
      p.partial = (float*)((char*)workspace + off);
      if (nslab > 1) { off += (size_t)nslab * q.M * q.N * sizeof(float); elems += (long long)q.M * q.N / 4; any_slab = true; }
    } else {
      p.partial = g.p[i - term].partial;
    }
  }
  // arrival counters behind the slabs (256-byte aligned), one int per tile of every product that has slabs
  const size_t tick_off = (off + 255) & ~(size_t)255;
  size_t tick_bytes = 0;
  for (int i = 0; i < nprob; ++i) {
    if (g.p[i].nslab > 0) { g.p[i].tick = (int*)((char*)workspace + tick_off + tick_bytes); tick_bytes += (size_t)g.p[i].ntiles * sizeof(int); }
    else if (g.p[i].to_slab) {                           // a further term of a product: its first term's counters
      int t0 = i; while (t0 > 0 && probs[t0 - 1].C == probs[i].C) --t0;
      g.p[i].tick = g.p[t0].tick;
    }
  }
  for (int i = nprob; i < TN_MAX_PROBLEMS; ++i) { g.p[i] = g.p[0]; g.p[i].nslab = 0; }
  g.nprob = nprob; g.splitk = splitk; g.nitems = items;
  g.ep_prob = ep_index; g.ep_scale = ep_scale; g.ep_rowv = ep_rowv; g.ep_shift = ep_shift;
  if (ep_index >= 0 && g.p[ep_index].nslab == 0) return 1;       // (single slab, or a further term of a product: nothing reduces it)
  static const char* ord_env = ds2_exp_getenv("DS2_TN_ORDER");
  g.order = !(ord_env && ord_env[0] == '0');
  if (any_slab) DS2_REQUIRE(workspace && workspace_bytes >= off, "ds2_gemm_bf16_tn_splitk_group: workspace too small");
  DS2_ATTR_ONCE((gemm_bf16_tn_glds_kernel<true>), G_LDS);
  // four waves x 128 x 128 (gemm_tn_w4.h) when every k-tile of every slice is full; DS2_GEMM_W4=0: the 8-wave kernel (A/B switch)
  static const char* w4_env = ds2_exp_getenv("DS2_GEMM_W4");
  bool w4 = ds2_is_gfx950() && !(w4_env && w4_env[0] == '0');
  for (int i = 0; i < nprob; ++i) {
    const ds2_tn_problem& q = probs[i];
    const int kc = g.p[i].kchunk, klast = q.K - (splitk - 1) * kc;
    w4 = w4 && (q.K % 64) == 0 && kc >= 128 && klast >= 128 && (long long)q.lda * 128 < (1ll << 31) && (long long)q.ldb * 128 < (1ll << 31);
  }
  // EXPERIMENT of round 6, measured and NOT the default (DS2_TN_FUSED_REDUCE=1 with DS2_EXPERIMENTAL=1 selects it; profiles/r06_experiments.txt):
  // fused reduce in the four-wave kernel — the last workgroup to deliver a slab of a tile sums the tile's slabs in slab order and writes C, the
  // same additions in the same order as splitk_reduce_group_kernel, without that launch.  Bit-identical, and SLOWER: c3 step +1.1 ms with a
  // full device-scope fence per workgroup (768 cache write-back + invalidate operations per launch drop the operand panels the XCD's L2 holds
  // for the workgroups that are still multiplying), +0.6 ms with release-only fences (the last arrivers' 768 KB of slab reads at a single
  // CU's HBM rate sit on the launch's tail, and the write-backs remain) — against 4 x 53 us of reduce launches saved.
  static const char* fr_env = ds2_exp_getenv("DS2_TN_FUSED_REDUCE");
  const bool fused = w4 && any_slab && (fr_env && fr_env[0] == '1') && workspace_bytes >= tick_off + tick_bytes;
  if (!fused) for (int i = 0; i < TN_MAX_PROBLEMS; ++i) g.p[i].tick = nullptr;
  else DS2_HIP(hipMemsetAsync((char*)workspace + tick_off, 0, tick_bytes, (hipStream_t)stream));
  if (w4) {
    DS2_ATTR_ONCE((gemm_bf16_tn_w4_kernel<0>), W4_LDS);
    hipLaunchKernelGGL(gemm_bf16_tn_w4_kernel<0>, dim3(items), dim3(256), W4_LDS, (hipStream_t)stream, g);
    DS2_LAUNCH_CHECK("gemm_bf16_tn_w4_kernel");
  } else {
    BArgs unused{};
    hipLaunchKernelGGL(gemm_bf16_tn_glds_kernel<true>, dim3(items), dim3(512), G_LDS, (hipStream_t)stream, unused, 0, 0, g);
    DS2_LAUNCH_CHECK("gemm_bf16_tn_glds_kernel<grouped>");
  }
  if (any_slab && !fused) {
    int blocks = (int)((elems + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, elems);
    DS2_LAUNCH_CHECK("splitk_reduce_group_kernel");
  }
  return 0;
}

// Several TN products C_p[M_p,N_p] = A_p[K_p,M_p]^T B_p[K_p,N_p] in ONE launch of the co-resident kernel (gemm_tn_group.h): a flat list of
// 128 x 128 tiles, one workgroup per CU (at most `max_workgroups`, rounded down to a multiple of 8) walking it.  No split-K, no workspace.
extern "C" int ds2_gemm_bf16_tn_group(int nprob, const ds2_tn_problem* probs, int max_workgroups, void* stream) {
  DS2_REQUIRE(nprob >= 1 && nprob <= TNG_MAX_PROBLEMS && probs, "ds2_gemm_bf16_tn_group: 1..%d problems", TNG_MAX_PROBLEMS);
  TnGroup g;
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const ds2_tn_problem& q = probs[i];
    DS2_REQUIRE(q.A && q.B && q.C && q.M >= 8 && q.N >= 8 && q.K > 0, "ds2_gemm_bf16_tn_group: problem %d: bad dims M=%d N=%d K=%d", i, q.M, q.N, q.K);
    DS2_REQUIRE((q.M % 8) == 0 && (q.N % 8) == 0 && (q.lda % 8) == 0 && (q.ldb % 8) == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N,
                "ds2_gemm_bf16_tn_group: problem %d: M, N, lda, ldb must be multiples of 8 (M=%d N=%d lda=%d ldb=%d)", i, q.M, q.N, q.lda, q.ldb);
    DS2_REQUIRE(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0, "ds2_gemm_bf16_tn_group: problem %d: operands must be 16-byte aligned", i);
    TnProb& p = g.p[i];
    p.A = (const __bf16*)q.A; p.B = (const __bf16*)q.B; p.C = q.C;
    p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    p.ntx = ceil_div(q.N, 128);
    p.first_tile = tiles;
    tiles += p.ntx * ceil_div(q.M, 128);
  }
  for (int i = nprob; i < TNG_MAX_PROBLEMS; ++i) g.p[i] = g.p[0];
  g.nprob = nprob; g.ntiles = tiles;
  const int cus = ds2_cus_current() < 8 ? 8 : ds2_cus_current();
  int grid = max_workgroups > 0 ? (max_workgroups < cus ? max_workgroups : cus) : cus;
  if (grid > tiles) grid = tiles;
  grid = grid < 8 ? 8 : grid / 8 * 8;                  // the tile walk deals whole runs to the 8 XCDs
  DS2_ATTR_ONCE((gemm_bf16_tn_group_kernel), L_LDS_REQ);
  hipLaunchKernelGGL(gemm_bf16_tn_group_kernel, dim3(grid), dim3(256), L_LDS_REQ, (hipStream_t)stream, g);
  DS2_LAUNCH_CHECK("gemm_bf16_tn_group_kernel");
  return 0;
}

// dst (R, ldd) bf16 = cast(src (R, C) fp32, pitch lds); ldd % 8 == 0, ldd >= C, pad columns zero.
// The BatchNorm1d in front of a recurrent layer folded into the layer's input projection (round 6): with the centred operand yc = y - m0
// (ds2_center_colstats) the projection  BN(y) W^T + b  is  yc (W diag(s))^T + (b + W c),  s = gamma rsqrt(var + eps),  c = beta - delta s.
// One block per weight row: W'[g][i] = bf16(W[g][i] s[i]) (row pitch ldw2, pad columns zero) and bias'[g] = b[g] + sum_i W[g][i] c[i] (fp32
// weights, thread-strided partial sums combined by xor-shuffles and across the four waves in wave order: run-to-run identical); block 0
// also writes s and c (the epilogue of the weight gradient needs them: ds2_scale_rank1_f32).
__global__ __launch_bounds__(256) void wih_fold_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ delta,
                                                       float eps, __bf16* __restrict__ W2, int ldw2, float* __restrict__ bias2,
                                                       float* __restrict__ colscale, float* __restrict__ colshift, int I) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const float* wrow = W + (long long)g * ldw;
  __bf16* orow = W2 + (long long)g * ldw2;
  float dot = 0.f;
  for (int c0 = threadIdx.x * 4; c0 < ldw2; c0 += 1024) {
    bf16x4 o = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    if (c0 < I) {                                          // (I % 4 == 0)
      const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + c0);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(var + c0), ga = *reinterpret_cast<const f32x4*>(gamma + c0);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c0), de = *reinterpret_cast<const f32x4*>(delta + c0);
      f32x4 sc;
      sc.x = ga.x * rsqrtf(vv.x + eps); sc.y = ga.y * rsqrtf(vv.y + eps); sc.z = ga.z * rsqrtf(vv.z + eps); sc.w = ga.w * rsqrtf(vv.w + eps);
      const f32x4 sh = be - de * sc;
      const f32x4 ws = w * sc;
      o = bf16x4{(__bf16)ws.x, (__bf16)ws.y, (__bf16)ws.z, (__bf16)ws.w};
      dot += (w.x * sh.x + w.y * sh.y) + (w.z * sh.z + w.w * sh.w);
      if (g == 0) { *reinterpret_cast<f32x4*>(colscale + c0) = sc; *reinterpret_cast<f32x4*>(colshift + c0) = sh; }
    }
    *reinterpret_cast<bf16x4*>(orow + c0) = o;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) dot += __shfl_xor(dot, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) bias2[g] = (bias ? bias[g] : 0.f) + (((red[0] + red[1]) + red[2]) + red[3]);
}
extern "C" int ds2_wih_fold_bf16(const float* W, int ldw, const float* bias, int R, int I, const float* var, const float* gamma, const float* beta,
                                 const float* delta, float eps, void* W2_bf16, int ldw2, float* bias2, float* colscale, float* colshift, void* stream) {
  DS2_REQUIRE(W && var && gamma && beta && delta && W2_bf16 && bias2 && colscale && colshift && R > 0 && I > 0, "ds2_wih_fold_bf16: null pointer");
  DS2_REQUIRE((I % 4) == 0 && (ldw % 4) == 0 && ldw2 >= I && (ldw2 % 8) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)W2_bf16 % 16) == 0 &&
              ((uintptr_t)var % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && ((uintptr_t)delta % 16) == 0 &&
              ((uintptr_t)colscale % 16) == 0 && ((uintptr_t)colshift % 16) == 0, "ds2_wih_fold_bf16: I %% 4, pitches and 16-byte alignment");
  hipLaunchKernelGGL(wih_fold_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, W, ldw, bias, var, gamma, beta, delta, eps, (__bf16*)W2_bf16, ldw2, bias2,
                     colscale, colshift, I);
  DS2_LAUNCH_CHECK("wih_fold_kernel");
  return 0;
}

// n contiguous bf16 -> fp32 (n % 8 == 0, 16-byte aligned): the widening of bf16 x-projections for a forward recurrence that cannot run as a
// persistent launch (ds2_rnn_fwd_x returned 1: cooldown after a starved launch, or a shape without a persistent kernel)
__global__ __launch_bounds__(256) void widen_bf16_kernel(const bf16x8* __restrict__ src, f32x4* __restrict__ dst, long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = __builtin_nontemporal_load(src + i);
    __builtin_nontemporal_store(f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, dst + 2 * i);
    __builtin_nontemporal_store(f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]}, dst + 2 * i + 1);
  }
}
extern "C" int ds2_cast_f32_from_bf16(const void* src, float* dst, long long n, void* stream) {
  DS2_REQUIRE(src && dst && n > 0 && (n % 8) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "ds2_cast_f32_from_bf16: bad args");
  const long long n8 = n / 8;
  const int blocks = (int)std::min<long long>((n8 + 255) / 256, 8192);
  hipLaunchKernelGGL(widen_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, (f32x4*)dst, n8);
  DS2_LAUNCH_CHECK("widen_bf16_kernel");
  return 0;
}

extern "C" int ds2_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, void* stream) {
  DS2_REQUIRE(src && dst && R > 0 && Cc > 0 && ld_dst >= Cc && (ld_dst % 8) == 0, "ds2_cast_bf16: bad args");
  const long long total = (long long)R * (ld_dst / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  const int vec = ((ld_src % 4) == 0) && (((uintptr_t)src % 16) == 0);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, (__bf16*)dst, ld_dst, R, Cc, vec);
  DS2_LAUNCH_CHECK("cast_bf16_kernel");
  return 0;
}

// dst (R, ld_dst) bf16 = the split form of src (R, C) fp32 (pitch ld_src) for the fp32 mode's three-term bf16 products (split_bf16_kernel):
// order 0 [hi | hi | lo], 1 [hi | lo | hi] (3 blocks), 2 [hi | lo] (2 blocks); every block pad8(C) columns, pad zero-filled.
// ld_dst >= blocks * pad8(C), ld_dst % 8 == 0.
extern "C" int ds2_split_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, int order, void* stream) {
  DS2_REQUIRE(src && dst && R > 0 && Cc > 0 && order >= 0 && order <= 2, "ds2_split_bf16: bad args");
  const int Cp = (Cc + 7) / 8 * 8, nb = order == 2 ? 2 : 3;
  DS2_REQUIRE(ld_dst >= nb * Cp && (ld_dst % 8) == 0 && ((uintptr_t)dst % 16) == 0, "ds2_split_bf16: ld_dst must be >= %d and a multiple of 8", nb * Cp);
  const long long total = (long long)R * (Cp / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  const int vec = ((ld_src % 4) == 0) && (((uintptr_t)src % 16) == 0);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, (__bf16*)dst, ld_dst, R, Cc, Cp, order, vec);
  DS2_LAUNCH_CHECK("split_bf16_kernel");
  return 0;
}

static int launch_cast_transpose(const float* src, int ld_src, void* dst_t, int ld_t, void* dst_r, int ld_r, int R, int Cc, float* colpart,
                                 void* stream) {
  dim3 grid(ceil_div(dst_r ? max(Cc, ld_r) : Cc, 64), ceil_div(ld_t, 64));
  const int vec = ((ld_src % 4) == 0) && (((uintptr_t)src % 16) == 0);
  hipLaunchKernelGGL(cast_transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, (__bf16*)dst_t, ld_t, (__bf16*)dst_r,
                     ld_r, R, Cc, vec, colpart);
  DS2_LAUNCH_CHECK("cast_transpose_bf16_kernel");
  return 0;
}

// dst (C, ldd) bf16 = cast(src (R, C) fp32)^T ; ldd % 8 == 0, ldd >= R, pad columns zero.
extern "C" int ds2_cast_transpose_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, void* stream) {
  DS2_REQUIRE(src && dst && R > 0 && Cc > 0 && ld_dst >= R && (ld_dst % 8) == 0, "ds2_cast_transpose_bf16: bad args");
  return launch_cast_transpose(src, ld_src, dst, ld_dst, nullptr, 0, R, Cc, nullptr, stream);
}

// Both bf16 copies from one read of src (R, C) fp32: dst_r (R, ld_r) row-major (may be NULL: not produced) and dst_t (C, ld_t) transposed; pads zero.
// ld_r % 8 == 0, C <= ld_r < C + 8 ; ld_t % 8 == 0, ld_t >= R.  colsum (C) fp32, optional: column sums of src from the same
// read (the bias gradient), needs ws of ds2_cast_bf16_both_workspace_bytes(R, C).
extern "C" size_t ds2_cast_bf16_both_workspace_bytes(int R, int Cc) { return (size_t)ceil_div(R, 64) * Cc * 2 * sizeof(float); }

extern "C" int ds2_cast_bf16_both(const float* src, int ld_src, void* dst_r, int ld_r, void* dst_t, int ld_t, int R, int Cc, float* colsum,
                                  void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(src && dst_t && R > 0 && Cc > 0, "ds2_cast_bf16_both: bad args");
  DS2_REQUIRE(ld_t >= R && (ld_t % 8) == 0 && (!dst_r || (ld_r >= Cc && ld_r <= ceil_div(Cc, 64) * 64 && (ld_r % 8) == 0)),   // (the kernel's 64-column tiles write zeros up to there)
              "ds2_cast_bf16_both: bad pitches (ld_r=%d ld_t=%d)", ld_r, ld_t);
  if (colsum) DS2_REQUIRE(ws && ws_bytes >= ds2_cast_bf16_both_workspace_bytes(R, Cc), "ds2_cast_bf16_both: workspace too small");
  int rc = launch_cast_transpose(src, ld_src, dst_t, ld_t, dst_r, ld_r, R, Cc, colsum ? (float*)ws : nullptr, stream);
  if (rc || !colsum) return rc;
  // the partial rows of tiles past R (ld_t > R padding) are never produced: exactly ceil(R/64) row tiles hold data
  return ds2i_col_finalize_sums((const float*)ws, ceil_div(R, 64), Cc, colsum, nullptr, (hipStream_t)stream);
}

extern "C" int ds2_transpose_bf16(const void* src, int ld_src, void* dst_t, int ld_t, int R, int Cc, float* colsum, void* ws, size_t ws_bytes,
                                  void* stream) {
  DS2_REQUIRE(src && dst_t && R > 0 && Cc > 0, "ds2_transpose_bf16: bad args");
  DS2_REQUIRE(ld_t >= R && (ld_t % 8) == 0 && (ld_src % 8) == 0 && ((uintptr_t)src % 16) == 0, "ds2_transpose_bf16: bad pitches / alignment");
  if (colsum) DS2_REQUIRE(ws && ws_bytes >= ds2_cast_bf16_both_workspace_bytes(R, Cc), "ds2_transpose_bf16: workspace too small");
  dim3 grid(ceil_div(Cc, 64), ceil_div(ld_t, 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, ld_src, (__bf16*)dst_t, ld_t, R, Cc,
                     colsum ? (float*)ws : nullptr);
  DS2_LAUNCH_CHECK("transpose_bf16_kernel");
  if (!colsum) return 0;
  return ds2i_col_finalize_sums((const float*)ws, ceil_div(R, 64), Cc, colsum, nullptr, (hipStream_t)stream);
}
