// NT bf16 GEMM, 256 x 256 tile, persistent, operands through a FOUR-STAGE LDS RING of 32-deep k-slices.   (included by gemm_bf16.hip)
//
// Why: the double-buffered 256 x 256 x 64 kernel above waits `vmcnt(0)` once per k-tile — at that moment NOTHING is in flight, the next
// 64 KB burst then needs a full memory round trip before its first byte lands, and the k-tile period becomes (64 KB at the CU's ingest rate,
// ~22 B/clk) + (one round trip) = 2900 + 600 clocks against 2048 clocks of MFMA.  Two 64 KB buffers cannot do better: a buffer is refilled
// only after its tile has been consumed.  Here the unit of staging is a 32-deep slice (A 256 x 32 + B 256 x 32 = 32 KB): four of them fit
// the same 128 KB, the wait at the sync point of slice s is a COUNTED `vmcnt` that retires slice s+1 only and leaves s+2, s+3 in flight, and
// slice s+4 goes out right behind the barrier into the slot whose reads that barrier has just retired — 64-96 KB are in flight at every moment.
//
//   per slice (2 k-steps of 16, 16 MFMA 32x32x16 per wave):
//      MFMAs (s, k-step 0) | reads (s, k-step 1) behind them
//      own reads complete, own DMA of slice s+1 landed (vmcnt 8), BARRIER
//      MFMAs (s, k-step 1) | reads (s+1, k-step 0) + DMA of slice s+4 into slot s % 4 behind them
//
// LDS image of a slice: [A: 256 rows x 64 B | B: 256 rows x 64 B]; one DMA wave-instruction lands 16 rows (1 KiB, lane-linear).  Bank
// conflicts: a ds_read_b128 lane group covers 16 rows at one k-segment; with 64-byte rows the 16-byte slot p of row r holds k-segment
// p ^ ((r >> 2) & 3) (applied to the per-lane SOURCE address of the DMA and to the read): the group's 16 (r & 3, (r >> 2) & 3) pairs are
// distinct, so it touches all 16 bank quads once.
//
// Persistent: one workgroup per CU walks tiles orig, orig + grid, ...; the slice stream runs across tile boundaries (the DMA of the next
// tile's first slices goes out during the last slices of the current one; the epilogue runs while they land).  vmcnt is ONE in-order
// counter for loads, LDS-DMA and stores, so the counted waits of the three syncs that follow an epilogue include its stores and the
// next tile's bias loads (R_EPS + R_BIASL younger operations); the epilogue stores are unconditional for that (lanes outside the matrix
// write a 16-byte sink).  Products and summation order are those of the double-buffered kernel: results are bit-identical.
constexpr int R_SLICE = 32768;                    // bytes of one slice image
constexpr int R_RING = 4 * R_SLICE;               // 128 KiB
constexpr int R_EPS = 32;                         // epilogue global stores per wave (fp32 output): NI * NJ * 2 halves * 2
constexpr int R_BIASL = 2;                        // bias loads per wave and tile

template <bool OBF>
__global__ __launch_bounds__(512) void gemm_bf16_nt_ring_kernel(BArgs g, int ntx, int nty) {
  constexpr int WN = 4, NWV = 8, NI = 4, NJ = 2;
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const __bf16* A = g.A;
  const __bf16* B = g.B;
  const int nt = ntx * nty;
  int orig = blockIdx.x;
  auto tile_origin = [&](int o, int& tm0, int& tn0) {
    const int xcd = o & 7, q8 = nt >> 3, r8 = nt & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
    tm0 = (tile / ntx) * 256; tn0 = (tile % ntx) * 256;
  };
  int m0, n0;
  tile_origin(orig, m0, n0);
  const int nsl = g.K >> 5;                            // slices per tile (K % 32 == 0, nsl >= 4)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- DMA stream: wave w moves pieces w and w + 8 (16 rows each) of A and of B; q[] = A lo, B lo, A hi, B hi
  const int prow = lane >> 2;
  const int segk = (((lane & 3) ^ ((lane >> 4) & 3)) << 3);     // k offset (elements) of the 16-byte segment this lane fetches
  const char* q[4];
  auto retarget = [&](int tm0, int tn0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave + NWV * i) * 16 + prow;
      q[2 * i] = reinterpret_cast<const char*>(A + (long long)min(tm0 + r, g.M - 1) * g.lda + segk);
      q[2 * i + 1] = reinterpret_cast<const char*>(B + (long long)min(tn0 + r, g.N - 1) * g.ldb + segk);
    }
  };
  auto dma_piece = [&](int slot, int i) {               // i: 0 A lo, 1 B lo, 2 A hi, 3 B hi
    glds16(q[i], ldsg + slot * R_SLICE + (i & 1) * 16384 + (wave + NWV * (i >> 1)) * 1024);
    q[i] += 64;
  };
  // The stream never stops: behind the workgroup's last tile it streams that tile once more (valid memory, slots nobody reads any
  // more), so every slice period issues exactly four DMA instructions per wave and the counted waits hold to the end without a special case.
  int dorig = orig, dleft = nsl;                        // tile whose slices the DMA stream is issuing; slices of it not yet issued
  retarget(m0, n0);
  auto dma_next_slice = [&]() {                         // called once per slice in front of its four pieces
    if (dleft == 0) {
      if (dorig + (int)gridDim.x < nt) dorig += (int)gridDim.x;
      int a, b;
      tile_origin(dorig, a, b);
      retarget(a, b);
      dleft = nsl;
    }
    --dleft;
  };

  // ---- fragment read addresses (slot 0): row (.. + l31), k-segment kk * 2 + half, swizzled with (row >> 2) & 3 = (l31 >> 2) & 3
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;
  const int gsw = (l31 >> 2) & 3;
  unsigned ra[2], rb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const unsigned sl = (unsigned)(((kk * 2 + half) ^ gsw) << 4);
    ra[kk] = lds0 + (wm * 128 + l31) * 64 + sl;
    rb[kk] = lds0 + 16384 + (wn * 64 + l31) * 64 + sl;
  }

  f32x16 acc[NI][NJ];
  f32x4 fa[2][NI], fb[2][NJ];
  constexpr int NR = NI + NJ;
#define R_ISA(r_) ((r_) == 0 || (r_) > NJ)
#define R_IDX(r_) ((r_) == 0 ? 0 : (r_) <= NJ ? (r_) - 1 : (r_) - NJ)
#define R_RD1(set, slotoff, kk, r_)                                                                                                   \
  do {                                                                                                                                \
    if ((r_) < NR && R_ISA(r_))                                                                                                       \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][(r_) < NR && R_ISA(r_) ? R_IDX(r_) : 0]) : "v"(ra[kk] + (slotoff)), \
                   "n"(((r_) < NR && R_ISA(r_) ? R_IDX(r_) : 0) * 2048));                                                             \
    else if ((r_) < NR)                                                                                                               \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][(r_) < NR && !R_ISA(r_) ? R_IDX(r_) : 0]) : "v"(rb[kk] + (slotoff)), \
                   "n"(((r_) < NR && !R_ISA(r_) ? R_IDX(r_) : 0) * 2048));                                                            \
  } while (0)
#define R_RETIRE_ALL(WAITSTR)                                                                                                         \
  asm volatile(WAITSTR " lgkmcnt(0)"                                                                                                  \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]),                      \
                 "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1])                       \
               :                                                                                                                      \
               : "memory")
  // one k-step (see G_STEP of the double-buffered kernel): behind MFMA m goes read m of the next k-step and, when DMA, piece m of the slice
  // the stream is at
#define R_STEP(cur, nxt, off_n, kk_n, DMA, slot_w)                                                                                    \
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
      R_RD1(nxt, off_n, kk_n, m_);                                                                                                    \
      if ((DMA) && m_ < 4) dma_piece(slot_w, m_);                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
    }                                                                                                                                 \
  } while (0)

  // bias of this lane's columns (epilogue layout: 4 consecutive columns per lane), fetched by inline asm so that the loads are counted by
  // hand like everything else in the vector-memory queue (a compiler-visible load would make hipcc wait vmcnt(0) at its first use)
  const char* zp = reinterpret_cast<const char*>(g_zero16);
  f32x4 pbv[NJ];
  auto load_bias = [&](int tn0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = tn0 + wn * (NJ * 32) + j * 32 + (lane & 7) * 4;
      const char* p = (g.bias && col + 4 <= g.N) ? reinterpret_cast<const char*>(g.bias + col) : zp;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pbv[j]) : "v"(p) : "memory");
    }
  };

  // ---- prologue: bias, slices 0 .. 3 of the first tile, first fragments
  load_bias(n0);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dma_next_slice();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(s, i);
  }
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // bias and slice 0 have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) R_RD1(0, 0u, 0, r);

  int slot = 0;                                        // ring slot of the slice being consumed
  int epi_syncs = 0;                                   // syncs left whose vmcnt count includes an epilogue's stores and the bias loads
  for (;;) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < nsl; ++s) {
      const unsigned soff = (unsigned)slot * R_SLICE, noff = (unsigned)((slot + 1) & 3) * R_SLICE;
      __builtin_amdgcn_sched_barrier(0);
      R_STEP(0, 1, soff, 1, false, 0);
      // own reads of this slot are complete and own DMA of the next slice has landed; past the barrier that holds for every wave
      static_assert(8 + R_EPS + R_BIASL == 42, "the literal below");
      if (epi_syncs > 0) {
        R_RETIRE_ALL("s_waitcnt vmcnt(42)");           // 8 + R_EPS + R_BIASL (see the end of the tile loop)
        --epi_syncs;
      } else {
        R_RETIRE_ALL("s_waitcnt vmcnt(8)");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      dma_next_slice();
      R_STEP(1, 0, noff, 0, true, slot);
      slot = (slot + 1) & 3;
    }
    // ---- epilogue of this tile (the next tile's slices 1 .. 3 are landing; its first fragments are on their way into set 0)
    {
      constexpr int EP = 40;
      float* patch = reinterpret_cast<float*>(ldsg + R_RING) + wave * (16 * EP);
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
              // every lane stores (the store count per wave is part of the vmcnt arithmetic): lanes outside the matrix write a sink
              v += pbv[j];
              f32x4* pc = (row < g.M && col < g.N) ? reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) : reinterpret_cast<f32x4*>(g_sink16);
              if (stream_out) __builtin_nontemporal_store(v, pc);
              else *pc = v;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }
    orig += (int)gridDim.x;
    if (orig >= nt) break;
    tile_origin(orig, m0, n0);
    load_bias(n0);
    // vector-memory queue at the next tile's sync s (oldest first): [slice s+1] slice s+2 .. | 32 stores | 2 bias loads | slices issued since.
    // Younger than slice s+1 at sync 0: slices 2, 3 + stores + bias = 8 + 34; at sync 1: slice 3, stores, bias, slice 4 = 8 + 34; at sync 2:
    // stores, bias, slices 4, 5 = 8 + 34; from sync 3 on the stores are older than the awaited slice: 8.
    epi_syncs = 3;
  }
  // the reads issued behind the last MFMAs (set 0, stale LDS) and the surplus DMA
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1])
               :
               : "memory");
#undef R_RD1
#undef R_ISA
#undef R_IDX
#undef R_RETIRE_ALL
#undef R_STEP
}

// ---- PING-PONG over the ring with v_mfma_f32_16x16x32_bf16 ------------------------------------------------------------------------------
// On random operands the chip is POWER-limited, and the 16x16x32 MFMA does the same FLOPs with half the accumulator-file traffic of the
// 32x32x16 one: an MFMA-only loop sustains 1850-1940 TF/s with it against 1540-1720 (scripts/probe_mfma_power.hip, same box).  A phase is a
// whole 32-deep slice: 12 fragment reads (8 A row blocks, 4 B column blocks: a 16 x 32 fragment is exactly the 1 KiB piece one DMA
// instruction lands), the wave's four DMA instructions of slice s+3, then 32 MFMAs = 512 clocks of its SIMD's matrix pipe while the other
// group loads.  Every wave's program:
//        [ 12 reads (s) | DMA of slice s+3 | own DMA of slice s+1 landed | lgkmcnt(0) ]  BARRIER  [ 32 MFMA (s) ]  BARRIER
// with group B (tile rows 128-255: the second wave of every SIMD) one barrier behind (it executes one extra barrier up front).  With B_n the
// n-th barrier: group A reads slice s in (B_2s, B_2s+1) and multiplies in (B_2s+1, B_2s+2); group B reads in (B_2s+1, B_2s+2), multiplies in
// (B_2s+2, B_2s+3).  RAW: slice s+1 is first read by A behind B_2s+2; every wave has waited for its own DMA of it in its read section of
// slice s, the last of which (group B's) ends at B_2s+2.  WAR: the slot of slice s-1 is refilled by the DMA issued in the read sections of
// slice s: its last reads (group B) complete before B_2s, and A's read section of slice s begins behind B_2s.  Slices 0 .. 2 are staged by
// the prologue; at the wait for slice s+1 the younger operations are slices s+2 and s+3 (8 DMA instructions) plus, for the first two slices
// of a tile that follows an epilogue, its 32 stores and the 4 bias loads.
// Swizzle for this fragment shape: lane l reads row l & 15 at k-segment l >> 4; with 64-byte rows the 16-byte slot p of row r holds k-segment
// p ^ ((4 - (r >> 2)) & 3): every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) then covers the 16 bank quads once.
// The product is formed TRANSPOSED (B fragment as the MFMA's first operand): a lane then holds four consecutive columns of one C row, and
// the accumulators leave as 16-byte stores straight from the registers — no LDS patch, the same 32 store instructions per wave.
constexpr int Q_EPS = 32, Q_BIASL = 4;
// DBG (timing experiments, WRONG RESULTS): bit 0 = no operand DMA behind the prologue, bit 1 = no fragment reads, bit 2 = no barriers
// NSLOT: ring slots (4 = 128 KB, 5 = all 160 KB of the LDS: one more slice in flight)
template <bool OBF, int DBG = 0, int NSLOT = 4>
__global__ __launch_bounds__(512) void gemm_bf16_nt_pp16_kernel(BArgs g, int ntx, int nty) {
  constexpr int WN = 4, NWV = 8, NI = 8, NJ = 4;        // 16 x 16 accumulator tiles per wave along M / N (128 x 64)
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const __bf16* A = g.A;
  const __bf16* B = g.B;
  const int nt = ntx * nty;
  int orig = blockIdx.x;
  // XCD-aware tile order: workgroup o runs on XCD o & 7; each XCD walks one contiguous run of a tile sequence in which any 32 consecutive
  // tiles (= what the XCD's 32 CUs hold at one time) form a compact block of SR row tiles x 32 / SR column tiles: an A row tile is then
  // shared by 32 / SR CUs of the same L2 and a B column tile by SR (row-major order shares A 24-fold and B not at all when N = 6144:
  // half of the operand bytes then miss the L2).  Sequence: super-rows of SR row tiles, inside one column by column.
  const int SR = g.super_rows > 0 ? g.super_rows : 1 << 20;
  auto tile_origin = [&](int o, int& tm0, int& tn0) {
    const int xcd = o & 7, q8 = nt >> 3, r8 = nt & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (o >> 3);
    const int per = SR * ntx, sr = t / per, rem = t - sr * per;
    const int rows = min(SR, nty - sr * SR);
    const int tn = rem / rows;
    tm0 = (sr * SR + rem - tn * rows) * 256; tn0 = tn * 256;
  };
  int m0, n0;
  tile_origin(orig, m0, n0);
  const int nsl = g.K >> 5;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int prow = lane >> 2;
  const int segk = (((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) << 3);
  const char* q[4];
  auto retarget = [&](int tm0, int tn0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave + NWV * i) * 16 + prow;
      q[2 * i] = reinterpret_cast<const char*>(A + (long long)min(tm0 + r, g.M - 1) * g.lda + segk);
      q[2 * i + 1] = reinterpret_cast<const char*>(B + (long long)min(tn0 + r, g.N - 1) * g.ldb + segk);
    }
  };
  auto dma_piece = [&](int slot, int i) {
    if ((DBG & 64) && !(i & 1)) return;                  // (bit 6: no A pieces, bit 7: no B pieces)
    if ((DBG & 128) && (i & 1)) return;
    glds16(q[i], ldsg + slot * R_SLICE + (i & 1) * 16384 + (wave + NWV * (i >> 1)) * 1024);
    if (!(DBG & 8)) q[i] += 64;                         // (bit 3: every DMA re-reads the same L2-resident 64 KB)
  };
  int dorig = orig, dleft = nsl;
  retarget(m0, n0);
  auto dma_next_slice = [&]() {
    if (dleft == 0) {
      if (dorig + (int)gridDim.x < nt) dorig += (int)gridDim.x;
      int a, b;
      tile_origin(dorig, a, b);
      if (DBG & 8) a = b = 0;
      retarget(a, b);
      dleft = nsl;
    }
    --dleft;
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;
  const int frow = lane & 15, fseg = lane >> 4;
  const unsigned lanepart = (unsigned)(frow * 64 + ((fseg ^ ((4 - (frow >> 2)) & 3)) << 4));
  const unsigned ra = lds0 + wm * 8192 + lanepart;
  const unsigned rb = lds0 + 16384 + wn * 4096 + lanepart;

  f32x4 acc[NI][NJ];
  f32x4 fa[NI], fb[NJ];
#define Q_PHASE(slotoff, slot_w, VMWAIT)                                                                                              \
  do {                                                                                                                                \
    if (!(DBG & 2))                                                                                                                   \
    asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\t"                    \
                 "ds_read_b128 %3, %12 offset:3072\n\t"                                                                                \
                 "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:1024\n\tds_read_b128 %6, %13 offset:2048\n\t"                    \
                 "ds_read_b128 %7, %13 offset:3072\n\tds_read_b128 %8, %13 offset:4096\n\tds_read_b128 %9, %13 offset:5120\n\t"        \
                 "ds_read_b128 %10, %13 offset:6144\n\tds_read_b128 %11, %13 offset:7168"                                              \
                 : "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3]), "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fa[2]), "=&v"(fa[3]),     \
                   "=&v"(fa[4]), "=&v"(fa[5]), "=&v"(fa[6]), "=&v"(fa[7])                                                              \
                 : "v"(rb + (slotoff)), "v"(ra + (slotoff)));                                                                          \
    if (!(DBG & 1)) {                                                                                                                 \
    dma_piece(slot_w, 0);                                                                                                             \
    dma_piece(slot_w, 1);                                                                                                             \
    dma_piece(slot_w, 2);                                                                                                             \
    dma_piece(slot_w, 3);                                                                                                             \
    VMWAIT; }                                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
                 : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]),             \
                   "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7])                                                                  \
                 :                                                                                                                    \
                 : "memory");                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                                \
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                                    \
    if (!(DBG & 16))                                                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_)                                                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                                               \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j_]), __builtin_bit_cast(bf16x8, fa[i_]), \
                                                             acc[i_][j_], 0, 0, 0);                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                                \
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                                \
  } while (0)

  // bias of this lane's columns (4 consecutive columns per 16-column tile), inline asm: counted by hand in the vector-memory queue
  const char* zp = reinterpret_cast<const char*>(g_zero16);
  f32x4 pbv[NJ];
  auto load_bias = [&](int tn0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = tn0 + wn * 64 + j * 16 + fseg * 4;
      const char* p = (g.bias && col + 4 <= g.N) ? reinterpret_cast<const char*>(g.bias + col) : zp;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pbv[j]) : "v"(p) : "memory");
    }
  };

  load_bias(n0);
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) {
    dma_next_slice();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(s, i);
  }
  if constexpr (NSLOT == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // bias and slice 0 have landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();          // group B runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  int slot = 0;
  int epi_syncs = 0;
  for (;;) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsl; ++s) {
      const unsigned soff = (unsigned)slot * R_SLICE;
      const int wslot = slot == 0 ? NSLOT - 1 : slot - 1;       // the slot of slice s - 1 takes slice s + NSLOT - 1
      dma_next_slice();
      static_assert(8 + Q_EPS + Q_BIASL == 44, "the literal below");
#define Q_VMWAIT                                                                   \
  do {                                                                             \
    if (epi_syncs > 0) {                                                           \
      if constexpr (NSLOT == 5) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");  \
      else asm volatile("s_waitcnt vmcnt(44)" ::: "memory");                       \
      --epi_syncs;                                                                 \
    } else {                                                                       \
      if constexpr (NSLOT == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  \
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        \
    }                                                                              \
  } while (0)
      Q_PHASE(soff, wslot, Q_VMWAIT);
#undef Q_VMWAIT
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    // ---- epilogue: straight from the accumulators (lane: C row m0 + wm*128 + i*16 + (lane & 15), columns .. + j*16 + (lane >> 4)*4 .. + 3)
    {
      float* C = g.C;
      const long long ldc = g.ldc;
      const bool stream_out = (g.nt_store & 1) != 0;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = m0 + wm * 128 + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * 64 + j * 16 + fseg * 4;
          f32x4 v = acc[i][j] + pbv[j];
          f32x4* pc = (row < g.M && col < g.N) ? reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) : reinterpret_cast<f32x4*>(g_sink16);
          if (stream_out) __builtin_nontemporal_store(v, pc);
          else *pc = v;
        }
      }
    }
    orig += (int)gridDim.x;
    if (orig >= nt) break;
    tile_origin(orig, m0, n0);
    load_bias(n0);
    epi_syncs = NSLOT - 2;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();          // group A catches the barrier group B is one behind on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef Q_PHASE
}
