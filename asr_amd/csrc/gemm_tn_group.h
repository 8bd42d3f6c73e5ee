// GROUPED TN-form GEMM in a CO-RESIDENT footprint — textually included by gemm_bf16.hip inside its anonymous namespace.
//
// What it is for.  The weight gradients of a recurrent layer (dW_hh of both directions, the n-gate rows of a GRU's dW_hh, dW_ih;
// asr_deepspeech/modules/blocks.py:76-78,88 replayed by autograd) are off the critical path of backward, and the persistent backward
// recurrence of the layer BELOW (rnn_bwd_ksplit_kernel: one workgroup per CU for the whole layer, 2 waves per SIMD, matrix pipe ~13 %
// busy, 10 KB of LDS) leaves most of every CU idle.  gemm_bf16_tn_glds_kernel cannot go there: 8 waves x 213 registers and 128 KB of LDS
// do not fit beside 8 waves x 192 registers.  This kernel is sized for what IS left on such a CU:
//   * 4 waves (ONE per SIMD) of <= 128 registers: the K-split kernel's two waves take 2 x 192 of a SIMD's 512;
//   * 64 KB of operand buffers, but 84 KB REQUESTED: two of these can never share a CU (2 x 84 > 160), so a persistent recurrence
//     workgroup always finds its 8 wave slots / 384 registers / 10 KB free whichever of the two kernels is dispatched first — residency
//     of the recurrence holds by construction, there is nothing to starve;
//   * tile 128 x 128 x 64 per workgroup, wave tile 64 x 64 = 2 x 2 MFMA 32x32x16: 64 accumulator registers.
// One launch covers ALL products of a layer: a problem list in the kernel arguments, one flat tile list, one workgroup per CU walking it
// (tile = first + i * stride inside its XCD's contiguous run, so the 32 workgroups of an XCD work on neighbouring tiles and share operand
// panels in that XCD's L2).  K is the long dimension here (T*B = 32 k): every tile is a full reduction — no split-K slabs, no reduce pass.
// LDS image, fragment gather (ds_read_b64_tr_b16) and the software pipeline are those of gemm_bf16_tn_glds_kernel with 256-byte rows:
// an operand tile is 64 k-rows x 128 columns; one DMA wave-instruction lands four k-rows; the 16-byte slot s of row r is stored at
// slot s ^ ((r & 3) << 2).
constexpr int L_TILE = 64 * 256;                  // bytes of one operand tile (64 k-rows x 128 bf16)
constexpr int L_BUF = 2 * L_TILE;                 // [A | B]
// FOUR stages: with one wave per SIMD a k-tile is 16 MFMAs = 512 clocks (0.2 us) of work, far less than a trip to L2 / HBM, so the DMA of
// k-tile kt + 3 is issued while kt is consumed (two to three tiles = 64-96 KB per CU in flight); the double buffer of the first version
// waited a full memory latency per k-tile (1.5 ms per c3 layer against 0.86 ms for round 3's kernels, profiles/r04_wgrad_side_ab.txt).
constexpr int L_STAGES = 4;
constexpr int L_LDS = L_STAGES * L_BUF;           // 128 KiB
constexpr int L_PATCH = 4 * 32 * 40 * 4;          // epilogue: 4 wave-private patches of 32 rows x 40 floats, IN the (then idle) stage buffers
constexpr int L_LDS_REQ = L_LDS;                  // > 80 KiB: at most one of these workgroups per CU; + the recurrence's 10 KB <= 160 KB
static_assert(L_PATCH <= L_LDS, "patch must fit in the stage buffers");

constexpr int TNG_MAX_PROBLEMS = 8;              // products per launch of THIS kernel (its 106 SGPRs hold the selected problem's fields)

struct TnProb {
  const __bf16* A; const __bf16* B; float* C;
  int M, N, K, lda, ldb, ldc;
  int ntx, first_tile;                            // column tiles; index of this problem's first tile in the flat list
};
struct TnGroup {
  TnProb p[TNG_MAX_PROBLEMS];
  int nprob, ntiles;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_bf16_tn_group_kernel(TnGroup g) {
  constexpr int NI = 2, NJ = 2, NP = 4, NR = NI + NJ;
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  // the zero page's address as an OPAQUE scalar: a known constant makes the compiler split every "k-row valid ? operand : zero page" select
  // of the staging code into two divergent arms with their own DMA instruction (6 branches per k-tile in the main loop)
  const char* zp = reinterpret_cast<const char*>(g_zero16);
  asm volatile("" : "+s"(zp));

  // this workgroup's tiles: XCD x (workgroups are dealt round-robin to the 8 XCDs) owns one contiguous run of the flat tile list
  const int nt = g.ntiles;
  const int xcd = blockIdx.x & 7, q8 = nt >> 3, r8 = nt & 7;
  const int run0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int run1 = run0 + (xcd < r8 ? q8 + 1 : q8);
  const int stride = (int)gridDim.x >> 3;

  // fragment read addresses (buffer 0, k-step 0, first of the two reads of a fragment): see gemm_bf16_tn_glds_kernel
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;
  const int p16 = lane & 15, q4 = p16 >> 2, g16 = (lane >> 4) & 1;
  const unsigned rowpart = (unsigned)((half * 8 + q4) * 256);
  unsigned va[NI], vb[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int col = wm * 64 + i * 32 + g16 * 16 + 4 * (p16 & 3);
    va[i] = lds0 + rowpart + ((((col >> 3) ^ (q4 << 2)) << 4) | (((col >> 2) & 1) << 3));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = wn * 64 + j * 32 + g16 * 16 + 4 * (p16 & 3);
    vb[j] = lds0 + L_TILE + rowpart + ((((col >> 3) ^ (q4 << 2)) << 4) | (((col >> 2) & 1) << 3));
  }
  // staging: piece p = wave + 4 i holds k-rows 4p .. 4p + 3 (1 KiB); lane -> k-row (lane >> 4), physical slot (lane & 15)
  const int rsub = lane >> 4;                                     // = (k-row & 3) for every piece
  const int gslot = (lane & 15) ^ (rsub << 2);                    // logical 16-byte segment this lane fetches
  float* patch = reinterpret_cast<float*>(ldsg) + wave * (32 * 40);

  for (int tile = run0 + ((int)blockIdx.x >> 3); tile < run1; tile += stride) {
    // the tile's problem: selected field by field with wave-uniform compares (a dynamically indexed kernel-argument struct would be
    // copied to scratch)
    const __bf16* PA = g.p[0].A; const __bf16* PB = g.p[0].B; float* PC = g.p[0].C;
    int M = g.p[0].M, N = g.p[0].N, K = g.p[0].K, lda = g.p[0].lda, ldb = g.p[0].ldb, ldc_ = g.p[0].ldc, ntx = g.p[0].ntx, first = 0;
#pragma unroll
    for (int k = 1; k < TNG_MAX_PROBLEMS; ++k)
      if (k < g.nprob && tile >= g.p[k].first_tile) {
        PA = g.p[k].A; PB = g.p[k].B; PC = g.p[k].C;
        M = g.p[k].M; N = g.p[k].N; K = g.p[k].K; lda = g.p[k].lda; ldb = g.p[k].ldb; ldc_ = g.p[k].ldc; ntx = g.p[k].ntx;
        first = g.p[k].first_tile;
      }
    const int tl = tile - first;
    const int m0 = (tl / ntx) * 128, n0 = (tl % ntx) * 128;
    const int nkt = (K + BK - 1) / BK;
    // columns past M / N are clamped to the last valid segment (they only feed C rows / columns that are never stored); k-rows past K
    // read the zero page (pointer select per lane, no control flow in the loop)
    const char* qA = reinterpret_cast<const char*>(PA + (long long)(wave * 4 + rsub) * lda + min(m0 + gslot * 8, M - 8));
    const char* qB = reinterpret_cast<const char*>(PB + (long long)(wave * 4 + rsub) * ldb + min(n0 + gslot * 8, N - 8));
    const long long pieceA = (long long)lda * 32, pieceB = (long long)ldb * 32;        // bytes between pieces i, i + 1 (16 k-rows)
    const long long stepA = (long long)lda * (BK * 2), stepB = (long long)ldb * (BK * 2);
    int kleft = K - (wave * 4 + rsub);                // k-rows left from this lane's row of piece 0 of the tile being staged
    auto stage_piece = [&](int buf, int i) {          // tiles are staged in order: the pointers just advance
      char* dA = ldsg + buf * L_BUF + (wave + 4 * i) * 1024;
      const bool kok = kleft > 16 * i;
      const char* pa = qA + i * pieceA;
      const char* pb = qB + i * pieceB;
      asm volatile("" : "+v"(pa), "+v"(pb));          // (computed unconditionally: the selects below stay two v_cndmask each, no branch)
      glds16(kok ? pa : zp, dA);
      glds16(kok ? pb : zp, dA + L_TILE);
      if (i == NP - 1) { qA += stepA; qB += stepB; kleft -= BK; }
    };
    auto stage_tile = [&](int buf) {
#pragma unroll
      for (int i = 0; i < NP; ++i) stage_piece(buf, i);
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 fa[2][NI], fb[2][NJ];
    // fragment r of a k-step in the order the next k-step consumes them: A0, B0, B1, A1
#define L_ISA(r_) ((r_) == 0 || (r_) > NJ)
#define L_IDX(r_) ((r_) == 0 ? 0 : (r_) <= NJ ? (r_) - 1 : (r_) - NJ)
#define L_RD2(dst, addr, off)                                                                                                         \
  do {                                                                                                                                \
    f32x2 lo_, hi_;                                                                                                                   \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                        \
                 : "=&v"(lo_), "=&v"(hi_)                                                                                             \
                 : "v"(addr), "n"(off), "n"((off) + 1024));                                                                           \
    dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);                                                                              \
  } while (0)
#define L_RD1(set, bufoff, kk, r_)                                                                                                    \
  do {                                                                                                                                \
    if (L_ISA(r_))                                                                                                                    \
      L_RD2(fa[set][L_ISA(r_) ? L_IDX(r_) : 0], va[L_ISA(r_) ? L_IDX(r_) : 0] + (bufoff), (kk) * 4096);                               \
    else                                                                                                                              \
      L_RD2(fb[set][!L_ISA(r_) ? L_IDX(r_) : 0], vb[!L_ISA(r_) ? L_IDX(r_) : 0] + (bufoff), (kk) * 4096);                             \
  } while (0)
#define L_RETIRE_ALL(WAITSTR)                                                                                                         \
  asm volatile(WAITSTR " lgkmcnt(0)"                                                                                                  \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[1][0]),      \
                 "+v"(fb[1][1])                                                                                                       \
               :                                                                                                                      \
               : "memory")
    // One k-step: MFMAs of fragment set `cur` row by row; behind MFMA m goes read m of k-step kk_n into set `nxt` (and piece m of tile
    // kt + 2 when DMA).  Counted waits as in the 256 x 256 kernels (a fragment is two reads): row 0 needs A0, B0, B1 of the previous
    // step's four reads (A1 may still fly: 2), row 1 needs A1 with this step's first two reads behind it (4).
#define L_STEP(cur, nxt, off_n, kk_n, DMA)                                                                                            \
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
      L_RD1(nxt, off_n, kk_n, m_);                                                                                                    \
      if ((DMA) && kt + (L_STAGES - 1) < nkt) stage_piece((kt + (L_STAGES - 1)) & (L_STAGES - 1), m_);                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
    }                                                                                                                                 \
  } while (0)
    // prologue: k-tiles 0 .. L_STAGES - 2 go out; tile 0 has landed when at most the DMA of the tiles behind it is outstanding
    stage_tile(0);
    if (nkt > 1) stage_tile(1);
    if (nkt > 2) stage_tile(2);
    if (nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory");
    else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) L_RD1(0, 0u, 0, r);
    for (int kt = 0; kt < nkt; ++kt) {
      const unsigned boff = (kt & (L_STAGES - 1)) * L_BUF, noff = ((kt + 1) & (L_STAGES - 1)) * L_BUF;
      __builtin_amdgcn_sched_barrier(0);
      L_STEP(0, 1, boff, 1, false);
      L_STEP(1, 0, boff, 2, false);
      L_STEP(0, 1, boff, 3, false);
      // own DMA of tile kt + 1 has landed (behind it only tile kt + 2's eight instructions may be outstanding; tile kt + 3 goes out below)
      // and own reads of buffer kt are complete; past the barrier that holds for every wave: tile kt + 1 may be read, and the buffer of
      // tile kt - 1 (every wave left it an iteration ago) may be refilled with tile kt + 3
      if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      L_RETIRE_ALL("s_waitcnt");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      L_STEP(1, 0, noff, 0, true);                     // (after the last k-tile: harmless reads of stale LDS, retired below)
    }
    L_RETIRE_ALL("s_waitcnt");
#undef L_RD1
#undef L_RD2
#undef L_ISA
#undef L_IDX
#undef L_RETIRE_ALL
#undef L_STEP

    // ---- epilogue: every 32 x 32 accumulator tile through the wave-private patch, out as 16-byte stores (8 rows x 128 B per instruction)
    __syncthreads();                                   // the patches lie in the stage buffers: every wave has left the k-loop
    {
      constexpr int EP = 40;
      float* C = PC;
      const long long ldc = ldc_;
      const bool wide = (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && (N % 4) == 0;
      const int prow = lane >> 3, pc4 = (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (wide) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int col = n0 + wn * 64 + j * 32 + pc4;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int rl = it * 8 + prow;
              const int row = m0 + wm * 64 + i * 32 + rl;
              const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[rl * EP + pc4]);
              if (row < M && col < N) *reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) = v;
            }
            __builtin_amdgcn_wave_barrier();
          } else {
            const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (row < M && col < N) C[(long long)row * ldc + col] = acc[i][j][r];
            }
          }
        }
      }
    }
    __syncthreads();                                   // every wave is done with the operand buffers before the next tile's DMA
  }
}
