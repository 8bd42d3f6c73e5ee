// NT bf16 GEMM, 256 x 256 x 64 tile, FOUR waves (one per SIMD) x 128 x 128, v_mfma_f32_32x32x16_bf16, accumulators in AGPRs.
// (included by gemm_bf16.hip behind gemm_nt_w4.h, whose DMA helper and LDS image it shares)
//
// Why this MFMA shape: ONE wave per SIMD cannot issue v_mfma_f32_16x16x32_bf16 back to back at the pipe's rate — an MFMA-only loop of 4
// waves per CU reaches 1.50 PF/s with it even on zero operands (scripts/probe_mfma_power.hip: issue-bound, not power-bound), and the
// 16x16x32 form of this kernel (gemm_nt_w4.h) runs its MFMA + LDS part alone at 1.38 PF/s.  A 32x32x16 MFMA occupies the pipe for 32 clocks
// per issue: the same probe reaches 2.45 PF/s (zeros) / 1.62 PF/s (random operands: the power limit) with 4 waves of 16 accumulator tiles.
//
//   registers   acc[4][4] f32x16 = 256 AGPRs;  the fragments of all four k-steps of a tile (4 x (4 A + 4 B) x 4) = 128 VGPRs
//   LDS / DMA   as gemm_nt_w4.h (and the 8-wave kernel): [2 buffers][A | B] x 256 rows x 128 B, slot p of row r holds k-segment p ^ ((r >> 1) & 7)
//   k-tile kt   64 MFMAs; slot s = the issue shadow of MFMA s (k-step s >> 4; 32 clocks = 8 issue slots each, at most 2 used):
//      s  0-15  reads of k-steps 1 and 2 (8 + 8 ds_read_b128); k-step 1 starts behind lgkmcnt(8)
//      s 16-23  reads of k-step 3;  s 31: lgkmcnt(0) — this wave has read everything of buffer cur;  s 32: BARRIER X — everybody has
//      s 33-44  DMA of tile kt + 2 -> cur, 12 of 16 instructions
//      s 45     vmcnt(12): own DMA of tile kt + 1 (issued one k-tile ago) has landed;  s 46: BARRIER Y — everybody's has
//      s 47-54  reads of tile kt + 1, k-step 0, from the other buffer;  s 55-58 the last 4 DMA instructions;  end: lgkmcnt(0)
//   Two tiles of DMA are in flight from s 33 to s 45 of the next k-tile's predecessor... i.e. a tile is waited for ~76 MFMAs (2400 clocks)
//   after its first instruction went out, and the next one is already on its way for 12 MFMAs by then.
// Product TRANSPOSED (B fragment first): a lane holds, per 32 x 32 accumulator tile, 4 x 4 consecutive columns of one C row -> 16-byte stores
// straight from the registers (32 rows x 32 B per instruction; the four stores of a tile fill whole 128-byte lines back to back).
// Same products in the same order as the 8-wave kernel: bit-identical results.
template <int DBG = 0>
__global__ __launch_bounds__(256) void gemm_bf16_nt_w4b_kernel(BArgs g, int ntx, int nty) {
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const int nt = ntx * nty;
  int orig = blockIdx.x;
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
  const int nkt = g.K >> 6;                              // K % 64 == 0, nkt >= 2
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;

  // ---- DMA (see gemm_nt_w4.h)
  unsigned voffA[8], voffB[8];
  const char* sA;
  const char* sB;
  auto retarget = [&](int tm0, int tn0) {
    int rl = lane;
    asm volatile("" : "+v"(rl));
    const int prow = rl >> 3;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int r = (wave + 4 * p) * 8 + prow;
      const int seg = (rl & 7) ^ ((r >> 1) & 7);
      voffA[p] = (unsigned)(min(r, g.M - 1 - tm0) * g.lda * 2 + seg * 16);
      voffB[p] = (unsigned)(min(r, g.N - 1 - tn0) * g.ldb * 2 + seg * 16);
    }
    sA = reinterpret_cast<const char*>(g.A + ((DBG & 2) ? 0ll : (long long)tm0 * g.lda));
    sB = reinterpret_cast<const char*>(g.B + ((DBG & 2) ? 0ll : (long long)tn0 * g.ldb));
  };
  retarget(m0, n0);
  const unsigned dstw = lds0 + wave * 1024;
  // DBG (timing experiments, WRONG RESULTS): 1 = no operand DMA in the steady state, 2 = the DMA re-reads one L2-resident 64 KB, 4 = no MFMA
  auto dma = [&](unsigned boff, int q) {                 // q = 0..7: A pieces, 8..15: B pieces
    if (DBG & 1) return;
    if (q < 8) w4_dma(voffA[q & 7], sA, dstw + boff + (q & 7) * 4096);
    else w4_dma(voffB[q & 7], sB, dstw + boff + 32768 + (q & 7) * 4096);
  };

  // ---- fragment reads: lane -> row (lane & 31) of a 32-row block, k-segment 2 * k-step + (lane >> 5); k-step ks = address ^ (ks << 5)
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned swz0 = (unsigned)((half ^ ((l31 >> 1) & 7)) << 4);
  const unsigned raA0 = lds0 + (wm * 128 + l31) * 128 + swz0;
  const unsigned raB0 = lds0 + 32768 + (wn * 128 + l31) * 128 + swz0;

  const bool bias_ok = g.bias != nullptr;
  const char* bias_base = bias_ok ? reinterpret_cast<const char*>(g.bias) : reinterpret_cast<const char*>(g_zero16);
  f32x16 acc[4][4];
  f32x4 fa[4][4], fb[4][4];                              // [k-step][32-row block]
#define WB_BC(x) __builtin_bit_cast(bf16x8, x)
#define WB_SB() __builtin_amdgcn_sched_barrier(0)
#define WB_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // read r (0-3: A blocks, 4-7: B blocks) of k-step KS from the buffer at byte offset BO
#define WB_RD8(KS, r, BO)                                                                                                             \
  do {                                                                                                                                \
    if ((r) < 4) WB_RD(fa[KS][(r) & 3], (raA0 ^ ((KS) << 5)) + (BO), ((r) & 3) * 4096);                                               \
    else WB_RD(fb[KS][(r) & 3], (raB0 ^ ((KS) << 5)) + (BO), ((r) & 3) * 4096);                                                       \
  } while (0)
  // LDS reads retired down to N outstanding; names k-step KS's fragment registers so that their consumers stay behind it
#define WB_LGKM(N, KS)                                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                                \
               : "+v"(fa[KS][0]), "+v"(fa[KS][1]), "+v"(fa[KS][2]), "+v"(fa[KS][3]), "+v"(fb[KS][0]), "+v"(fb[KS][1]), "+v"(fb[KS][2]),  \
                 "+v"(fb[KS][3])                                                                                                      \
               : "n"(N)                                                                                                               \
               : "memory")
#define WB_MFMA(KS, t)                                                                                                                \
  if (!(DBG & 4)) acc[(t) & 3][(t) >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WB_BC(fb[KS][(t) >> 2]), WB_BC(fa[KS][(t) & 3]), acc[(t) & 3][(t) >> 2], 0, 0, 0)
  // one k-tile.  DMA: stage tile kt + 2; VMW: the vmcnt wait in front of barrier Y as a string ("" = none); NEXT: read tile kt + 1's k-step 0
#define WB_KTILE(DMA, VMW, NEXT, BIAS)                                                                                                    \
  do {                                                                                                                                \
    const unsigned nboff = boff ^ 65536u;                                                                                             \
    WB_SB();                                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                                                               \
      WB_MFMA(0, s_);                                                                                                                 \
      WB_SB();                                                                                                                        \
      if (s_ < 8) WB_RD8(1, s_, boff);                                                                                                \
      else WB_RD8(2, s_ - 8, boff);                                                                                                   \
      WB_SB();                                                                                                                        \
    }                                                                                                                                 \
    WB_LGKM(8, 1);                                                                                                                    \
    WB_SB();                                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                                                               \
      WB_MFMA(1, s_);                                                                                                                 \
      WB_SB();                                                                                                                        \
      if (s_ < 8) WB_RD8(3, s_, boff);                                                                                                \
      if (s_ == 15) { WB_LGKM(0, 2); WB_LGKM(0, 3); }                                                                                 \
      WB_SB();                                                                                                                        \
    }                                                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                                                               \
      WB_MFMA(2, s_);                                                                                                                 \
      WB_SB();                                                                                                                        \
      if (s_ == 0) __builtin_amdgcn_s_barrier();                                                                                      \
      if ((DMA) && s_ >= 1 && s_ <= 12) dma(boff, s_ - 1);                                                                            \
      if ((NEXT) && s_ == 13) asm volatile(VMW ::: "memory");                                                                         \
      if ((NEXT) && s_ == 14) __builtin_amdgcn_s_barrier();                                                                           \
      if ((NEXT) && s_ == 15) WB_RD8(0, 0, nboff);                                                                                    \
      WB_SB();                                                                                                                        \
    }                                                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) {                                                                               \
      WB_MFMA(3, s_);                                                                                                                 \
      WB_SB();                                                                                                                        \
      if ((NEXT) && s_ < 7) WB_RD8(0, s_ + 1, nboff);                                                                                 \
      if ((DMA) && s_ >= 7 && s_ <= 10) dma(boff, 12 + s_ - 7);                                                                       \
      if ((DMA) && s_ == 11 && !(DBG & 2)) { sA += 128; sB += 128; }                                                                  \
      if (BIAS) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pbv[s_ >> 2][s_ & 3])                                            \
                             : "v"(bias_ok ? min(bcol + (s_ >> 2) * 32 + (s_ & 3) * 8, g.N - 4) * 4 : 0), "s"(bias_base) : "memory");          \
      WB_SB();                                                                                                                        \
    }                                                                                                                                 \
    if (NEXT) WB_LGKM(0, 0);                                                                                                          \
    WB_SB();                                                                                                                          \
    boff = nboff;                                                                                                                     \
  } while (0)

  auto stage_first_two = [&]() {
#pragma unroll
    for (int q = 0; q < 16; ++q) dma(0u, q);
    sA += 128; sB += 128;
#pragma unroll
    for (int q = 0; q < 16; ++q) dma(65536u, q);
    sA += 128; sB += 128;
  };
  stage_first_two();
  bool after_epilogue = false;
  for (;;) {
    // k-tiles 0 and 1 have landed (behind an epilogue its 64 stores are younger than that DMA: 63 = the counter's maximum retires the 32 DMA
    // instructions and one store)
    if (after_epilogue) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WB_SB();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WB_SB();
#pragma unroll
    for (int r = 0; r < 8; ++r) WB_RD8(0, r, 0u);
    WB_LGKM(0, 0);
    unsigned boff = 0u;
    int kt = 0;
    f32x4 pbv[4][4];                                     // bias values; first of this lane's columns (set in front of the last k-tile)
    int bcol = 0;
    if (nkt > 2) {                                       // k-tile 0: tile 1 is known to have landed
      WB_KTILE(1, "", 1, 0);
      kt = 1;
    }
    for (; kt < nkt - 2; ++kt) WB_KTILE(1, "s_waitcnt vmcnt(12)", 1, 0);
    if (kt < nkt - 1) WB_KTILE(0, "s_waitcnt vmcnt(0)", 1, 0);
    const bool more = orig + (int)gridDim.x < nt;
    int m0n = 0, n0n = 0;
    if (more) {
      tile_origin(orig + (int)gridDim.x, m0n, n0n);
      retarget(m0n, n0n);
    }
    // bias of this lane's columns: fetched in the shadow of the last k-step's MFMAs (three of the four fragment sets are dead by then) and
    // waited for BEFORE the next tile's DMA goes out (a wait for a load issued behind that DMA would wait for the DMA as well).  Columns
    // beyond N are clamped (their lanes store nothing).
    {
      int el = lane;
      asm volatile("" : "+v"(el));
      bcol = n0 + wn * 128 + (el >> 5) * 4;
    }
    WB_KTILE(0, "", 0, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pbv[j][0]), "+v"(pbv[j][1]), "+v"(pbv[j][2]), "+v"(pbv[j][3]) : : "memory");
    if (more) stage_first_two();
    // ---- epilogue: lane holds C[m0 + wm*128 + i*32 + (lane & 31)][n0 + wn*128 + j*32 + q*8 + (lane >> 5)*4 .. + 3] = acc[i][j][4q .. 4q + 3]
    {
      float* C = g.C;
      const long long ldc = g.ldc;
      const bool stream_out = (g.nt_store & 1) != 0;
      int el = lane;
      asm volatile("" : "+v"(el));
      const int er = el & 31, eh = el >> 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 128 + i * 32 + er;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * 128 + j * 32 + q * 8 + eh * 4;
            f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} + pbv[j][q];
            f32x4* pc = (row < g.M && col < g.N) ? reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) : reinterpret_cast<f32x4*>(g_sink16);
            if (stream_out) __builtin_nontemporal_store(v, pc);
            else *pc = v;
          }
        }
      }
    }
    if (!more) break;
    orig += (int)gridDim.x; m0 = m0n; n0 = n0n;
    after_epilogue = true;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef WB_BC
#undef WB_SB
#undef WB_RD
#undef WB_RD8
#undef WB_LGKM
#undef WB_MFMA
#undef WB_KTILE
}
