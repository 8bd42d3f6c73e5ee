// NT bf16 GEMM, 256 x 256 x 64 tile, FOUR waves (one per SIMD), 128 x 128 per wave, accumulators in the AGPR half of the register file.
// (included by gemm_bf16.hip)
//
// Why: the 8-wave kernel above (2 x 4 waves of 128 x 64) reads 192 KB of fragments per 256 x 256 x 64 k-tile and lands 64 KB of DMA in the
// same LDS: 256 KB = 2048 clocks of the 128 B/clk array against 2048 clocks of MFMA — LDS-bound by construction.  With 128 x 128
// per wave the fragments of a k-tile are 4 waves x 32 KB = 128 KB: 192 KB in all, 1536 clocks, a quarter of the LDS time free.  The price is
// one wave per SIMD: nothing else issues while that wave waits, so the schedule below is placed by hand — every LDS read, every DMA
// instruction and every barrier sits in the issue shadow of an MFMA (a v_mfma_f32_16x16x32_bf16 occupies its SIMD's matrix pipe for 16
// clocks = 4 issue slots; at most one other instruction is placed behind an MFMA), and a wave arrives at a barrier right after issuing an
// MFMA, so the pipe still has work while the barrier resolves.
//
//   registers   acc[8][8] f32x4 = 256 AGPRs;  fragments of BOTH k-steps of a tile (2 x (8 A + 8 B) x 4) = 128 VGPRs
//   LDS         [2 buffers][A 256 rows x 128 B | B 256 rows x 128 B] = 128 KB, the 8-wave kernel's image and swizzle (slot p of row r holds
//               k-segment p ^ ((r >> 1) & 7)): the 16 rows x one k-segment of a 16x16x32 fragment read fall on 16 distinct bank quads in
//               every hardware lane group of a ds_read_b128 ({0-3, 12-15, 20-27}, ...)
//   DMA         wave w moves the 8-row pieces w, w + 4, ..., w + 28 of A and of B: 16 x global_load_lds_dwordx4 per wave and k-tile; source =
//               SGPR base (advanced by 128 B per k-tile with scalar adds) + one 32-bit lane offset per piece (constant within a tile)
//   k-tile kt   (buffer cur = kt & 1; fragments of its k-step 0 already in registers); slot s = the issue shadow of MFMA s:
//      phase A  64 MFMAs of k-step 0 | 8 reads A(k-step 1) behind MFMAs 0, 2 .. 14 | lgkmcnt(0), BARRIER X_A at 20: nobody reads cur.A any more |
//               DMA A(kt + 2) -> cur.A at 21, 25 .. 49, the 8 reads B(k-step 1) between them at 23, 27 .. 51 | lgkmcnt(0), BARRIER X_B at 55: cur.B
//               is free | DMA B(kt + 2) -> cur.B from 57 on
//      phase B  64 MFMAs of k-step 1 | rest of DMA B at 0, 2 .. 8 | vmcnt(16) at 10: own DMA of tile kt + 1 has landed (tile kt + 2's 16
//               instructions stay in flight), BARRIER Y at 11: everybody's has | 16 reads of tile kt + 1, k-step 0, from the other buffer at
//               13, 16 .. 58 | lgkmcnt(0)
//   Two tiles of DMA are in flight for most of a k-tile; a tile is read one whole k-tile (~1 us) after its DMA went out.
// The product is formed TRANSPOSED (B fragment as the MFMA's first operand): a lane holds four consecutive columns of one C row and the
// accumulators leave as 16-byte stores straight from the registers.
//
// Persistent: one workgroup per CU walks tiles orig, orig + grid, ... in the XCD-aware order of the 8-wave kernel; the DMA of the next tile's
// first two k-tiles goes out before the epilogue's stores.  vmcnt is ONE in-order counter (loads, LDS-DMA, stores; at most 63): the wait at the
// top of a tile that follows an epilogue is vmcnt(63), which retires the 32 DMA instructions issued before the 64 stores and one store.
//
// Measured (profiles/r05_gemm_w4_ab.txt, same box, random operands): dX (K = 6144) 381 -> 342 us, forward projection (K = 1024) 534 -> 508,
// c4's forward projection 646 -> 612; bit-identical to the 8-wave kernel on every shape.  Tried on the way and not kept (same record): the
// same tile on v_mfma_f32_32x32x16_bf16 (slower: 364 us on dX — its MFMA + LDS part alone runs at 1.27 PF/s against 1.38 here), one or two
// barriers per k-tile instead of three (the refill of cur then starts later: 361), an L2 prefetch of the operand stream two k-tiles ahead of its
// DMA (no change), start phases of the workgroups staggered so that their epilogues interleave (slower).  Timing ablations: the MFMA + LDS part
// alone takes 292 us on dX (one wave per SIMD issues 16x16x32 MFMAs at 1.50 PF/s at best: scripts/probe_mfma_power.hip), the operand stream alone
// 270 us (148 from an L2-resident 64 KB), both together 342-383.
constexpr int W4_LDS = 4 * G_TILE;
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

// 1 KiB of an operand tile, global -> LDS: lane offset (bytes) from a uniform base; M0 = LDS destination of lane 0 (one wait state between the
// scalar write of M0 and the LDS-DMA instruction that reads it: inside an asm statement the compiler's hazard recogniser does not see the pair)
__device__ __forceinline__ void w4_dma(unsigned voff, const char* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// OBF: C leaves as bf16 (g.C is a __bf16*, ldc in bf16 elements, N % 8 == 0): the x-projections of a recurrent layer in the bf16 training mode,
// which the forward recurrence reads once (394 MB written + read per c3 layer instead of 788).  A lane's four columns of the 16-column blocks
// j, j + 1 are packed to two dwords each; two v_permlane16_swap hand the odd 16-lane rows' block-j halves to the even rows and the even rows'
// block-(j + 1) halves to the odd rows, so that every lane owns EIGHT consecutive columns of one block: 32 sixteen-byte stores per wave and
// tile (64 B contiguous per row and store instruction, as in the fp32 form) instead of 64.
template <int DBG = 0, bool OBF = false>
__global__ __launch_bounds__(256) void gemm_bf16_nt_w4_kernel(BArgs g, int ntx, int nty) {
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

  // ---- DMA: lane -> (row inside the 8-row piece, 16-byte segment), swizzled on the source side
  unsigned voffA[8], voffB[8];
  const char* sA;                                        // uniform: A + (m0 * lda + k) * 2 of the NEXT k-tile to stage
  const char* sB;
  auto retarget = [&](int tm0, int tn0) {
    int rl = lane;                                       // (opaque copy: the per-piece rows are recomputed per tile instead of living — spilled — across the loop)
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
  const unsigned dstw = lds0 + wave * 1024;              // piece p of A -> + p * 4096, of B -> + 32768 + p * 4096; buffer -> + 65536
  // DBG (timing experiments, WRONG RESULTS): 1 = no operand DMA in the steady state, 2 = the DMA re-reads one L2-resident 64 KB, 4 = no MFMA
  auto dma_a = [&](unsigned boff, int p) { if (!(DBG & 1)) w4_dma(voffA[p], sA, dstw + boff + p * 4096); };
  auto dma_b = [&](unsigned boff, int p) { if (!(DBG & 1)) w4_dma(voffB[p], sB, dstw + boff + 32768 + p * 4096); };

  // ---- fragment reads: lane -> row (lane & 15) of a 16-row block, k-segment (lane >> 4) of the k-step
  const int frow = lane & 15, fseg = lane >> 4;
  const unsigned swz0 = (unsigned)((fseg ^ ((frow >> 1) & 7)) << 4);
  const unsigned raA0 = lds0 + (wm * 128 + frow) * 128 + swz0;           // k-step 0, buffer 0; k-step 1 = ^ 64; buffer 1 = ^ 65536
  const unsigned raB0 = lds0 + 32768 + (wn * 128 + frow) * 128 + swz0;

  f32x4 acc[8][8];
  f32x4 fa[2][8], fb[2][8];
#define W4_BC(x) __builtin_bit_cast(bf16x8, x)
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
#define W4_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // every LDS read of this wave has returned; names the fragment registers so that their consumers stay behind it
#define W4_LGKM0(F)                                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                 \
               : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])                       \
               :                                                                                                                      \
               : "memory")
  // one k-tile.  DMA: stage tile kt + 2 (0 / 1); VMW: the vmcnt wait in front of barrier Y as a string ("" = none); NEXT: read tile kt + 1's
  // k-step 0 (see the schedule at the top of the file)
#define W4_KTILE(DMA, VMW, NEXT)                                                                                                      \
  do {                                                                                                                                \
    const unsigned cA1 = (raA0 ^ 64u) + boff, cB1 = (raB0 ^ 64u) + boff, nA0 = raA0 + (boff ^ 65536u), nB0 = raB0 + (boff ^ 65536u);   \
    W4_SB();                                                                                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 64; ++m_) {                                                                               \
      if (!(DBG & 4)) acc[m_ & 7][m_ >> 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W4_BC(fb[0][m_ >> 3]), W4_BC(fa[0][m_ & 7]), acc[m_ & 7][m_ >> 3], 0, 0, 0); \
      W4_SB();                                                                                                                        \
      if (m_ < 16 && !(m_ & 1)) W4_RD(fa[1][(m_ >> 1) & 7], cA1, ((m_ >> 1) & 7) * 2048);                                             \
      if (m_ == 19) W4_LGKM0(fa[1]);                                                                                                  \
      if (m_ == 20) __builtin_amdgcn_s_barrier();                                                                                     \
      if (m_ >= 21 && m_ <= 51 && (m_ & 1)) {                                                                                         \
        if (((m_ - 21) >> 1) & 1) W4_RD(fb[1][((m_ - 21) >> 2) & 7], cB1, (((m_ - 21) >> 2) & 7) * 2048);                             \
        else if (DMA) dma_a(boff, ((m_ - 21) >> 2) & 7);                                                                              \
      }                                                                                                                               \
      if (m_ == 54) W4_LGKM0(fb[1]);                                                                                                  \
      if (m_ == 55) __builtin_amdgcn_s_barrier();                                                                                     \
      if ((DMA) && (m_ == 57 || m_ == 59 || m_ == 61)) dma_b(boff, (m_ - 57) >> 1);                                                   \
      W4_SB();                                                                                                                        \
    }                                                                                                                                 \
    _Pragma("unroll") for (int m_ = 0; m_ < 64; ++m_) {                                                                               \
      if (!(DBG & 4)) acc[m_ & 7][m_ >> 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W4_BC(fb[1][m_ >> 3]), W4_BC(fa[1][m_ & 7]), acc[m_ & 7][m_ >> 3], 0, 0, 0); \
      W4_SB();                                                                                                                        \
      if ((DMA) && m_ < 10 && !(m_ & 1)) dma_b(boff, 3 + (m_ >> 1));                                                                  \
      if ((DMA) && m_ == 9 && !(DBG & 2)) { sA += 128; sB += 128; }                                                                   \
      if ((NEXT) && m_ == 10) asm volatile(VMW ::: "memory");                                                                         \
      if ((NEXT) && m_ == 11) __builtin_amdgcn_s_barrier();                                                                           \
      if ((NEXT) && m_ >= 13 && m_ <= 58 && (m_ - 13) % 3 == 0) {                                                                     \
        if ((m_ - 13) / 3 < 8) W4_RD(fa[0][((m_ - 13) / 3) & 7], nA0, (((m_ - 13) / 3) & 7) * 2048);                                  \
        else W4_RD(fb[0][((m_ - 13) / 3 - 8) & 7], nB0, (((m_ - 13) / 3 - 8) & 7) * 2048);                                            \
      }                                                                                                                               \
      W4_SB();                                                                                                                        \
    }                                                                                                                                 \
    if (NEXT) { W4_LGKM0(fa[0]); W4_LGKM0(fb[0]); }                                                                                   \
    W4_SB();                                                                                                                          \
    boff ^= 65536u;                                                                                                                   \
  } while (0)

  // ---- prologue of a tile: k-tiles 0 and 1 -> buffers 0 and 1
  auto stage_first_two = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) { dma_a(0u, p); dma_b(0u, p); }
    sA += 128; sB += 128;
#pragma unroll
    for (int p = 0; p < 8; ++p) { dma_a(65536u, p); dma_b(65536u, p); }
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
    W4_SB();
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_RD(fa[0][i], raA0, i * 2048);
#pragma unroll
    for (int j = 0; j < 8; ++j) W4_RD(fb[0][j], raB0, j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    W4_LGKM0(fa[0]);
    W4_LGKM0(fb[0]);
    unsigned boff = 0u;
    int kt = 0;
    if (nkt > 2) {                                       // k-tile 0: tile 1 is known to have landed
      W4_KTILE(1, "", 1);
      kt = 1;
    }
    for (; kt < nkt - 2; ++kt) W4_KTILE(1, "s_waitcnt vmcnt(16)", 1);
    if (kt < nkt - 1) W4_KTILE(0, "s_waitcnt vmcnt(0)", 1);
    // the last k-tile; behind its barrier nobody reads the operand buffers any more
    const bool more = orig + (int)gridDim.x < nt;
    int m0n = 0, n0n = 0;
    if (more) {
      tile_origin(orig + (int)gridDim.x, m0n, n0n);
      retarget(m0n, n0n);
    }
    // bias of this lane's columns (4 consecutive columns per 16-column block): fetched here and waited for BEFORE the next tile's DMA goes out
    // (a wait for a load issued behind that DMA would wait for the DMA as well: one in-order counter)
    f32x4 pbv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = n0 + wn * 128 + j * 16 + fseg * 4;
      const char* bp = (g.bias && col < g.N) ? reinterpret_cast<const char*>(g.bias + col) : reinterpret_cast<const char*>(g_zero16);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pbv[j]) : "v"(bp) : "memory");
    }
    W4_KTILE(0, "", 0);
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(pbv[0]), "+v"(pbv[1]), "+v"(pbv[2]), "+v"(pbv[3]), "+v"(pbv[4]), "+v"(pbv[5]), "+v"(pbv[6]), "+v"(pbv[7])
                 :
                 : "memory");
    if (more) stage_first_two();
    // ---- epilogue: lane holds C[m0 + wm*128 + i*16 + (lane & 15)][n0 + wn*128 + j*16 + (lane >> 4)*4 .. + 3]
    {
      float* C = g.C;
      const long long ldc = g.ldc;
      const bool stream_out = (g.nt_store & 1) != 0;
      // (addresses from an opaque copy of the lane id made HERE: computed from `lane` they are hoisted above the main loop and spilled)
      int el = lane;
      asm volatile("" : "+v"(el));
      const int frow = el & 15, fseg = el >> 4;
      if constexpr (OBF) {
        __bf16* Cb = reinterpret_cast<__bf16*>(C);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = m0 + wm * 128 + i * 16 + frow;
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x4 va = acc[i][j] + pbv[j], vb = acc[i][j + 1] + pbv[j + 1];
            const bf16x4w pa = {(__bf16)va[0], (__bf16)va[1], (__bf16)va[2], (__bf16)va[3]};
            const bf16x4w pb = {(__bf16)vb[0], (__bf16)vb[1], (__bf16)vb[2], (__bf16)vb[3]};
            const f32x2w a2 = __builtin_bit_cast(f32x2w, pa), b2 = __builtin_bit_cast(f32x2w, pb);
            const u32pair s0 = permlane16_swap(a2[0], b2[0]), s1 = permlane16_swap(a2[1], b2[1]);
            const int col = n0 + wn * 128 + (j + (fseg & 1)) * 16 + (fseg >> 1) * 8;
            const f32x4 v = {s0.a, s1.a, s0.b, s1.b};
            f32x4* pc = (row < g.M && col < g.N) ? reinterpret_cast<f32x4*>(Cb + (long long)row * ldc + col) : reinterpret_cast<f32x4*>(g_sink16);
            if (stream_out) __builtin_nontemporal_store(v, pc);
            else *pc = v;
          }
        }
      } else
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + wm * 128 + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = n0 + wn * 128 + j * 16 + fseg * 4;
          const f32x4 v = acc[i][j] + pbv[j];
          f32x4* pc = (row < g.M && col < g.N) ? reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) : reinterpret_cast<f32x4*>(g_sink16);
          if (stream_out) __builtin_nontemporal_store(v, pc);
          else *pc = v;
        }
      }
    }
    if (!more) break;
    orig += (int)gridDim.x; m0 = m0n; n0 = n0n;
    after_epilogue = true;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef W4_BC
#undef W4_SB
#undef W4_RD
#undef W4_LGKM0
#undef W4_KTILE
}
