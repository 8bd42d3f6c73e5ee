// Grouped split-K TN bf16 GEMM (the weight-gradient products of one recurrent layer in one launch), FOUR waves x 128 x 128 per 256 x 256 x 64
// k-tile, accumulators in AGPRs: the four-wave schedule of gemm_nt_w4.h on the TN kernel's K-row-major LDS image.   (included by gemm_bf16.hip)
//
//   item        (problem, K slice, 256 x 256 tile, row-major) as gemm_bf16_tn_glds_kernel<true>, but handed to the workgroups in XCD-sized runs of
//               32 consecutive items (below); every k-tile of every slice is full
//               (the launcher takes this kernel only when every K is a multiple of 64)
//   LDS         [2 buffers][A | B], an operand tile = 64 k-rows x 512 B (256 columns); one DMA instruction lands 2 k-rows (1 KiB): wave w moves
//               pieces w, w + 4, ..., w + 28 of A and of B
//   fragments   a 16-column x 32-k MFMA operand (lane l: column l & 15, k = 8 (l >> 4) .. + 7) is TWO ds_read_b64_tr_b16: inside a 16-lane group
//               lane p addresses row (p >> 2) of a 4-row block at columns 4 (p & 3) .. + 3 and receives rows 0..3 of column p; the group l >> 4
//               reads the rows 8 (l >> 4) + (0..3) and, 2048 B further, + (4..7).  A 32-lane read group therefore touches 8 rows
//               {q, 8 + q} x 32 B: the 16-byte slot s of row r is stored at s ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1)) (on the DMA's source
//               address and on the read), which puts those 8 x 32 B on all 64 banks once
//   schedule    as gemm_nt_w4.h (a fragment read is two instructions here)
// Product TRANSPOSED (B fragment first): a lane holds four consecutive columns of one C row -> 16-byte stores straight from the registers.
// Same products in the same order as the 8-wave kernel: bit-identical slabs.
template <int DBG = 0>
__global__ __launch_bounds__(256) void gemm_bf16_tn_w4_kernel(TnSGroup grp) {
  extern __shared__ __attribute__((aligned(1024))) char ldsg[];
  const __bf16* A;
  const __bf16* B;
  int pM, pN, plda, pldb, zs, orig, nt, kchunk, ntx, pK;
  float* Cfinal; float* Cslab; long long ldcf;
  bool partial;
  int* tick; int nslab_all; const float* slabs;          // fused reduce (TnSProb::tick): this tile's arrival counter, slabs per tile, slab 0
  {
    // blockIdx.x -> item: workgroup ids go round the 8 XCDs, one workgroup per CU, so the 32 workgroups an XCD holds at a time are the ids
    // 256 c + x + 8 j.  They take 32 CONSECUTIVE items = (with at most 8 column tiles) a few whole tile rows of one K slice, which share their
    // A / B panels through that XCD's L2: 12 panels for 64 panel reads at 4 column tiles.
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
    A = grp.p[0].A; B = grp.p[0].B; Cfinal = grp.p[0].C; Cslab = grp.p[0].partial; tick = grp.p[0].tick; nslab_all = grp.p[0].nslab_all;
    pM = grp.p[0].M; pN = grp.p[0].N; pK = grp.p[0].K; plda = grp.p[0].lda; pldb = grp.p[0].ldb; ldcf = grp.p[0].ldc; ntx = grp.p[0].ntx;
    nt = grp.p[0].ntiles; kchunk = grp.p[0].kchunk;
    int first = 0, slab0 = grp.p[0].slab0, to_slab = grp.p[0].to_slab;
#pragma unroll
    for (int k = 1; k < TN_MAX_PROBLEMS; ++k)
      if (k < grp.nprob && item >= grp.p[k].first_item) {
        A = grp.p[k].A; B = grp.p[k].B; Cfinal = grp.p[k].C; Cslab = grp.p[k].partial; tick = grp.p[k].tick; nslab_all = grp.p[k].nslab_all;
        pM = grp.p[k].M; pN = grp.p[k].N; pK = grp.p[k].K; plda = grp.p[k].lda; pldb = grp.p[k].ldb; ldcf = grp.p[k].ldc; ntx = grp.p[k].ntx;
        nt = grp.p[k].ntiles; kchunk = grp.p[k].kchunk; first = grp.p[k].first_item; slab0 = grp.p[k].slab0; to_slab = grp.p[k].to_slab;
      }
    const int local = item - first;
    zs = local / nt; orig = local % nt;
    partial = to_slab != 0;
    slabs = Cslab;
    Cslab += (long long)(slab0 + zs) * pM * pN;
  }
  int tile = orig;
  if (!grp.order) {                                      // DS2_TN_ORDER=0 (A/B): each XCD a contiguous run of every slice's tiles
    const int xcd = orig & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int m0 = (tile / ntx) * 256, n0 = (tile % ntx) * 256;
  const int kbeg = zs * kchunk;
  const int kend = min(pK, kbeg + kchunk);
  const int nkt = (kend - kbeg) >> 6;                    // full k-tiles only, nkt >= 2
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ldsg;

  // ---- DMA: piece p = wave + 4 i holds k-rows 2p, 2p + 1; lane -> (k-row, physical 16-byte slot); columns past M / N are clamped to the last
  // valid segment (they only feed C rows / columns that are never stored)
  unsigned voffA[8], voffB[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (wave + 4 * i) * 2 + (lane >> 5);
    const int gs = (lane & 31) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
    voffA[i] = (unsigned)(r * plda * 2 + min(m0 + gs * 8, pM - 8) * 2);
    voffB[i] = (unsigned)(r * pldb * 2 + min(n0 + gs * 8, pN - 8) * 2);
  }
  const char* sA = reinterpret_cast<const char*>(A + (long long)kbeg * plda);       // uniform: first k-row of the NEXT k-tile to stage
  const char* sB = reinterpret_cast<const char*>(B + (long long)kbeg * pldb);
  const long long stepA = (long long)plda * 128, stepB = (long long)pldb * 128;      // 64 k-rows
  const unsigned dstw = lds0 + wave * 1024;
  auto dma_a = [&](unsigned boff, int p) { if (!(DBG & 1)) w4_dma(voffA[p], sA, dstw + boff + p * 4096); };
  auto dma_b = [&](unsigned boff, int p) { if (!(DBG & 1)) w4_dma(voffB[p], sB, dstw + boff + 32768 + p * 4096); };

  // ---- fragment read addresses (buffer 0, k-step 0, first of the two reads); block i of 16 columns: va[i]
  const int p16 = lane & 15, q4 = p16 >> 2, g4 = lane >> 4;
  const unsigned rowpart = (unsigned)((g4 * 8 + q4) * 512);
  const unsigned xv = (unsigned)((q4 << 2) | ((g4 & 1) << 1));
  unsigned va[8], vb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ca = wm * 128 + i * 16 + 4 * (p16 & 3), cb = wn * 128 + i * 16 + 4 * (p16 & 3);
    va[i] = lds0 + rowpart + ((((ca >> 3) ^ xv) << 4) | (((ca >> 2) & 1) << 3));
    vb[i] = lds0 + 32768 + rowpart + ((((cb >> 3) ^ xv) << 4) | (((cb >> 2) & 1) << 3));
  }

  f32x4 acc[8][8];
  f32x4 fa[2][8], fb[2][8];
#define TW_BC(x) __builtin_bit_cast(bf16x8, x)
#define TW_SB() __builtin_amdgcn_sched_barrier(0)
  // one MFMA operand = two transposing reads, each landing in one half of the 4-register tuple (joined right behind the reads: a register-tuple
  // definition, no instruction — scripts/check_isa.py looks for copies between a read and its wait)
#define TW_RD(dst, addr, off)                                                                                                         \
  do {                                                                                                                                \
    f32x2 lo_, hi_;                                                                                                                   \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                        \
                 : "=&v"(lo_), "=&v"(hi_)                                                                                             \
                 : "v"(addr), "n"(off), "n"((off) + 2048));                                                                           \
    dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);                                                                              \
  } while (0)
#define TW_LGKM0(F)                                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                 \
               : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])                       \
               :                                                                                                                      \
               : "memory")
  // one k-tile (see gemm_nt_w4.h).  The read addresses va / vb stand on buffer cur; each is flipped to the other buffer right in front of its
  // read of tile kt + 1, i.e. after its last use on cur.
#define TW_KTILE(DMA, VMW, NEXT)                                                                                                      \
  do {                                                                                                                                \
    TW_SB();                                                                                                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 64; ++m_) {                                                                               \
      if (!(DBG & 4)) acc[m_ & 7][m_ >> 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(TW_BC(fb[0][m_ >> 3]), TW_BC(fa[0][m_ & 7]), acc[m_ & 7][m_ >> 3], 0, 0, 0); \
      TW_SB();                                                                                                                        \
      if (m_ < 16 && !(m_ & 1)) TW_RD(fa[1][(m_ >> 1) & 7], va[(m_ >> 1) & 7], 16384);                                                \
      if (m_ == 19) TW_LGKM0(fa[1]);                                                                                                  \
      if (m_ == 20) __builtin_amdgcn_s_barrier();                                                                                     \
      if (m_ >= 21 && m_ <= 51 && (m_ & 1)) {                                                                                         \
        if (((m_ - 21) >> 1) & 1) TW_RD(fb[1][((m_ - 21) >> 2) & 7], vb[((m_ - 21) >> 2) & 7], 16384);                                \
        else if (DMA) dma_a(boff, ((m_ - 21) >> 2) & 7);                                                                              \
      }                                                                                                                               \
      if (m_ == 54) TW_LGKM0(fb[1]);                                                                                                  \
      if (m_ == 55) __builtin_amdgcn_s_barrier();                                                                                     \
      if ((DMA) && (m_ == 57 || m_ == 59 || m_ == 61)) dma_b(boff, (m_ - 57) >> 1);                                                   \
      TW_SB();                                                                                                                        \
    }                                                                                                                                 \
    _Pragma("unroll") for (int m_ = 0; m_ < 64; ++m_) {                                                                               \
      if (!(DBG & 4)) acc[m_ & 7][m_ >> 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(TW_BC(fb[1][m_ >> 3]), TW_BC(fa[1][m_ & 7]), acc[m_ & 7][m_ >> 3], 0, 0, 0); \
      TW_SB();                                                                                                                        \
      if ((DMA) && m_ < 10 && !(m_ & 1)) dma_b(boff, 3 + (m_ >> 1));                                                                  \
      if ((DMA) && m_ == 9) { sA += stepA; sB += stepB; }                                                                             \
      if ((NEXT) && m_ == 10) asm volatile(VMW ::: "memory");                                                                         \
      if ((NEXT) && m_ == 11) __builtin_amdgcn_s_barrier();                                                                           \
      if (m_ >= 13 && m_ <= 58 && (m_ - 13) % 3 == 0) {                                                                               \
        if ((m_ - 13) / 3 < 8) {                                                                                                      \
          va[((m_ - 13) / 3) & 7] ^= 65536u;                                                                                          \
          if (NEXT) TW_RD(fa[0][((m_ - 13) / 3) & 7], va[((m_ - 13) / 3) & 7], 0);                                                    \
        } else {                                                                                                                      \
          vb[((m_ - 13) / 3 - 8) & 7] ^= 65536u;                                                                                      \
          if (NEXT) TW_RD(fb[0][((m_ - 13) / 3 - 8) & 7], vb[((m_ - 13) / 3 - 8) & 7], 0);                                            \
        }                                                                                                                             \
      }                                                                                                                               \
      TW_SB();                                                                                                                        \
    }                                                                                                                                 \
    if (NEXT) { TW_LGKM0(fa[0]); TW_LGKM0(fb[0]); }                                                                                   \
    TW_SB();                                                                                                                          \
    boff ^= 65536u;                                                                                                                   \
  } while (0)

  // ---- prologue: k-tiles 0 and 1 -> buffers 0 and 1
#pragma unroll
  for (int p = 0; p < 8; ++p) { dma_a(0u, p); dma_b(0u, p); }
  sA += stepA; sB += stepB;
#pragma unroll
  for (int p = 0; p < 8; ++p) { dma_a(65536u, p); dma_b(65536u, p); }
  sA += stepA; sB += stepB;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  TW_SB();
#pragma unroll
  for (int i = 0; i < 8; ++i) TW_RD(fa[0][i], va[i], 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) TW_RD(fb[0][j], vb[j], 0);
  TW_LGKM0(fa[0]);
  TW_LGKM0(fb[0]);
  unsigned boff = 0u;
  int kt = 0;
  if (nkt > 2) {                                         // k-tile 0: tile 1 has landed with the prologue
    TW_KTILE(1, "", 1);
    kt = 1;
  }
  for (; kt < nkt - 2; ++kt) TW_KTILE(1, "s_waitcnt vmcnt(16)", 1);
  if (kt < nkt - 1) TW_KTILE(0, "s_waitcnt vmcnt(0)", 1);
  TW_KTILE(0, "", 0);

  // ---- epilogue: lane holds C[m0 + wm*128 + i*16 + (lane & 15)][n0 + wn*128 + j*16 + (lane >> 4)*4 .. + 3]   (N % 8 == 0)
  {
    float* C = partial ? Cslab : Cfinal;
    const long long ldc = partial ? (long long)pN : ldcf;
    int el = lane;
    asm volatile("" : "+v"(el));
    const int frow = el & 15, fseg = el >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wm * 128 + i * 16 + frow;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = n0 + wn * 128 + j * 16 + fseg * 4;
        if (row < pM && col < pN) *reinterpret_cast<f32x4*>(C + (long long)row * ldc + col) = acc[i][j];
      }
    }
    // ---- fused reduce: whoever delivers the LAST slab of this tile adds all of them (slab 0 first, as splitk_reduce_group_kernel does) and
    // writes C.  Release: every thread's slab stores are made visible at device scope before the workgroup's arrival is counted; acquire: the
    // last arriver invalidates its view before it reads the other workgroups' slabs (they may sit on another XCD).
    if (partial && tick) {
      __shared__ int s_last;
      // (release ONLY here — write back, no invalidate: a full fence by each of the 768 workgroups drops the operand panels its XCD's L2
      // holds for the neighbours that are still multiplying, measured +0.22 ms per launch; the acquire is the last arriver's alone)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(tick + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nslab_all - 1;
      __syncthreads();
      if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const long long slab = (long long)pM * pN;
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
          const int row = m0 + wm * 128 + i * 16 + frow;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int col = n0 + wn * 128 + j * 16 + fseg * 4;
            if (row < pM && col < pN) {
              const float* sp = slabs + (long long)row * pN + col;
              f32x4 sum = *reinterpret_cast<const f32x4*>(sp);
              for (int k = 1; k < nslab_all; ++k) sum += *reinterpret_cast<const f32x4*>(sp + k * slab);
              *reinterpret_cast<f32x4*>(Cfinal + (long long)row * ldcf + col) = sum;
            }
          }
        }
      }
    }
  }
#undef TW_BC
#undef TW_SB
#undef TW_RD
#undef TW_LGKM0
#undef TW_KTILE
}
