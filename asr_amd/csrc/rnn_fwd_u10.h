// 10-UNIT-SLICE split forward recurrence (fp32 mode) — textually included by rnn.hip inside its anonymous namespace.
//
// The split form of the persistent forward kernel (rnn_fwd_persistent_kernel<.., SP = true>) keeps a workgroup's W_hh slice in registers as
// a hi and a lo bf16 fragment set.  With 16-unit slices that is 2 planes x G column tiles x H / 256 chunks per wave: 160 registers per lane
// for the LSTM of BASELINE C4 (G = 4, H = 1280) before a single operand is gathered — it does not fit next to the 32-row tile B = 32
// needs, and 16-row tiles would take 320 workgroups.  So that shape ran one launch per time step (19.6 us per step, 96 of C4's 255 ms).
//
// Here a workgroup owns TEN hidden units: G x 10 = 40 gate columns in three 16-column MFMA tiles (column c = gate c / 10, unit c % 10; the last
// eight columns are zero padding), 128 slices x 2 directions = exactly one workgroup per CU at H = 1280, 120 registers of W_hh fragments,
// and the operand is gathered plane by plane (lo first: smallest terms first, and 40 instead of 80 registers of gathered data alive).
// Everything else is the split kernel's protocol: four exchange buffers of two bf16 planes, consumers poll 16-byte lane vectors (8 units
// of one row) until none of their dwords is the sentinel, a publisher resets its own part of the buffer two steps ahead.  What changes on
// the publishing side is the granularity: ten units are five DWORDS per row and plane (a slice starts at an even unit, so a dword — two
// units — always has one owner), stored as five 4-byte stores where the 16-unit kernels store whole vectors; a 16-byte vector then has
// up to two owners and a consumer simply keeps polling until both have written.
// Results: fp32-grade (hi.hi + lo.hi + hi.lo, fp32 accumulation, fp32 state and gate math) — within ~1e-6 of the fp32 kernels, not
// bit-identical to them (tests/test_gpu_round4.py::test_u10_split_forward_recurrence_vs_fp64_and_fp32_kernels).
__device__ __forceinline__ void store4_sc1(void* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory"); }

constexpr int U10 = 10;
__host__ __device__ constexpr int u10_tiles(int G) { return (G * U10 + 15) / 16; }
// shapes the packed operand is produced for (independent of the batch): whole 32-unit chunks, whole 10-unit slices, at most one workgroup
// per CU for both directions of one batch tile, at most five chunks per wave
inline bool u10_shape_ok(int H) { return H >= 160 && (H % 160) == 0 && H <= 1280; }

// packed operand: [plane hi | plane lo][dir][slice H/10][tile][chunk H/32][lane 64] x 16 bytes (8 bf16: column (lane & 15) of the tile, k = 8 (lane >> 4) ..)
__global__ __launch_bounds__(256) void rnn_pack_u10_kernel(const float* __restrict__ whh, void* __restrict__ wp_hi, void* __restrict__ wp_lo, int G, int H) {
  const int gs = H / U10, nch = H / 32, ntl = u10_tiles(G);
  const long long nf = (long long)2 * gs * ntl * nch * 64;
  for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < nf; ii += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(ii & 63);
    long long r = ii >> 6;
    const int c = r % nch; r /= nch;
    const int tl = r % ntl; r /= ntl;
    const int slice = r % gs, dir = r / gs;
    const int col = tl * 16 + (lane & 15), k0 = c * 32 + (lane >> 4) * 8;
    const bool on = col < G * U10;
    const int g = col / U10, j = slice * U10 + col % U10;
    const float* src = whh + ((long long)dir * G * H + (on ? g * H + j : 0)) * H + k0;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = on ? src[e] : 0.f;
      hi[e] = (__bf16)v;
      lo[e] = (__bf16)(v - (float)hi[e]);
    }
    reinterpret_cast<bf16x8*>(wp_hi)[ii] = hi;
    reinterpret_cast<bf16x8*>(wp_lo)[ii] = lo;
  }
}
size_t u10_plane_bytes(int G, int H) { return (size_t)2 * (H / U10) * u10_tiles(G) * (H / 32) * 1024; }

template <int G, int MB, int NCW>
__global__ __launch_bounds__(NW * 64) void rnn_fwd_u10_kernel(RnnArgs a, char* xbuf, unsigned* census, int spin_limit) {
  constexpr int NTL = u10_tiles(G), PAIRS = MB * 16 * U10;
  static_assert(PAIRS <= NW * 64, "one (row, unit) pair per thread");
  static_assert(NCW * MB <= 12, "one poll statement per plane");
  __shared__ __attribute__((aligned(16))) f32x4 red[2][NW][MB * NTL][64];       // double-buffered: ONE workgroup barrier per time step
  const PRole role = persist_role(a, census, spin_limit, 1);
  if (!role.active) return;
  const int dir = role.dir, bt = role.bt, slice = role.slice;
  const int T = a.T, B = a.B, H = a.H;
  const int nch = H >> 5, gs = a.p_gs;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long planebytes = (long long)2 * a.nbt16 * nch * 1024;          // one packed h plane: [dir][tile][chunk][64 lanes][16 B]
  const long long bufbytes = 2 * planebytes;                                 // one exchange buffer: hi plane, lo plane
  const long long dirbase = (long long)dir * a.nbt16 * nch * 1024;

  // ---- W_hh slice -> registers (once): hi and lo fragment sets
  f32x4 wreg[2][NCW][NTL];
  bool cval[NCW];
  const long long wplane = (long long)2 * gs * NTL * nch * 256;              // floats of one plane
#pragma unroll
  for (int k = 0; k < NCW; ++k) {
    const int c = wave + NW * k;
    cval[k] = c < nch;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl)
        wreg[pl][k][tl] = cval[k] ? *reinterpret_cast<const f32x4*>(a.wp + pl * wplane + (((((long long)dir * gs + slice) * NTL + tl) * nch + c) * 256) + lane * 4)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- this thread's (batch row, hidden unit) pair: fixed for the whole layer
  const int q = threadIdx.x;
  const int row = q / U10, u = q - row * U10;
  const bool ppub = q < PAIRS;                                               // rows beyond B publish zeros: consumers wait for every vector
  const int mbi = row >> 4, r16 = row & 15, reg = r16 & 3;
  const int b = bt * MB * 16 + row, j = slice * U10 + u;
  const bool pact = ppub && b < B;
  const int plen = pact ? a.lens[b] : 0;
  int src_t[G], src_lane[G];
  float pb[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int col = g * U10 + u;                                             // this pair's column of gate g: tile col >> 4, column col & 15
    src_t[g] = mbi * NTL + (col >> 4);
    src_lane[g] = (r16 >> 2) * 16 + (col & 15);                              // 16x16 MFMA result: lane (row / 4) * 16 + column, register row % 4
    pb[g] = pact ? a.bhh[(dir * G + g) * H + j] : 0.f;
  }
  float pprev = 0.f;                                                         // h_{t-1} (GRU) / c_{t-1} (LSTM)
  // publish: the thread of an EVEN unit stores the dword (units j, j + 1) of its row: vector = units 8 (ju >> 3) .. + 7 of chunk j >> 5
  const bool pub_thr = ppub && !(u & 1);
  long long pub_off;
  {
    const int ju = j & 31;
    pub_off = dirbase + ((((long long)(bt * MB + mbi) * nch + (j >> 5)) * 64) + (ju >> 3) * 16 + r16) * 16 + ((ju & 7) >> 1) * 4;
  }
  const int t0 = dir == 0 ? 0 : T - 1;
  const long long dH = (dir == 0 ? 1LL : -1LL) * B * 2 * H, dG = dH * G;
  long long eH = (((long long)t0 * B + b) * 2 + dir) * H + j, eG = (((long long)t0 * B + b) * 2 + dir) * G * H + j;
  float pgx[G], pgx_next[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { pgx[g] = 0.f; pgx_next[g] = 0.f; }
  if (pact) {
#pragma unroll
    for (int g = 0; g < G; ++g) pgx[g] = ldnt(a.gx + eG + g * H);
  }

  // this wave's vectors of one plane of the packed exchange buffer (byte offsets from the buffer's direction base)
  unsigned goff[NCW * MB], pend0 = 0;
#pragma unroll
  for (int k = 0; k < NCW; ++k)
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      goff[k * MB + i] = (unsigned)(((((long long)(bt * MB + i) * nch + (wave + NW * k)) * 64) + lane) * 16);
      if (cval[k]) pend0 |= 1u << (k * MB + i);
    }

  vm_drained();
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? s : T - 1 - s;
    f32x4 acc[MB][NTL];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl) acc[i][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const char* xin = xbuf + (long long)((s - 1) & 3) * bufbytes + dirbase;
      // plane 1 (lo) first, then plane 0 (hi): lo.W_hi, then hi.W_lo and hi.W_hi
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const char* base = xin + (ph == 0 ? planebytes : 0);
        u32x4_ av[NCW * MB];
#pragma unroll
        for (int k = 0; k < NCW * MB; ++k) av[k] = u32x4_{0u, 0u, 0u, 0u};
        int spins = 0;
        unsigned pend = pend0;
        while (pend) {
          poll_pass<NCW * MB>(av, goff, base, pend);
#pragma unroll
          for (int k = 0; k < NCW * MB; ++k)
            if (pend & (1u << k)) {
              const u32x4_ c = av[k];
              const bool ok = c.x != PSENT && c.y != PSENT && c.z != PSENT && c.w != PSENT;
              if (__ballot(ok) == ~0ull) pend &= ~(1u << k);
            }
          pend = __builtin_amdgcn_readfirstlane(pend);
          if (pend && ++spins > spin_limit) {
            if (lane == 0 && atomicCAS(&a.status[0], 0, 1) == 0) {
              a.status[1] = slice; a.status[2] = bt; a.status[3] = dir; a.status[4] = s; a.status[5] = wave;
              a.status[6] = (int)pend; a.status[7] = 0;
              __threadfence_system();
            }
            return;
          }
        }
#pragma unroll
        for (int k = 0; k < NCW; ++k)
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, av[k * MB + i]);
#pragma unroll
            for (int tl = 0; tl < NTL; ++tl) {
              if (ph == 0) {
                acc[i][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, __builtin_bit_cast(bf16x8, wreg[0][k][tl]), acc[i][tl], 0, 0, 0);
              } else {
                acc[i][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, __builtin_bit_cast(bf16x8, wreg[1][k][tl]), acc[i][tl], 0, 0, 0);
                acc[i][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, __builtin_bit_cast(bf16x8, wreg[0][k][tl]), acc[i][tl], 0, 0, 0);
              }
            }
          }
      }
    }
    vm_drained();                                       // (the gather has waited for everything; tell the compiler)
    if (s > 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) pgx[g] = pgx_next[g];  // x-projections of THIS step: loaded one step ago
    }
    const bool more = s + 1 < T;
    if (more && pact) {
#pragma unroll
      for (int g = 0; g < G; ++g) pgx_next[g] = ldnt(a.gx + eG + dG + g * H);
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl) red[s & 1][wave][i * NTL + tl][lane] = acc[i][tl];
    __syncthreads();

    // ---- gate math (the step kernel's epilogue)
    float out_g[4] = {0.f, 0.f, 0.f, 0.f}, out_aux = 0.f, hnew = 0.f;
    const bool live = pact && t < plen;
    if (live) {
      float gh[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[s & 1][w][src_t[g]][src_lane[g]][reg];
        gh[g] = sum + pb[g];
      }
      if constexpr (G == 3) {
        const float r = sigmoidf_(pgx[0] + gh[0]);
        const float z = sigmoidf_(pgx[1] + gh[1]);
        const float n = tanhf_(__builtin_fmaf(r, gh[2], pgx[2]));
        out_g[0] = r; out_g[1] = z; out_g[2] = n; out_g[3] = gh[2];
        out_aux = gh[2];
        hnew = __builtin_fmaf(z, pprev, (1.f - z) * n);
        pprev = hnew;
      } else {
        const float ig = sigmoidf_(pgx[0] + gh[0]);
        const float fg = sigmoidf_(pgx[1] + gh[1]);
        const float gg = tanhf_(pgx[2] + gh[2]);
        const float og = sigmoidf_(pgx[G - 1] + gh[G - 1]);
        const float c = __builtin_fmaf(fg, pprev, ig * gg);
        out_g[0] = ig; out_g[1] = fg; out_g[2] = gg; out_g[3] = og;
        out_aux = c;
        hnew = og * tanhf_(c);
        pprev = c;
      }
    } else {
      pprev = 0.f;                                      // beyond the sample's length: state is zero
    }
    // ---- publish h_s: the neighbour's value comes over the wave (a pair of units never straddles two waves: pairs start at even threads),
    // one dword per plane; then the reset of the same dwords two buffers ahead (safety argument: rnn_fwd_persistent_kernel)
    {
      const float h1 = __shfl_down(hnew, 1, 64);
      if (pub_thr) {
        const __bf16 a0 = (__bf16)hnew, a1 = (__bf16)h1;
        const __bf16 l0 = (__bf16)(hnew - (float)a0), l1 = (__bf16)(h1 - (float)a1);
        const unsigned dhi = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
        const unsigned dlo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        char* cur = xbuf + (long long)(s & 3) * bufbytes + pub_off;
        char* nxt = xbuf + (long long)((s + 2) & 3) * bufbytes + pub_off;
        store4_sc1(cur, dhi);
        store4_sc1(cur + planebytes, dlo);
        store4_sc1(nxt, PSENT);
        store4_sc1(nxt + planebytes, PSENT);
      }
    }
    // ---- the step's saved-for-backward outputs: last, in the shadow of the exchange (fp32 mode: activated gates in gx, aux, h)
    if (pact) {
      float* gx = a.gx + eG;
      stnt(&gx[0], out_g[0]); stnt(&gx[H], out_g[1]); stnt(&gx[2 * H], out_g[2]);
      if (G == 4) { stnt(&gx[3 * H], out_g[3]); a.aux[eH] = out_aux; }
      else stnt(a.aux + eH, live ? out_aux : 0.f);
      a.hbuf[eH] = hnew;
    }
    eH += dH; eG += dG;
  }
}

// 1 = launched, 0 = not eligible, < 0 = error.  a.wp = the hi plane of the 10-unit packed operand (lo plane behind it).
template <int G>
int try_launch_fwd_u10(RnnArgs a, hipStream_t st) {
  static const char* env = getenv("DS2_RNN_PERSISTENT");
  if (env && env[0] == '0') return 0;
  if (a.dbg & ~(128 | 256)) return 0;
  if (!u10_shape_ok(a.H) || a.T < 2 || a.gates_bf || a.h_bf) return 0;
  const int mb = a.B > 16 ? 2 : 1, nbt = ceil_div(a.B, 16 * mb), gs = a.H / U10, nch = a.H / 32, ncw = ceil_div(nch, NW);
  if (ncw > 5 || (long long)gs * nbt * 2 > cu_count()) return 0;             // every workgroup resident at once: one per CU
  a.nsl = a.H / 16;
  a.nbt16 = ceil_div(a.B, 32) * 2;
  a.p_nbt = nbt; a.p_gs = gs; a.p_cux = CUS_PER_XCD; a.p_census = 0;         // 128 slices never fit one XCD: placement-independent protocol
  char* xbuf = reinterpret_cast<char*>(a.pk);
  const size_t xbytes = 4 * 2 * fwd_xbuf_bytes(a.B, a.H, 1);
  if (!a.prearmed) DS2_HIP(hipMemsetAsync(xbuf, 0xff, xbytes + CENSUS_BYTES, st));
  unsigned* census = reinterpret_cast<unsigned*>(xbuf + xbytes);
  dim3 grid(gs * nbt * 2), block(NW * 64);
  static const char* sl = getenv("DS2_RNN_SPIN_LIMIT");
  const int spin_limit = sl ? atoi(sl) : (1 << 20);
#define DS2_U10(NCW_)                                                                                                        \
  case NCW_:                                                                                                                 \
    if (mb == 2) hipLaunchKernelGGL((rnn_fwd_u10_kernel<G, 2, NCW_>), grid, block, 0, st, a, xbuf, census, spin_limit);       \
    else hipLaunchKernelGGL((rnn_fwd_u10_kernel<G, 1, NCW_>), grid, block, 0, st, a, xbuf, census, spin_limit);               \
    break;
  switch (ncw) {
    DS2_U10(1) DS2_U10(2) DS2_U10(3) DS2_U10(4) DS2_U10(5)
    default: return 0;
  }
#undef DS2_U10
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn 10-unit forward launch failed: %s", hipGetErrorString(e));
  return 1;
}
