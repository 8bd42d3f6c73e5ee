// K-SPLIT persistent backward recurrence (bf16 training path) — textually included by rnn.hip inside its anonymous namespace.
//
// The all-gather form (rnn_bwd_persistent_kernel) gives a workgroup 32 OUTPUT units: it must read all of dGh_{t+1} of its 16 batch rows
// (16 x G*H bf16 = 96 KB at H = 1024) every time step, and that read — 32 CUs x 96 KB out of one XCD's L2 — is what the step costs
// (profiles/r02_probe_persist_timeline.txt: 1.85-2.3 of 3.3 us).  Here the product  dh_partial = dGh W_hh  is split along K instead:
//
//   * a workgroup owns 32 hidden units u of one (direction, 16-row batch tile): it does their gate-derivative math, so the G*32 columns
//     k = g*H + u of dGh are born in its own registers;
//   * it multiplies THOSE columns by the matching G*32 ROWS of W_hh (all H output columns; the same 192 KB of bf16 fragments per
//     workgroup as before, resident in registers) — 16 x H partial sums of dh over its K slice — and publishes them as bf16;
//   * the owner of 32 output units gathers the 16 x 32 partials of all H/32 producers (32 KB instead of 96 KB) and adds them in a fixed
//     order in fp32.
// Per CU and step 32 KB are read and 32 KB written (no reset traffic, see the tag below) against 96 KB + 6 KB before.  The sum over K is
// now an fp32 sum of bf16-rounded partial sums: results differ from the step kernels' by one more bf16 rounding per partial (stated
// tolerance in tests/test_gpu_kernels.py), they are still run-to-run bit-identical (fixed summation tree).
//
// What a step costs here is (a) the exchange latency and (b) INSTRUCTIONS: two waves share a SIMD, so every VALU instruction of the
// dependent chain gather -> gate math -> MFMA -> publish costs 8 cycles of it, and all eight waves share ONE address unit that spends
// ~16 cycles on a vector-memory wave-instruction whatever its width (160 such instructions per step in the first version = 1.1 us).  Hence:
// every exchange access is 16 bytes per lane; results leave through LDS as 16-byte stores by four waves (4 instructions per step instead of
// 40); addresses are stepped, not recomputed.
//
// Exchange format.  One 128-byte LINE per (consumer workgroup j', consumer wave w, producer p) = the partials of rows 2w, 2w+1 x the
// consumer's 32 units: piece (h, q) of 16 bytes = row 2w + h, units [4q..4q+3 | 16+4q..16+4q+3] as bf16.  A consumer wave's lines of all
// producers are contiguous (gs x 128 B = 4 KB at H = 1024: four 1 KB wave loads), and a line has exactly ONE reader.  The product is
// computed TRANSPOSED (A = W_hh fragment, B = dGh fragment), so that a lane of the producer holds 4 consecutive units of one batch row per
// 16-column tile: its 16-byte piece is two tiles' registers, one `global_store_dwordx4` per tile pair and wave covers eight lines.
// The payload is its own flag without any reset: bit 0 of every 8-byte half of a piece (the last mantissa bit of its first value, which
// stays part of the value: +-1 bf16 ulp, sign alternating with the tag) carries a TAG = (step >> 1) & 1, two slots are used alternately
// (slot = step & 1), so what a slot holds before step s lands has the opposite tag (the launcher fills both slots with 0xff: tag 1 before
// steps 0 / 1).  Slot reuse is safe: a producer publishes step s only after it has gathered step s-1 from every member of its group, and a
// member publishes step s-1 only after all of its waves have read step s-2.
//
// The gather needs no LDS and no barrier: lane l of consumer wave w reads piece (l & 7) of producer 8i + (l >> 3), i = 0..gs/8-1, sums over i
// in registers (8 fp32 sums) and a 3-stage reduce-scatter over lane bits 5, 4, 3 (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8)
// leaves every lane with the complete sum of ONE (row, unit) pair — the pair whose gate math it then does.  One workgroup barrier per step
// remains: the 16 x G*32 bf16 tile of dGh goes through LDS so that every wave can take it as its MFMA operand (and leave through it).
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
// TR: the GRU TRAINING instance (packed bf16 gate records in, bf16 dGx out — what engine.py's bf16 train step passes): the generic operand
// fetch is compiled out, see ASM_FETCH below.
// SP: the SPLIT form of the fp32 mode (DS2_F32_RNN=split; plain fp32 buffers in and out): W_hh as hi and lo bf16 fragment sets, the dGh tile
// as a hi and a lo LDS plane, the product as lo.hi + hi.lo + hi.hi, and every partial sum published as TWO tagged bf16 pieces in two planes of
// the exchange slots: hi' = the tagged bf16 of the fp32 partial (as in the bf16 form) and lo = bf16(partial - hi') — the lo piece absorbs
// both the bf16 rounding AND the tag bit of the hi piece, so a consumer's hi' + lo is the fp32 partial to 2^-17 and the tag costs nothing.
// The gather reads both planes (2 NL loads per lane, one poll statement), everything behind it is the same reduce-scatter.
template <int G, int NT, bool TR = false, bool SP = false>
__global__ __launch_bounds__(NW * 64) void rnn_bwd_ksplit_kernel(RnnArgs a, char* xbuf, unsigned* census, int spin_limit) {
  static_assert(!TR || G == 3, "the training instance exists for the GRU");
  static_assert(!(TR && SP), "the split form takes the plain fp32 buffers");
  constexpr int NPL = SP ? 2 : 1;
  static_assert(NW == 8 && NT >= 2 && (NT % 2) == 0, "8 waves: wave w owns output columns [w*H/8, (w+1)*H/8) = NT 16-column tiles");
  constexpr int NL = NT / 2;                                  // 1 KB wave loads per gather = producers / 8 = tile pairs per wave
  constexpr int AST = 48;                                     // bf16 per staged row: 96 B pitch — the operand read (row = lane & 15, 16-byte chunk lane >> 4) is conflict-free under ds_read_b128's lane groups (80 B was not: 4 of its 8 cycles, scripts/probe_lds_pair.hip)
  // planes 0..G-1: dGh gate by gate (the MFMA operand); GRU plane 3: d(pre-activation of n) = the n column of dGx.  Double-buffered: ONE barrier per step
  __shared__ __attribute__((aligned(16))) __bf16 As[NPL][2][4][16][AST];
  const PRole role = persist_role(a, census, spin_limit, 2);
  if (!role.active) return;
  // (s_setprio 3 here — win every issue arbitration against a co-resident weight-gradient kernel, gemm_tn_group.h — changed nothing
  // in the co-residency experiment: what the recurrence loses there is queueing in the CU's memory path, profiles/r04_wgrad_side_ab.txt)
  const int dir = role.dir, bt = role.bt, slice = role.slice;          // slice = my 32 units = my producer index
  const bool l2_local = role.local != 0;
  const int T = a.T, B = a.B, H = a.H, lddy = a.lddy;
  const int gs = a.p_gs;                                                // workgroups per exchange group = H / 32 = 8 * NL
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsl = H >> 4, nchb = (G * H) >> 5;
  const long long groupbytes = (long long)gs * gs * 1024;               // [consumer][wave][producer][128 B]
  const long long slotbytes = (long long)2 * a.p_nbt * groupbytes;
  const long long planebytes = 2 * slotbytes;                           // SP: the lo plane of both slots lies behind the hi plane's
  char* gbase = xbuf + (long long)(dir * a.p_nbt + bt) * groupbytes;

  // ---- my K slice of W_hh (rows g*H + 32*slice .. +31, gate by gate) x this wave's NT column tiles -> registers (once).  The packed
  // backward operand of the other kernels holds exactly these fragments: [dir][16-column slice][32-row chunk][lane] (rnn_pack_kernel).
  f32x4 wreg[NPL][G][NT];
  const long long wplane = (long long)2 * nsl * nchb * 256;             // floats of one packed backward operand (SP: hi, then lo)
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wreg[pl][g][nt] = *reinterpret_cast<const f32x4*>(a.wp + pl * wplane + ((((long long)dir * nsl + (wave * NT + nt)) * nchb + (g * (H >> 5) + slice)) * 256) + lane * 4);

  // ---- this lane's (batch row, hidden unit) pair: where the reduce-scatter below leaves its complete sum.  Lane = (b5, b4, b3, h, q):
  // piece (h, q) of the line; of the 8 sums of a piece [dword d = 2 b5 + b4][half b3] it keeps unit (d >> 1) * 16 + 4q + (d & 1) * 2 + b3
  const int q4 = lane & 3, hrow = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
  const int unit = b5 * 16 + 4 * q4 + 2 * b4 + b3, row = 2 * wave + hrow;
  const int b = bt * 16 + row, j = slice * 32 + unit;
  const bool pact = b < B;
  const int plen = pact ? a.lens[b] : 0;
  const __bf16* gates_bf = a.gates_bf;
  float dcar = 0.f;                                                     // GRU dh*z / LSTM dc*f of the step before (own pair)

  // Element offsets of this lane's pair at the step being processed, stepped by a wave-uniform stride per time step (one 64-bit add each instead
  // of the multiply chains of a fresh index computation):
  //   eH into the (T,B,2,H) tensors, eG into the (T,B,2,G*H) tensors, eY into dy
  const int t0 = dir == 0 ? T - 1 : 0;
  const long long dH = (dir == 0 ? -1LL : 1LL) * B * 2 * H, dG = dH * G, dY = (dir == 0 ? -1LL : 1LL) * B * lddy;
  long long eH = (((long long)t0 * B + b) * 2 + dir) * H + j, eG = (((long long)t0 * B + b) * 2 + dir) * G * H + j, eY = ((long long)t0 * B + b) * lddy + j;
  // BatchNorm1d backward of the layer above, applied on the fly (ds2_rnn_bwd_bn): dy_used = bn1 dy + bnx x + bn0 with this unit's constants
  //   k1 = gamma rstd, k2 = k1 s0 / M, k3 = k1 rstd s1 / M  ->  bn1 = k1, bnx = -k3, bn0 = k3 mean - k2     (norm.hip: bn1d_bwd_apply_kernel)
  const bool bn = a.bn_x != nullptr;
  // bn_x as bf16 (a.bn_x_bf16): the lane loads the aligned DWORD that holds its element and its neighbour's — the same load instruction as the
  // fp32 form — and keeps its half (even elements << 16, odd ones & 0xffff0000); the widening happens where the value is consumed, in the
  // coefficient math at the END of a step (not on the dependent chain)
  const bool bxbf = bn && a.bn_x_bf16 != 0;
  const unsigned bx_sh = (bxbf && !(j & 1)) ? 16u : 0u, bx_mask = bxbf ? 0xffff0000u : 0xffffffffu;
  const long long dX = (dir == 0 ? -1LL : 1LL) * B * a.ldbnx;
  long long eX = ((long long)t0 * B + b) * a.ldbnx + j;
  float bn1 = 1.f, bnx = 0.f, bn0 = 0.f;
  if (bn) {
    const float rs = rsqrtf(a.bn_var[j] + a.bn_eps), k1 = a.bn_gamma[j] * rs, invm = 1.f / (float)((long long)T * B);
    const float k2 = k1 * a.bn_s0[j] * invm, k3 = k1 * rs * a.bn_s1[j] * invm;
    bn1 = k1; bnx = -k3; bn0 = __builtin_fmaf(k3, a.bn_mean[j], -k2);
  }
  struct Ops { bf16x4_ rec; float g0, g1, g2, g3, ax, dy, prev, bx; };   // see rnn_bwd_persistent_kernel: fetched one step ahead, kept raw
  // GRU training path (packed gate records + bf16 dGx, the shapes the bench runs): the four operand loads of a step are issued by INLINE ASM
  // and retired by a COUNTED wait (`fetch_wait`).  The vector-memory counter retires in order and counts stores too; left to the compiler,
  // the wait in front of the coefficient math at the end of a step was `vmcnt(0)` — i.e. it waited for the step's four publish stores and
  // its result store (issued just before, write-through: ~0.5 us) although the loads it needs were issued a whole step earlier.  With the
  // loads invisible to the compiler's counter bookkeeping, the wait names exactly what may still be outstanding: the NL publish stores.
  // Every lane issues every load (inactive lanes / a missing BatchNorm / the last step read a valid dummy address and their value is
  // discarded after the wait), so no result register is merged or copied between its load and its wait.
  constexpr bool ASM_FETCH = TR;
  constexpr bool asm_fetch = TR;                               // (the launcher picks TR exactly when gates_bf and dgx_bf are given)
  struct Raw { f32x2_ rec; float dy, bx, prev; };
  // (a separate template instance, not a run-time branch: with the compiler-managed `fetch` below on another path of the same code, its
  // pending loads put the compiler's own vmcnt(0) right back behind the counted wait)
  auto fetch_asm = [&](long long fH, long long fY, long long fX, bool has_prev) {
    Raw r;
    const bool on = pact && asm_fetch;
    const void* prec = on ? (const void*)(gates_bf + 4 * fH) : (const void*)a.dy;
    const float* pdy = on ? a.dy + fY : a.dy;
    const float* pbx = (on && bn) ? (bxbf ? a.bn_x + (fX >> 1) : a.bn_x + fX) : a.dy;
    const float* ppv = (on && has_prev) ? a.hbuf + fH + dH : a.hbuf;
    asm volatile("global_load_dwordx2 %0, %4, off nt\n\tglobal_load_dword %1, %5, off nt\n\tglobal_load_dword %2, %6, off nt\n\t"
                 "global_load_dword %3, %7, off"
                 : "=&v"(r.rec), "=&v"(r.dy), "=&v"(r.bx), "=&v"(r.prev)
                 : "v"(prec), "v"(pdy), "v"(pbx), "v"(ppv)
                 : "memory");
    return r;
  };
  // the loads of `r` have landed once at most YOUNGER vector-memory operations are outstanding (the counter retires in order and at least
  // that many were issued behind them); the record is used in place — same registers from the load to the last use
#define DS2_KS_FETCH_WAIT(r, YOUNGER) asm volatile("s_waitcnt vmcnt(%4)" : "+v"((r).rec), "+v"((r).dy), "+v"((r).bx), "+v"((r).prev) : "n"(YOUNGER) : "memory")
  auto raw_ops = [&](const Raw& r, bool has_prev) {
    return Ops{__builtin_bit_cast(bf16x4_, r.rec), 0.f, 0.f, 0.f, 0.f, 0.f, r.dy, has_prev ? r.prev : 0.f, __uint_as_float((__float_as_uint(r.bx) << bx_sh) & bx_mask)};
  };
  // operands of the step whose offsets are (fH, fG, fY); has_prev: that step has a predecessor in FORWARD order (= the step processed after it)
  auto fetch = [&](long long fH, long long fG, long long fY, long long fX, bool has_prev) {
    Ops o{bf16x4_{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!pact) return o;
    if (gates_bf) {
      o.rec = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_*>(gates_bf) + fH);
      if (G == 4) o.ax = ldnt(a.aux + fH);
    } else {
      const float* gp = a.gx + fG;
      o.g0 = ldnt(gp); o.g1 = ldnt(gp + H); o.g2 = ldnt(gp + 2 * H);
      if (G == 4) { o.g3 = ldnt(gp + 3 * H); o.ax = ldnt(a.aux + fH); }
      else o.g3 = ldnt(a.aux + fH);
    }
    o.dy = ldnt(a.dy + fY);
    if (bn) o.bx = __uint_as_float((__float_as_uint(ldnt(a.bn_x + (bxbf ? (fX >> 1) : fX))) << bx_sh) & bx_mask);   // ONE load either way (no branch in the step)
    if (has_prev) o.prev = (G == 3) ? a.hbuf[fH + dH] : a.aux[fH + dH];     // h / c of the previous frame in forward order = the NEXT step's row
    return o;
  };
  // The gate-derivative math is LINEAR in dh (GRU) / in dh and dc (LSTM): everything that does not depend on the carry — the products of the
  // saved gates, 1 - n^2, tanh(c) ... — is folded into five or six coefficients per pair as soon as a step's operands have landed, i.e. at the
  // END of the step before, in the shadow of the exchange; behind the gather only dh = dy + carry (+ dcar) and one multiply per output remain
  // on the step's dependent chain.  (Same formulas as gru_bwd_point / lstm_bwd_point, re-associated: agreement with the other kernel
  // families stays inside the K-split tolerance.)  Rows beyond B / frames beyond the sample's length: all coefficients zero.
  struct Coef { float k0, k1, k2, k3, kc, kcar, dy; };
  auto coefficients = [&](Ops o, int step) {
    Coef c{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tt = dir == 0 ? T - 1 - step : step;
    if (!(pact && tt < plen)) return c;
    if (TR || gates_bf) { o.g0 = (float)o.rec[0]; o.g1 = (float)o.rec[1]; o.g2 = (float)o.rec[2]; o.g3 = (float)o.rec[3]; }
    c.dy = bn ? __builtin_fmaf(bn1, o.dy, __builtin_fmaf(bnx, o.bx, bn0)) : o.dy;
    if constexpr (G == 3) {                                // r = g0, z = g1, n = g2, hn = g3
      const float cn = (1.f - o.g1) * (1.f - o.g2 * o.g2);
      c.kc = cn;                                           // d(pre-activation of n) = dh * cn
      c.k0 = cn * o.g3 * o.g0 * (1.f - o.g0);              // dGh_r
      c.k1 = (o.prev - o.g2) * o.g1 * (1.f - o.g1);        // dGh_z
      c.k2 = cn * o.g0;                                    // d(hn)
      c.kcar = o.g1;                                       // dh * z -> the next step's dcar
    } else {                                               // i = g0, f = g1, g = g2, o = g3, c = ax, c_prev = prev
      const float tc = tanhf_(o.ax);
      c.kc = o.g3 * (1.f - tc * tc);                       // dc = dcar + dh * kc
      c.k0 = o.g2 * o.g0 * (1.f - o.g0);
      c.k1 = o.prev * o.g1 * (1.f - o.g1);
      c.k2 = o.g0 * (1.f - o.g2 * o.g2);
      c.k3 = tc * o.g3 * (1.f - o.g3);                     // dGh_o = dh * k3
      c.kcar = o.g1;                                       // dc * f -> the next step's dcar
    }
    return c;
  };
  Ops nxt{};
  bool nraw_prev = false;
  Coef cf;
  if constexpr (ASM_FETCH) {
    Raw r0 = fetch_asm(eH, eY, eX, T > 1);
    DS2_KS_FETCH_WAIT(r0, 0);
    cf = coefficients(raw_ops(r0, T > 1), 0);
  } else {
    nxt = fetch(eH, eG, eY, eX, T > 1);
    cf = coefficients(nxt, 0);
  }

  // Results.  bf16 training path (dgx_bf given): the bf16 values are in the LDS planes anyway — after the step's barrier, wave p < 4 sends
  // plane p out with ONE 16-byte store per lane (lane = row (lane >> 2), 8 units (lane & 3)): GRU planes 0, 1, 3 -> dGx columns r, z, n and
  // plane 2 -> the bf16 copy of d(hn); LSTM plane g -> dGx column g.  The fp32 d(hn) (aux) is written lane by lane only when nobody asked
  // for the bf16 copy.  Plain fp32 buffers (tests) and the last step: lane by lane.
  const bool lds_out = TR || a.dgx_bf != nullptr;
  const bool aux_f32 = G == 3 && !(lds_out && a.dhn_bf);
  auto store_lane = [&](long long fH, long long fG, const float (&dgx)[G], float dax, bool all) {
    if (!pact) return;
    if (all) {
      if (a.dgx_bf) {
        __bf16* gb = a.dgx_bf + fG;
#pragma unroll
        for (int g = 0; g < G; ++g) __builtin_nontemporal_store((__bf16)dgx[g], gb + g * H);
      } else {
        float* gp = a.gx + fG;
#pragma unroll
        for (int g = 0; g < G; ++g) stnt(gp + g * H, dgx[g]);
      }
      if (G == 3 && a.dhn_bf) __builtin_nontemporal_store((__bf16)dax, a.dhn_bf + fH);
    }
    if (aux_f32) stnt(a.aux + fH, dax);
  };
  // plane store of wave p < 4: destination element offset of this lane's 8 units, stepped like eH / eG
  const int srow = lane >> 2, sb = bt * 16 + srow;
  const bool s_hn = G == 3 && wave == 2;                                    // this wave's plane is d(hn): (T,B,2,H) bf16 copy
  const bool s_on = lds_out && wave < 4 && sb < B && !(s_hn && !a.dhn_bf);
  __bf16* s_dst = s_hn ? a.dhn_bf : a.dgx_bf;
  long long sE = s_hn ? (((long long)t0 * B + sb) * 2 + dir) * H + slice * 32 + (lane & 3) * 8
                      : (((long long)t0 * B + sb) * 2 + dir) * G * H + (G == 3 ? (wave == 3 ? 2 : wave) : wave) * H + slice * 32 + (lane & 3) * 8;
  const long long sD = s_hn ? dH : dG;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  // gather: this wave's lines of all producers are contiguous.  publish: lane (q = lane >> 4, c = lane & 15) holds, per tile pair, units
  // [4q..4q+3 | 16+4q..] of batch row c: piece (c & 1, q) of the line of consumer wave c >> 1 — a wave-uniform base per store + one
  // per-lane 32-bit offset
  unsigned goff[NPL * NL];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int i = 0; i < NL; ++i) goff[pl * NL + i] = (unsigned)(pl * planebytes + (((slice * 8 + wave) * gs) * 128) + i * 1024 + lane * 16);
  const unsigned pend0 = (1u << (NPL * NL)) - 1u;
  const unsigned pub_lane_off = (unsigned)(((((lane & 15) >> 1) * gs + slice) * 128) + ((lane & 1) * 4 + (lane >> 4)) * 16);
  const long long pub_wave_off = (long long)(wave * NL) * 8 * gs * 128;
  const long long pub_step = (long long)8 * gs * 128;                    // next consumer workgroup (tile pair)

  PTRACE_DECL;
  vm_drained();
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? T - 1 - s : s;
    PTRACE(0);
    float carry = 0.f;
    if (s > 0) {
      const char* xin = gbase + (long long)((s - 1) & 1) * slotbytes;
      const unsigned tagw = ((unsigned)(s - 1) >> 1) & 1u;
      u32x4_ av[NPL * NL];
#pragma unroll
      for (int i = 0; i < NPL * NL; ++i) av[i] = u32x4_{0u, 0u, 0u, 0u};
      int spins = 0;
      unsigned pend = pend0;
      while (pend) {
#ifdef DS2_RNN_TRACE
        pt_acc[7] += 1;                                     // (trace build: poll passes, summed over the steps)
#endif
        poll_pass<NPL * NL>(av, goff, xin, pend);
#pragma unroll
        for (int i = 0; i < NPL * NL; ++i)
          if (pend & (1u << i)) {
            const bool ok = (((av[i].x ^ tagw) | (av[i].z ^ tagw)) & 1u) == 0u;      // both halves of this lane's piece carry the step's tag
            if (__ballot(ok) == ~0ull) pend &= ~(1u << i);
          }
        pend = __builtin_amdgcn_readfirstlane(pend);
        if (pend && ++spins > spin_limit) {
          if (lane == 0 && atomicCAS(&a.status[0], 0, 2) == 0) {
            a.status[1] = slice; a.status[2] = bt; a.status[3] = dir; a.status[4] = s; a.status[5] = wave;
            a.status[6] = (int)pend; a.status[7] = l2_local;
            __threadfence_system();
          }
          return;
        }
      }
      // ---- sum over this lane's NL producers (packed fp32 adds: S2[d] = the two bf16 of dword d), then the reduce-scatter over the 8
      // producers of a load.  (Dwords 0 / 2 carry the tag in the last mantissa bit of their low half: it is part of the value.)
      f32x2_ S2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < NPL * NL; ++i) {
        const u32x4_ c = av[i];
        S2[0] += f32x2_{__builtin_bit_cast(float, c.x << 16), __builtin_bit_cast(float, c.x & 0xffff0000u)};
        S2[1] += f32x2_{__builtin_bit_cast(float, c.y << 16), __builtin_bit_cast(float, c.y & 0xffff0000u)};
        S2[2] += f32x2_{__builtin_bit_cast(float, c.z << 16), __builtin_bit_cast(float, c.z & 0xffff0000u)};
        S2[3] += f32x2_{__builtin_bit_cast(float, c.w << 16), __builtin_bit_cast(float, c.w & 0xffff0000u)};
      }
      const float S[8] = {S2[0][0], S2[0][1], S2[1][0], S2[1][1], S2[2][0], S2[2][1], S2[3][0], S2[3][1]};   // S[2 d + half]
      float R[4], Q[2];
#ifndef DS2_KSPLIT_SHFL
#pragma unroll
      for (int k = 0; k < 4; ++k) {        // lanes 0-31 keep sums 0..3, lanes 32-63 sums 4..7
        const u32pair r = permlane32_swap(S[k], S[k + 4]);
        R[k] = r.a + r.b;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {        // even 16-lane rows keep 0..1, odd rows 2..3 of what the half kept
        const u32pair r = permlane16_swap(R[k], R[k + 2]);
        Q[k] = r.a + r.b;
      }
      {
        const float keep = b3 ? Q[1] : Q[0], send = b3 ? Q[0] : Q[1];
        const int got = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        carry = keep + __builtin_bit_cast(float, got);
      }
#else   // reference form of the same tree with ds_bpermute shuffles (scripts/probe_ksplit_reduce.hip checks the two against each other)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mine = b5 ? S[k + 4] : S[k], give = b5 ? S[k] : S[k + 4];
        const float got = __shfl_xor(give, 32, 64);
        R[k] = b5 ? got + mine : mine + got;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float mine = b4 ? R[k + 2] : R[k], give = b4 ? R[k] : R[k + 2];
        const float got = __shfl_xor(give, 16, 64);
        Q[k] = b4 ? got + mine : mine + got;
      }
      {
        const float keep = b3 ? Q[1] : Q[0], send = b3 ? Q[0] : Q[1];
        carry = keep + __shfl_xor(send, 8, 64);
      }
#endif
    }
    vm_drained();                                          // (the gather has waited for everything; tell the compiler)
    PTRACE(1);
    // the NEXT step's gate-math operands: issued here, right behind the gather, so that they have a whole step to land (the next poll's
    // wait retires them too: vmcnt is in order); their coefficients are formed at the end of this step
    const bool more = s + 1 < T;
    Raw nraw;
    if constexpr (ASM_FETCH) {
      nraw_prev = s + 2 < T;
      nraw = fetch_asm(more ? eH + dH : eH, more ? eY + dY : eY, more ? eX + dX : eX, nraw_prev);
    } else {
      if (more) nxt = fetch(eH + dH, eG + dG, eY + dY, eX + dX, s + 2 < T);
    }

    float dgh[G], dgx[G], dax = 0.f;
    if constexpr (G == 3) {
      const float dh = cf.dy + carry + dcar;
      dgh[0] = dh * cf.k0; dgh[1] = dh * cf.k1; dgh[2] = dh * cf.k2;
      dgx[0] = dgh[0]; dgx[1] = dgh[1]; dgx[2] = dh * cf.kc;
      dax = dgh[2];
      dcar = dh * cf.kcar;
    } else {
      const float dh = cf.dy + carry;
      const float dc = __builtin_fmaf(dh, cf.kc, dcar);
      dgh[0] = dc * cf.k0; dgh[1] = dc * cf.k1; dgh[2] = dc * cf.k2; dgh[G - 1] = dh * cf.k3;
#pragma unroll
      for (int g = 0; g < G; ++g) dgx[g] = dgh[g];
      dcar = dc * cf.kcar;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) bs[g] += dgx[g];
    if (G == 3) bs[3] += dax;
    PTRACE(2);
    if (!more) {                                           // nobody reads the partials of the last step
      store_lane(eH, eG, dgx, dax, true);
      break;
    }

    // ---- my K slice of dGh_s (+ the n column of dGx) -> LDS: every wave's MFMA operand, and the way the bf16 results leave
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const __bf16 hi = (__bf16)dgh[g];
      As[0][s & 1][g][row][unit] = hi;
      if constexpr (SP) As[NPL - 1][s & 1][g][row][unit] = (__bf16)(dgh[g] - (float)hi);
    }
    if (G == 3 && !SP) As[0][s & 1][3][row][unit] = (__bf16)dgx[2];
    __syncthreads();
    PTRACE(3);
    bf16x8 af[NPL][G];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int g = 0; g < G; ++g) af[pl][g] = *reinterpret_cast<const bf16x8*>(&As[pl][s & 1][g][lane & 15][(lane >> 4) * 8]);
    u32x4_ outv = u32x4_{0u, 0u, 0u, 0u};
    if (lds_out && wave < 4) outv = *reinterpret_cast<const u32x4_*>(&As[0][s & 1][wave][srow][(lane & 3) * 8]);
    // transposed product: acc[nt][r] of lane (q, c) = partial dh of batch row c, output unit 16 nt + 4q + r.
    // PUBLISH: piece = [units 4q..4q+3 | 16+4q..16+4q+3] of batch row c as bf16 (round to nearest even); bit 0 of each 8-byte half — the last
    // mantissa bit of its first value — is REPLACED by the tag and stays part of the value the consumer adds.
    // Schedule: the two waves of a SIMD share its matrix pipe, and left to itself the older wave issues all of its G * NT MFMAs first — the
    // younger one then converts and stores its whole result behind 2 x G * NT MFMAs (0.85 us against 0.37 in the timeline probe).  So the
    // product goes tile pair by tile pair, software-pipelined by one: MFMAs of pair p + 1, THEN convert + store pair p (which stalls this
    // wave on pair p's results and hands the pipe to the other wave); sched_barriers keep the compiler from regrouping it.
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xout = gbase + (long long)(s & 1) * slotbytes + pub_wave_off;
    const unsigned tagw = ((unsigned)s >> 1) & 1u;
    auto mfma_pair = [&](int pr) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if constexpr (SP) {                                  // smallest terms first: W_hi.d_lo, W_lo.d_hi
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            acc[2 * pr + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[0][g][2 * pr + q]), af[NPL - 1][g], acc[2 * pr + q], 0, 0, 0);
            acc[2 * pr + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[NPL - 1][g][2 * pr + q]), af[0][g], acc[2 * pr + q], 0, 0, 0);
          }
        }
        acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[0][g][2 * pr]), af[0][g], acc[2 * pr], 0, 0, 0);
        acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[0][g][2 * pr + 1]), af[0][g], acc[2 * pr + 1], 0, 0, 0);
      }
    };
    auto publish_pair = [&](int pr) {
      const f32x4 lo = acc[2 * pr], hi = acc[2 * pr + 1];
      const bf16x2_ p0 = __builtin_convertvector((f32x2_{lo[0], lo[1]}), bf16x2_), p1 = __builtin_convertvector((f32x2_{lo[2], lo[3]}), bf16x2_);
      const bf16x2_ p2 = __builtin_convertvector((f32x2_{hi[0], hi[1]}), bf16x2_), p3 = __builtin_convertvector((f32x2_{hi[2], hi[3]}), bf16x2_);
      const u32x4_ v = {(__builtin_bit_cast(unsigned, p0) & ~1u) | tagw, __builtin_bit_cast(unsigned, p1),
                        (__builtin_bit_cast(unsigned, p2) & ~1u) | tagw, __builtin_bit_cast(unsigned, p3)};
      if (l2_local) store16_base<true>(xout + (long long)pr * pub_step, pub_lane_off, v);
      else store16_base<false>(xout + (long long)pr * pub_step, pub_lane_off, v);
      if constexpr (SP) {
        // the lo piece: what the TAGGED hi piece leaves of the fp32 partial (so the tag bit of the hi piece is compensated exactly), itself
        // tagged in the last bit of two of its values (2^-17 of the partial)
        auto lo2 = [](float x0, float x1, unsigned hw) {
          const f32x2_ r = {x0 - __builtin_bit_cast(float, hw << 16), x1 - __builtin_bit_cast(float, hw & 0xffff0000u)};
          return __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_));
        };
        const u32x4_ w = {(lo2(lo[0], lo[1], v.x) & ~1u) | tagw, lo2(lo[2], lo[3], v.y), (lo2(hi[0], hi[1], v.z) & ~1u) | tagw, lo2(hi[2], hi[3], v.w)};
        if (l2_local) store16_base<true>(xout + planebytes + (long long)pr * pub_step, pub_lane_off, w);
        else store16_base<false>(xout + planebytes + (long long)pr * pub_step, pub_lane_off, w);
      }
    };
#ifndef DS2_KS_PIPE
#define DS2_KS_PIPE 1
#endif
#if DS2_KS_PIPE == 1
    mfma_pair(0);
#pragma unroll
    for (int pr = 0; pr < NL; ++pr) {
      if (pr + 1 < NL) mfma_pair(pr + 1);
      __builtin_amdgcn_sched_barrier(0);
      publish_pair(pr);
      __builtin_amdgcn_sched_barrier(0);
    }
#elif DS2_KS_PIPE == 2          // tuning variants (scripts/ab_ks_pipe.sh): pair by pair without look-ahead
#pragma unroll
    for (int pr = 0; pr < NL; ++pr) {
      mfma_pair(pr);
      __builtin_amdgcn_sched_barrier(0);
      publish_pair(pr);
      __builtin_amdgcn_sched_barrier(0);
    }
#elif DS2_KS_PIPE == 3          // two pairs per group, pipelined by one group
    mfma_pair(0);
    if (NL > 1) mfma_pair(1);
#pragma unroll
    for (int pr = 0; pr < NL; pr += 2) {
      if (pr + 2 < NL) mfma_pair(pr + 2);
      if (pr + 3 < NL) mfma_pair(pr + 3);
      __builtin_amdgcn_sched_barrier(0);
      publish_pair(pr);
      if (pr + 1 < NL) publish_pair(pr + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#elif DS2_KS_PIPE == 4          // look-ahead of two pairs
    mfma_pair(0);
    if (NL > 1) mfma_pair(1);
#pragma unroll
    for (int pr = 0; pr < NL; ++pr) {
      if (pr + 2 < NL) mfma_pair(pr + 2);
      __builtin_amdgcn_sched_barrier(0);
      publish_pair(pr);
      __builtin_amdgcn_sched_barrier(0);
    }
#else                           // everything first, then the conversions and stores (the compiler's own order)
#pragma unroll
    for (int pr = 0; pr < NL; ++pr) mfma_pair(pr);
#pragma unroll
    for (int pr = 0; pr < NL; ++pr) publish_pair(pr);
#endif
    PTRACE(4);
    PTRACE(5);
    // ---- this step's results go out LAST: their stores retire in the shadow of the exchange (nothing of the next step depends on them)
    if (s_on) __builtin_nontemporal_store(outv, reinterpret_cast<u32x4_*>(s_dst + sE));
    store_lane(eH, eG, dgx, dax, !lds_out);
    eH += dH; eG += dG; eY += dY; eX += dX; sE += sD;
    // the operands were requested right behind this step's gather; behind them only this step's NL publish stores (and, on four waves,
    // one result store, which then has to retire too) are outstanding
    if constexpr (ASM_FETCH) {
      DS2_KS_FETCH_WAIT(nraw, NL);
      cf = coefficients(raw_ops(nraw, nraw_prev), s + 1);                             // (this wait is in the exchange's shadow)
    } else {
      cf = coefficients(nxt, s + 1);
    }
    PTRACE(6);
  }
  if (a.bsum && pact) {
    float* o = a.bsum + (((long long)b * 2 + dir) * 4) * H + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) o[g * H] = bs[g];
  }
  PTRACE_DUMP(1);
#undef DS2_KS_FETCH_WAIT
}

// bytes of the two exchange slots: [2 slots][2 dirs x batch tiles][gs consumers][8 waves][gs producers][128 B]
size_t ksplit_xbuf_bytes(int B, int H) { const size_t gs = (size_t)H / 32; return (size_t)2 * 2 * ceil_div(B, 16) * gs * gs * 1024; }
bool ksplit_shape_ok(int H) { return H >= 256 && (H % 256) == 0 && H <= 1280; }

// 1 = launched, 0 = not eligible (the caller tries the all-gather persistent kernel, then the step kernels), 2 = eligible but cooling
// down after a starved launch (step kernels), < 0 = error
template <int G, bool SP = false>
int try_launch_ksplit_bwd(RnnArgs a, hipStream_t st) {
  constexpr int NPL = SP ? 2 : 1;
  static const char* env = getenv("DS2_RNN_PERSISTENT");
  static const char* envk = ds2_exp_getenv("DS2_RNN_KSPLIT");               // "0": keep the all-gather backward kernel (A/B runs)
  if ((env && env[0] == '0') || (envk && envk[0] == '0') || a.dbg) return 0;   // any selector: not this kernel
  if (!((a.gates_bf && a.dgx_bf) || (!a.gates_bf && !a.dgx_bf && a.gx))) return 0;
  if (SP && (a.gates_bf || a.dgx_bf)) return 0;                      // the split form: plain fp32 buffers
  if (!ksplit_shape_ok(a.H) || a.T < 2) return 0;
  const int gs = a.H / 32, nt = a.H / 128, nbt = ceil_div(a.B, 16);
  if (nt * G * 4 * NPL > (SP ? 150 : 176)) return 0;                 // W_hh fragments must leave room for the rest (256 registers per lane)
  if ((long long)gs * nbt * 2 > cu_count()) return 0;
  if (!persist_allowed(a.hctx, true)) return 2;                             // eligible, but a starved launch's cooldown is running
  a.p_nbt = nbt; a.p_gs = gs; a.p_cux = CUS_PER_XCD;
  a.p_census = xcd_local_fits(gs, 2 * nbt) ? 1 : 0;
  char* xbuf = reinterpret_cast<char*>(a.pk);
  const size_t xbytes = NPL * ksplit_xbuf_bytes(a.B, a.H);
  if (!a.prearmed) DS2_HIP(hipMemsetAsync(xbuf, 0xff, xbytes + CENSUS_BYTES, st));     // both slots: tag 1 = "not the data of steps 0 / 1"; census words = -1
  unsigned* census = reinterpret_cast<unsigned*>(xbuf + xbytes);
  dim3 grid(a.p_census ? cu_count() : gs * nbt * 2), block(NW * 64);
  static const char* sl = getenv("DS2_RNN_SPIN_LIMIT");
  const int spin_limit = sl ? atoi(sl) : (1 << 20);
  switch (nt) {
#define DS2_KS(NT_)                                                                                                       \
  case NT_:                                                                                                               \
    if constexpr (SP && NT_ * G * 4 * 2 <= 150) {                                                                         \
      hipLaunchKernelGGL((rnn_bwd_ksplit_kernel<G, NT_, false, true>), grid, block, 0, st, a, xbuf, census, spin_limit);  \
    } else if constexpr (!SP && NT_ * G * 4 <= 176) {                                                                     \
      if (G == 3 && a.gates_bf && a.dgx_bf) {                                                                             \
        if constexpr (G == 3) hipLaunchKernelGGL((rnn_bwd_ksplit_kernel<3, NT_, true>), grid, block, 0, st, a, xbuf, census, spin_limit); \
      } else {                                                                                                            \
        hipLaunchKernelGGL((rnn_bwd_ksplit_kernel<G, NT_, false>), grid, block, 0, st, a, xbuf, census, spin_limit);      \
      }                                                                                                                   \
    } else {                                                                                                              \
      return 0;                                                                                                           \
    }                                                                                                                     \
    break;
    DS2_KS(2) DS2_KS(4) DS2_KS(6) DS2_KS(8) DS2_KS(10)
#undef DS2_KS
    default: return 0;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn k-split backward launch failed: %s", hipGetErrorString(e));
  return 1;
}
