// Bidirectional GRU / LSTM recurrence (fp32), padded + masked formulation of
// pack_padded_sequence -> aten::gru/lstm -> pad_packed_sequence (blocks.py:87-89) and its backward.
//
// The input projections X W_ih^T + b_ih for all t and both directions are one big MFMA GEMM
// (gemm.hip); this file is the strictly sequential part, in two kernel families with bit-identical results:
//   * STEP kernels: one launch per time step and BOTH directions per launch (dir 0 walks t = s, dir 1 walks t = T-1-s); the kernel
//     boundary is the all-to-all seam, the W_hh slices a block re-reads every step stay in its XCD's L2.  fp32 mode, and the fallback.
//   * PERSISTENT kernels (bf16 mode, further down): one launch per layer, W_hh slice in registers, h_t exchanged between the resident
//     workgroups by tagged-payload polling, one workgroup barrier per time step.  5.8 -> 2.1 us (forward) / 6.9 -> 3.1 us (backward) per step.
//   * K-SPLIT persistent backward (rnn_bwd_ksplit.h, the bf16 default where H % 256 == 0): the product dGh W_hh split along K, bf16 partial
//     sums of dh exchanged instead of dGh (a third of the gather, no reset traffic): 2.4-2.5 us per step at c3.  Equal to the other two
//     families within a stated tolerance (one more bf16 rounding per partial sum), not to the bit.
//
// Step kernel = [h_{t-1} (BT x H) @ W_hh^T slice] on the f32 matrix cores (v_mfma_f32_16x16x4_f32)
// fused with the gate non-linearities, the per-sample length mask and the state write-back.
// A block owns 16 hidden units x all gates x BT (16|32) batch rows of one direction; K is split over
// the block's 8 waves and the partial tiles are reduced through LDS.
//
// FRAGMENT-MAJOR OPERANDS.  An MFMA 16x16x4 operand wants lane l to hold row (l & 15), k-quad (l >> 4):
// loading that straight from a row-major matrix makes the 64 lanes of one instruction touch 64 different
// 16-byte pieces of 16 rows — measured 12 B/clk/CU, the whole kernel's bound.  So both operands live in
// memory *in fragment order*: one (16 rows x 16 k) chunk = 64 lanes x 16 B = ONE contiguous 1 KiB block,
// and every wave-load is a perfectly coalesced `global_load_dwordx4` with no predication (zero padded):
//   * W_hh (fwd) / W_hh^T (bwd) are re-packed once per optimizer step (ds2_rnn_pack_whh_f32);
//   * h_t (fwd) and dGh_t (bwd) are written in packed form by the epilogue of the step that produces
//     them (ping-pong buffer), next to the plain-layout copies the big GEMMs consume.
//
// Saved for backward (in place of the x-projections): activated gates; aux = W_hn h + b_hn (GRU)
// or the cell state c (LSTM); h per direction.  Rows t >= len[b] hold zeros everywhere, which is
// what makes the reverse direction start at each sample's own last frame.
#include "common.h"
#include "permlane.h"
#include <type_traits>
#include <algorithm>


namespace {

#ifndef DS2_RNN_NW
#define DS2_RNN_NW 8
#endif
constexpr int NW = DS2_RNN_NW;          // waves per block

// Selector bits of ds2_debug_flags (include/ds2hip.h): 8 / 16 tile shapes of the wide step kernels, 64 step kernels instead of the persistent
// ones, 128 all-gather persistent backward instead of the K-split one.  The WORK-SKIPPING bits 1 (no h W_hh product) and 2 (no gate
// epilogue) of scripts/ablate_rnn.py exist only in a library built with -DDS2_ABLATE (make ABLATE=1); the shipped one ignores them.
#ifdef DS2_ABLATE
#define DS2_ABLATE_BIT(flags, bit) (((flags) & (bit)) != 0)
#else
#define DS2_ABLATE_BIT(flags, bit) false
#endif

// Streamed-once traffic (gate pre-activations in, gates / h / aux out, dy in) is marked non-temporal so that it does
// not evict the per-XCD working set that IS re-read every step (W_hh slices 3.1 MB + packed h at H=1024) from the 4 MB L2.
__device__ __forceinline__ float ldnt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float ldnt_bf(const __bf16* p) {
  return __uint_as_float((unsigned)__builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p)) << 16);
}

struct RnnArgs {
  float* gx;          // (T,B,2,G*H)  fwd: in x-proj / out gates ; bwd: in gates / out d(pre-activations wrt x-proj)
  float* aux;         // (T,B,2,H)    GRU fwd: out hn ; GRU bwd: in hn / out d(hn) ; LSTM: cell state (read-only in bwd)
  float* hbuf;        // (T,B,2,H)    h per direction (fwd: out, bwd: in)
  const float* wp;    // packed weights: fwd [2][nsl][G][nch][256] ; bwd [2][nsl][nchb][256]
  const float* bhh;   // (2, G*H) (fwd only)
  const float* dy;    // (T,B,H) grad wrt y = h_fwd + h_bwd (bwd only), row pitch lddy
  float* pk;          // packed moving operand, ping-pong: [2 parity][2 dir][nbt16][nchK][256]  (h fwd / dGh bwd)
  float* dcar;        // (2 parity, 2 dir, B, H) bwd carry: GRU dh*z ; LSTM dc*f
  __bf16* dgx_bf;     // bwd, optional: (T,B,2,G*H) bf16 — d(pre-activations) go HERE instead of overwriting gx (fp32)
  __bf16* gates_bf;   // optional (T,B,2,H,4) bf16: saved-for-backward record per hidden unit, ONE 8-byte store in forward and ONE load
                      // in backward instead of four each — GRU [r, z, n, hn], LSTM [i, f, g, o]; gx / (GRU) aux are then not written
  const int* lens;    // (B) valid output frames per sample
  int T, B, H, lddy;
  int nsl, nbt16;     // hidden slices of 16 ; allocated 16-row batch tiles (multiple of MB)
  int dbg;            // ablation flags (0 in production)
  // persistent kernels only: batch tiles per direction, workgroups per exchange group (= hidden slices), XCD census on/off, CUs per XCD
  int p_nbt, p_gs, p_census, p_cux;
  // persistent kernels only, optional (NULL: not produced): operands of the TN-form weight-gradient GEMMs, so that the backward pass needs
  // no cast / transpose pass over h, d(hn) or dGx
  __bf16* h_bf;       // fwd: (T,B,2,H) bf16 copy of hbuf
  __bf16* dhn_bf;     // bwd, GRU: (T,B,2,H) bf16 copy of d(hn) (aux)
  float* bsum;        // bwd: (B,2,4,H) per-batch-row sums over time of [d r, d z, d n, d(hn)] (GRU) / [d i, d f, d g, d o] (LSTM): bias gradients
  // bwd, K-split kernel only, optional (bn_x != NULL): `dy` is the gradient wrt the OUTPUT of the BatchNorm1d that follows this layer and the
  // kernel applies that BatchNorm's elementwise backward itself (ds2_rnn_bwd_bn): dy_used = k1 dy - k2 - k3 (x - mean)
  const float* bn_x;  // (T*B, H) pitch ldbnx: the BatchNorm's input = this layer's y
  const float *bn_mean, *bn_var, *bn_gamma, *bn_s0, *bn_s1;   // (H) batch statistics, weight, column sums of dy and of dy * xhat
  float bn_eps;
  int ldbnx;
  // persistent kernels: the caller's starvation record (ds2_rnn_ctx.status_dev, 8 ints) — {set, block x, y, z, step, wave, ok-mask lo, hi} of the
  // first wave that gave up polling; and (host side only, never dereferenced on the device) the context itself
  int* status;
  ds2_rnn_ctx* hctx;
  // forward, persistent kernels of the bf16 training mode only (ds2_rnn_fwd_x): the x-projections as the bf16 tensor the projection GEMM
  // wrote (ds2_gemm_bf16_nt_obf16), same (T,B,2,G*H) layout; `gx` is then NULL and never written (the gates go to gates_bf)
  const __bf16* gxb;
  // forward, persistent kernels, optional (NULL: not produced): (2, ceil(B/16), H) sums of h over time and over the 16 batch rows of a tile,
  // per direction — the column sums of this layer's output y = h_fwd + h_bwd without a pass over it (ds2_center_colstats)
  float* hsum;
  int bn_x_bf16;      // bwd, K-split kernel: bn_x is really a bf16 tensor (the centred operand of ds2_center_colstats; bn_mean is then its delta), ldbnx even
  int prearmed;       // host side only: the caller has filled the whole workspace with 0xff (ds2_rnn_ctx.ws_prearmed): persistent launchers skip their own fill
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));

// Timeline probe (scripts/probe_rnn_timeline.hip builds this file with -DDS2_RNN_TRACE; compiled out of the library): lane 0 of
// every wave of workgroup (0,0,0) stamps s_memtime at phase boundaries of the forward step kernel.
#ifdef DS2_RNN_TRACE
__device__ unsigned long long* g_rnn_trace = nullptr;   // [step][wave][8]
#define RNN_TRACE(step, k)                                                                                     \
  do {                                                                                                         \
    if (g_rnn_trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)         \
      g_rnn_trace[((long long)(step) * NW + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime();       \
  } while (0)
// persistent kernels: per-wave sums of the spans between consecutive stamps (registers only: no memory traffic inside the time loop),
// written out by workgroup 0 when the launch ends: g_rnn_trace[(kind * NW + wave) * 8 + k], kind 0 forward / 1 backward
#define PTRACE_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_amdgcn_s_memtime()
#define PTRACE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt_acc[k] += now_ - pt_last; pt_last = now_; } while (0)
#define PTRACE_DUMP(kind)                                                                                                     \
  do {                                                                                                                        \
    if (g_rnn_trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                                              \
      for (int k_ = 0; k_ < 8; ++k_) g_rnn_trace[((kind) * NW + (threadIdx.x >> 6)) * 8 + k_] = pt_acc[k_];                       \
  } while (0)
#else
#define RNN_TRACE(step, k) do { } while (0)
#define PTRACE_DECL do { } while (0)
#define PTRACE(k) do { } while (0)
#define PTRACE_DUMP(kind) do { } while (0)
#endif

// Pin every pointer argument into SGPRs at kernel entry: the compiler otherwise sinks the s_load of pointers that are first
// used inside a branch (lens, bhh) to that branch, which costs a second dependent scalar-load round trip (~0.3 us) before
// the first global load of every step kernel can issue.
__device__ __forceinline__ void hoist_kernargs(const RnnArgs& a) {
  // ("memory": also a compiler barrier, so that the operand loads issued before this point stay before the epilogue loads after it)
  asm volatile("" ::"s"(a.gx), "s"(a.aux), "s"(a.hbuf), "s"(a.wp), "s"(a.bhh), "s"(a.dy), "s"(a.pk), "s"(a.dcar), "s"(a.lens), "s"(a.dgx_bf), "s"(a.gates_bf)
               : "memory");
}

// element (row r, column k) of a packed [tiles][chunks][64 lanes][16 bytes] operand.
//   fp32 (16x16x4 MFMA):  chunk = k/16, lane = ((k%16)/4)*16 + r%16, component = k%4   (4 floats per lane)
//   bf16 (16x16x32 MFMA): chunk = k/32, lane = ((k%32)/8)*16 + r%16, component = k%8   (8 bf16 per lane)
// Either way one chunk = 1 KiB and a wave loads it with one coalesced global_load_dwordx4.
template <bool BF>
__device__ __forceinline__ long long packed_index(int r, int k, int nchunks) {
  if (BF) return (((long long)(r >> 4) * nchunks + (k >> 5)) * 64 + (((k & 31) >> 3) << 4) + (r & 15)) * 8 + (k & 7);
  return (((long long)(r >> 4) * nchunks + (k >> 4)) * 64 + (((k & 15) >> 2) << 4) + (r & 15)) * 4 + (k & 3);
}
template <bool BF>
__device__ __forceinline__ void packed_store(void* base, long long idx, float v) {
  if (BF) reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
  else reinterpret_cast<float*>(base)[idx] = v;
}
template <bool BF> __host__ __device__ constexpr int kchunk() { return BF ? 32 : 16; }

// backward: d(pre-activation) either overwrites the saved gate in the fp32 buffer or goes to the bf16 side buffer that the
// bf16-mode GEMMs consume directly (half the store bytes per step, no separate cast pass)
__device__ __forceinline__ void dgx_store(float* gx, __bf16* gb, int off, float v) {
  if (gb) __builtin_nontemporal_store((__bf16)v, gb + off);
  else stnt(gx + off, v);
}

// acc[i][j] += A-tile i (16 rows) x B-tile j (16 rows)^T over `nch` packed chunks.
// pa + i*sa / pb + j*sb point at this lane's float4 of chunk 0; consecutive chunks are 256 floats apart.
// Chunk c belongs to wave (c % NW); each wave keeps PF chunks of loads in flight.
// `after_prefetch` runs right after the first PF chunks of operand loads have been issued: the step kernels wait for their
// memory-resident kernel arguments and issue the epilogue-operand loads there, i.e. UNDER the operand fetch instead of before it.
template <bool BF, int MB, int NB, int PF, typename F>
__device__ __forceinline__ void mfma_packed(f32x4 (&acc)[MB][NB], int nch, int wave, const float* __restrict__ pa, long long sa,
                                            const float* __restrict__ pb, long long sb, F&& after_prefetch) {
  f32x4 fa[PF][MB], fb[PF][NB];
  auto load = [&](f32x4(&a)[MB], f32x4(&b)[NB], int c) {
    if (c < nch) {   // wave-uniform
#pragma unroll
      for (int i = 0; i < MB; ++i) a[i] = *reinterpret_cast<const f32x4*>(pa + i * sa + (long long)c * 256);
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = *reinterpret_cast<const f32x4*>(pb + j * sb + (long long)c * 256);
    }
  };
  int c = wave;
#pragma unroll
  for (int p = 0; p < PF; ++p) load(fa[p], fb[p], c + p * NW);
  after_prefetch();
  for (; c < nch; c += NW * PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      f32x4 ta[MB], tb[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) ta[i] = fa[p][i];
#pragma unroll
      for (int j = 0; j < NB; ++j) tb[j] = fb[p][j];
      load(fa[p], fb[p], c + (p + PF) * NW);
      if (c + p * NW < nch) {   // wave-uniform
        if constexpr (BF) {
#pragma unroll
          for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ta[i]), __builtin_bit_cast(bf16x8, tb[j]),
                                                                  acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
              for (int j = 0; j < NB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[i][e], tb[j][e], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
}

// The split form of the fp32 mode (three-term products on the bf16 matrix cores, see rnn_fwd_persistent_kernel SP): both operands come as a hi
// and a lo bf16 plane (pa_lo / pb_lo floats behind pa / pb), acc += a_lo.b_hi + a_hi.b_lo + a_hi.b_hi per chunk — same bytes per chunk as the
// fp32 operands (2 x 16 B per lane), 3 x 16 MFMA cycles instead of 8 x 32.
template <int MB, int NB, int PF, typename F>
__device__ __forceinline__ void mfma_packed_split(f32x4 (&acc)[MB][NB], int nch, int wave, const float* __restrict__ pa, long long sa, long long pa_lo,
                                                  const float* __restrict__ pb, long long sb, long long pb_lo, F&& after_prefetch) {
  f32x4 fa[PF][2][MB], fb[PF][2][NB];
  auto load = [&](f32x4(&a)[2][MB], f32x4(&b)[2][NB], int c) {
    if (c < nch) {   // wave-uniform
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        a[0][i] = *reinterpret_cast<const f32x4*>(pa + i * sa + (long long)c * 256);
        a[1][i] = *reinterpret_cast<const f32x4*>(pa + pa_lo + i * sa + (long long)c * 256);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        b[0][j] = *reinterpret_cast<const f32x4*>(pb + j * sb + (long long)c * 256);
        b[1][j] = *reinterpret_cast<const f32x4*>(pb + pb_lo + j * sb + (long long)c * 256);
      }
    }
  };
  int c = wave;
#pragma unroll
  for (int p = 0; p < PF; ++p) load(fa[p], fb[p], c + p * NW);
  after_prefetch();
  for (; c < nch; c += NW * PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      f32x4 ta[2][MB], tb[2][NB];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < MB; ++i) ta[pl][i] = fa[p][pl][i];
#pragma unroll
        for (int j = 0; j < NB; ++j) tb[pl][j] = fb[p][pl][j];
      }
      load(fa[p], fb[p], c + (p + PF) * NW);
      if (c + p * NW < nch) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ta[1][i]), __builtin_bit_cast(bf16x8, tb[0][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ta[0][i]), __builtin_bit_cast(bf16x8, tb[1][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ta[0][i]), __builtin_bit_cast(bf16x8, tb[0][j]), acc[i][j], 0, 0, 0);
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward step.  block = NW waves = MB 16-row batch tiles x NS 16-unit slices x all gates, K split over the waves.
// grid = (slice group, batch tile, direction): the linear workgroup id is x + gridDim.x * (bt + nbt * dir), so the batch tiles (and
// directions) of one slice index sit a multiple of gridDim.x blocks apart, i.e. on the same XCD when gridDim.x % 8 == 0: each XCD's L2
// holds every W_hh slice once, and keeps it across launches (profiles/r01_probe_l2_residency.txt).
// ------------------------------------------------------------------------------------------
// The leading parameters (13 dwords) are PRELOADED into SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count,
// Makefile): the operand fetch AND the HBM-latency epilogue loads (gate pre-activations, previous state, bias) need nothing from
// memory to compute their addresses, so they all issue at wave launch instead of behind a scalar load of the kernel-argument
// segment that measured 0.6-2 us (profiles/r01_probe_rnn_timeline.txt).  The remaining arguments (RnnArgs) arrive under them.
//   prev = h buffer (GRU) / cell-state buffer (LSTM);  s_H = s | H << 16;  T_B = T | B << 16;  nbt16_dbg = tiles | flags << 16
template <int G, int MB, int NS, bool BF>
__global__ __launch_bounds__(NW * 64) void rnn_fwd_step_kernel(const float* pk, const float* wp, float* gxbase, const float* prev,
                                                               const float* bhh, int s_H, int T_B, int nbt16_dbg, RnnArgs a) {
  __shared__ __attribute__((aligned(16))) f32x4 red[NW][MB * NS * G][64];
  constexpr int NTHR = NW * 64;
  constexpr int PAIRS = (MB * NS * 256 + NTHR - 1) / NTHR;   // (b, j) pairs per thread
  const int dir = blockIdx.z;
  const int slice = blockIdx.x, bt = blockIdx.y;
  const int s = s_H & 0xffff, H = (int)((unsigned)s_H >> 16);
  const int T = T_B & 0xffff, B = (int)((unsigned)T_B >> 16);
  RNN_TRACE(s, 0);
  const int nbt16 = nbt16_dbg & 0xffff, dbg = nbt16_dbg >> 16;
  const int nsl = (H + 15) >> 4;
  const int j0 = slice * (16 * NS), b0 = bt * (16 * MB);   // slice = blockIdx.x = NS consecutive 16-unit slices
  const int nch = (H + kchunk<BF>() - 1) / kchunk<BF>();
  const bool has_prev = s > 0;
  const int t = dir == 0 ? s : T - 1 - s;
  const int tp = dir == 0 ? t - 1 : t + 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  float* pk_out = const_cast<float*>(pk) + ((long long)((s & 1) * 2 + dir) * nbt16) * nch * 256;
  const float* pk_in = pk + ((long long)(((s + 1) & 1) * 2 + dir) * nbt16) * nch * 256;

  // ---- epilogue operands and store addresses: computed / loaded under the GEMM's operand fetch (see mfma_packed), used after it.
  // Nothing here depends on a loaded value (the length mask is applied after the GEMM).
  float pgx[PAIRS][G], pb[PAIRS][G], pprev[PAIRS];
  int plen[PAIRS];
  bool pact[PAIRS];
  float* gxp[PAIRS];             // this pair's gate row in gx
  long long rowH[PAIRS], hpi[PAIRS];
  auto issue_hbm_loads = [&]() {
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int q = threadIdx.x + i * NTHR;
      const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8, mb = sub / NS, ns = sub % NS;
      const int b = b0 + mb * 16 + brow, j = j0 + ns * 16 + jl;
      pact[i] = (sub < MB * NS) && b < B && j < H;
      plen[i] = 0;
      pprev[i] = 0.f;
      const long long row = ((long long)t * B + b) * 2 + dir;
      gxp[i] = gxbase + row * G * H + j;
      rowH[i] = row * H + j;
      hpi[i] = packed_index<BF>(b, j, nch);
#pragma unroll
      for (int g = 0; g < G; ++g) { pgx[i][g] = 0.f; pb[i][g] = 0.f; }
      if (pact[i]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          pgx[i][g] = ldnt(gxp[i] + g * H);
          pb[i][g] = bhh[(dir * G + g) * H + j];
        }
        if (has_prev) pprev[i] = prev[(((long long)tp * B + b) * 2 + dir) * H + j];
      }
    }
  };
  auto issue_epilogue_loads = [&]() {
    // the memory-resident arguments are needed from here on (lengths now, output pointers after the barrier)
    hoist_kernargs(a);
    RNN_TRACE(s, 1);
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int q = threadIdx.x + i * NTHR;
      const int b = b0 + ((q >> 8) / NS) * 16 + ((q >> 4) & 15);
      if (pact[i]) plen[i] = a.lens[b];
    }
  };

  f32x4 acc[MB][NS * G];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int g = 0; g < NS * G; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  {
    // one code path (no GEMM at s == 0 / under ablation = zero chunks): the compiler must not merge the epilogue loads of two
    // branches back in front of the operand loads
    // The HBM-latency epilogue operands (gate pre-activations, bias, previous state) go out FIRST: they need only preloaded arguments, cost
    // ~0.1 us of issue in front of the operand fetch, and their 1-2 us HBM round trip then ends long before the gate math needs them
    // (issued behind the operand loads they queued behind 160 KB in the in-order memory pipe and were ~0.3 us late: 5.9 -> 5.65 us/step).
    issue_hbm_loads();
    asm volatile("" ::: "memory");
    const int nch_eff = (has_prev && !DS2_ABLATE_BIT(dbg, 1)) ? nch : 0;
    const float* pa = pk_in + ((long long)(bt * MB) * nch) * 256 + lane * 4;                      // + mb*nch*256 + c*256
    const float* pw = wp + ((((long long)dir * nsl + slice * NS) * G) * nch) * 256 + lane * 4;      // + (n*G+g)*nch*256 + c*256
    mfma_packed<BF, MB, NS * G, (NS == 2 ? 3 : 4)>(acc, nch_eff, wave, pa, (long long)nch * 256, pw, (long long)nch * 256, issue_epilogue_loads);
  }
  RNN_TRACE(s, 2);
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int g = 0; g < NS * G; ++g) red[wave][i * NS * G + g][lane] = acc[i][g];
  __syncthreads();
  RNN_TRACE(s, 3);
  if (DS2_ABLATE_BIT(dbg, 2)) return;

#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    if (!pact[i]) continue;
    const int q = threadIdx.x + i * NTHR;
    const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8;
    const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
    float* gx = gxp[i];
    float* ho = a.hbuf + rowH[i];
    float* ax = a.aux + rowH[i];
    if (!(t < plen[i])) {
      if (a.gates_bf) {
        __builtin_nontemporal_store(bf16x4_{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}, reinterpret_cast<bf16x4_*>(a.gates_bf) + rowH[i]);
        if (G == 4) stnt(ax, 0.f);
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) stnt(&gx[g * H], 0.f);
        stnt(ax, 0.f);
      }
      stnt(ho, 0.f);
      packed_store<BF>(pk_out, hpi[i], 0.f);
      continue;
    }
    float gh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += red[w][sub * G + g][src_lane][reg];
      gh[g] = sum + pb[i][g];
    }
#ifdef DS2_RNN_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RNN_TRACE(s, 6);                                   // partial sums read back from LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RNN_TRACE(s, 7);                                   // epilogue operands (gate pre-activations, bias, h_prev) have landed
#endif
    float hnew;
    if constexpr (G == 3) {
      const float r = sigmoidf_(pgx[i][0] + gh[0]);
      const float z = sigmoidf_(pgx[i][1] + gh[1]);
      const float n = tanhf_(__builtin_fmaf(r, gh[2], pgx[i][2]));      // explicit fma: the persistent kernel must contract identically
      if (a.gates_bf) {
        __builtin_nontemporal_store(bf16x4_{(__bf16)r, (__bf16)z, (__bf16)n, (__bf16)gh[2]}, reinterpret_cast<bf16x4_*>(a.gates_bf) + rowH[i]);
      } else {
        stnt(&gx[0], r); stnt(&gx[H], z); stnt(&gx[2 * H], n);
        stnt(ax, gh[2]);
      }
      hnew = __builtin_fmaf(z, pprev[i], (1.f - z) * n);
    } else {
      const float ig = sigmoidf_(pgx[i][0] + gh[0]);
      const float fg = sigmoidf_(pgx[i][1] + gh[1]);
      const float gg = tanhf_(pgx[i][2] + gh[2]);
      const float og = sigmoidf_(pgx[i][G - 1] + gh[G - 1]);
      const float c = __builtin_fmaf(fg, pprev[i], ig * gg);
      if (a.gates_bf) {
        __builtin_nontemporal_store(bf16x4_{(__bf16)ig, (__bf16)fg, (__bf16)gg, (__bf16)og}, reinterpret_cast<bf16x4_*>(a.gates_bf) + rowH[i]);
      } else {
        stnt(&gx[0], ig); stnt(&gx[H], fg); stnt(&gx[2 * H], gg); stnt(&gx[(G - 1) * H], og);
      }
      *ax = c;                       // c_{t} is re-read by the next step (same block): keep it cacheable
      hnew = og * tanhf_(c);
    }
    *ho = hnew;
    packed_store<BF>(pk_out, hpi[i], hnew);
  }
  RNN_TRACE(s, 4);
#ifdef DS2_RNN_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RNN_TRACE(s, 5);
#endif
}

// ------------------------------------------------------------------------------------------
// PERSISTENT forward (bf16 operands): the whole recurrence of one layer in ONE launch.
//
// Same decomposition as the step kernel (workgroup = 16 hidden units x all gates x MB batch tiles of one direction, K split over the 8
// waves), but the workgroups stay resident for all T steps, so that
//   * the W_hh slice is read ONCE and lives in REGISTERS (NCW chunks x G gates x 16 B per lane = 96 KB per workgroup at H = 1024):
//     per step a CU moves the 64 KB moving operand only, not 160 KB (the step kernel's bound, profiles/r01_pmc_rnn_mempath.txt);
//   * there is no launch boundary (1.3-1.7 us per step); the all-to-all of h_t is done with TAGGED PAYLOAD POLLING: every 16-byte
//     chunk of the packed h_t buffer is pre-set to a sentinel no bf16 pair of finite values can produce (0xFFFFFFFF = two NaNs), the
//     producer fires its piece with system-coherent (sc1) 16-byte stores and every consumer wave re-reads its own chunks with sc1
//     loads until no dword is the sentinel — the payload is the flag.  Measured 2.64 us per step for the complete 64 KB exchange
//     against 5.6 us with separate flags and 5.65 us for the step kernel's whole step (scripts/probe_step_exchange.hip).
// Buffers: FOUR packed h buffers used round-robin (h_s lives in buffer s & 3).  At step s a workgroup whose eight waves have all gathered
// h_{s-1} (each wave polls only its own 8 of the 64 producers, so this needs the workgroup barrier) knows that every member of its group
// has FINISHED gathering h_{s-2}; it resets its own piece of that buffer ((s+2) & 3) to the sentinel, two steps before h_{s+2} is due
// there, and makes sure the previous reset has been acknowledged before h_s goes out (see the loop).  Spins are bounded: a wave that never sees its operand records who it is and leaves (the others
// then starve and leave too); the host checks that record at the train step's sync point and raises.  The launcher only takes this
// path when every workgroup can be co-resident (grid <= CU count; nothing else runs on the device during a forward pass).
// Results are bit-identical to the step kernels: same K split, same accumulation and reduction order (tests/test_gpu_kernels.py).
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
// (the trailing s_nop: a > 64-bit VMEM store needs wait states before its data VGPRs may be overwritten; the compiler inserts them for its
//  own stores but cannot see inside inline asm, and back-to-back publishes then shipped partly overwritten payloads)
__device__ __forceinline__ void store16_sc1(void* p, u32x4_ v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory"); }
constexpr unsigned PSENT = 0xffffffffu;

// `s_waitcnt vmcnt(0)` in the form the compiler's own wait-insertion pass understands: after it, the scoreboard knows that every vector
// load issued so far has landed.  Placed right behind the gather (whose asm statement has really waited, invisibly to the compiler) and in
// front of the time loop, it keeps the pass from putting its own conservative vmcnt(0) — it cannot count outstanding operations across
// the loop's back edge — behind the loads the step has JUST issued (observed: a full HBM round trip in front of every step's LDS writes).
__device__ __forceinline__ void vm_drained() { __builtin_amdgcn_s_waitcnt(0x0f70); }   // vmcnt(0), expcnt / lgkmcnt untouched

// L2-local flavour of the publish: a PLAIN 16-byte store stays in the storing XCD's L2 (an sc1 store is written through and DROPPED from it, so
// that even a same-XCD reader then pays the fabric round trip).  Only used when the census below has put every member of an exchange group on
// ONE XCD: that XCD's L2 is then the coherence point of the whole exchange (readers bypass their L1 with sc1 loads).
// scripts/probe_xcd_local.hip: 32 KB gather 1.65 -> 1.06-1.38 us per step, 96 KB gather 2.54 -> 2.14-2.23 us.
__device__ __forceinline__ void store16_l2(void* p, u32x4_ v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store16_x(void* p, u32x4_ v, bool l2_local) {
  if (l2_local) store16_l2(p, v);
  else store16_sc1(p, v);
}
// ---- gather: ONE asm statement per poll pass --------------------------------------------------------------------------------------------
// An asynchronous load issued from inline asm leaves its destination registers unprotected until one's OWN s_waitcnt: between two asm
// statements the compiler is free to copy them, and it does as soon as the surrounding code changes shape (seen in the probe: whole-array
// v_mov copies right behind the load, i.e. of data that had not landed).  So the loads of one poll pass, their wave-uniform "chunk still
// pending" predicates and the wait are ONE statement: its outputs are final when it ends.  Chunk K is read iff bit K of `pend` is set.
#define DS2_PL(K) "s_bitcmp1_b32 %[pend], " #K "\n\ts_cbranch_scc0 .Lds2poll%=_" #K "\n\tglobal_load_dwordx4 %[d" #K "], %[o" #K "], %[base] sc1\n.Lds2poll%=_" #K ":\n\t"
#define DS2_PD(K) [d##K] "+v"(v[K])
#define DS2_PO(K) [o##K] "v"(off[K])
#define DS2_PTAIL [base] "s"(base), [pend] "s"(pend) : "memory", "scc"
template <int N> struct PollBlock;
#define DS2_POLLBLOCK(N, LOADS, DSTS, OFFS)                                                                                          \
  template <> struct PollBlock<N> {                                                                                                  \
    static __device__ __forceinline__ void run(u32x4_* v, const unsigned* off, const char* base, unsigned pend) {                    \
      asm volatile(LOADS "s_waitcnt vmcnt(0)" : DSTS : OFFS, DS2_PTAIL);                                                             \
    }                                                                                                                                \
  };
#define DS2_C ,
DS2_POLLBLOCK(1, DS2_PL(0),
              DS2_PD(0),
              DS2_PO(0))
DS2_POLLBLOCK(2, DS2_PL(0) DS2_PL(1),
              DS2_PD(0) DS2_C DS2_PD(1),
              DS2_PO(0) DS2_C DS2_PO(1))
DS2_POLLBLOCK(3, DS2_PL(0) DS2_PL(1) DS2_PL(2),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2))
DS2_POLLBLOCK(4, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3))
DS2_POLLBLOCK(5, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4))
DS2_POLLBLOCK(6, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5))
DS2_POLLBLOCK(7, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6))
DS2_POLLBLOCK(8, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6) DS2_PL(7),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6) DS2_C DS2_PD(7),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6) DS2_C DS2_PO(7))
DS2_POLLBLOCK(9, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6) DS2_PL(7) DS2_PL(8),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6) DS2_C DS2_PD(7) DS2_C DS2_PD(8),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6) DS2_C DS2_PO(7) DS2_C DS2_PO(8))
DS2_POLLBLOCK(10, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6) DS2_PL(7) DS2_PL(8) DS2_PL(9),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6) DS2_C DS2_PD(7) DS2_C DS2_PD(8) DS2_C DS2_PD(9),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6) DS2_C DS2_PO(7) DS2_C DS2_PO(8) DS2_C DS2_PO(9))
DS2_POLLBLOCK(11, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6) DS2_PL(7) DS2_PL(8) DS2_PL(9) DS2_PL(10),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6) DS2_C DS2_PD(7) DS2_C DS2_PD(8) DS2_C DS2_PD(9) DS2_C DS2_PD(10),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6) DS2_C DS2_PO(7) DS2_C DS2_PO(8) DS2_C DS2_PO(9) DS2_C DS2_PO(10))
DS2_POLLBLOCK(12, DS2_PL(0) DS2_PL(1) DS2_PL(2) DS2_PL(3) DS2_PL(4) DS2_PL(5) DS2_PL(6) DS2_PL(7) DS2_PL(8) DS2_PL(9) DS2_PL(10) DS2_PL(11),
              DS2_PD(0) DS2_C DS2_PD(1) DS2_C DS2_PD(2) DS2_C DS2_PD(3) DS2_C DS2_PD(4) DS2_C DS2_PD(5) DS2_C DS2_PD(6) DS2_C DS2_PD(7) DS2_C DS2_PD(8) DS2_C DS2_PD(9) DS2_C DS2_PD(10) DS2_C DS2_PD(11),
              DS2_PO(0) DS2_C DS2_PO(1) DS2_C DS2_PO(2) DS2_C DS2_PO(3) DS2_C DS2_PO(4) DS2_C DS2_PO(5) DS2_C DS2_PO(6) DS2_C DS2_PO(7) DS2_C DS2_PO(8) DS2_C DS2_PO(9) DS2_C DS2_PO(10) DS2_C DS2_PO(11))
#undef DS2_C
#undef DS2_POLLBLOCK
// One poll pass over N chunks: blocks of <= 12 loads + their wait (N <= 12 is one statement = one round trip — every bf16 shape with 16-row
// tiles; the few wider shapes pay one more round trip per further block, and only for blocks that still have something pending).
template <int N>
__device__ __forceinline__ void poll_pass(u32x4_* v, const unsigned* off, const char* base, unsigned pend) {
  static_assert(N >= 1 && N <= 32, "pending mask is 32 bits");
  // both are wave-uniform by construction; say so in a way the compiler can see ("s" operands must be in SGPRs)
  pend = __builtin_amdgcn_readfirstlane(pend);
  {
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    base = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | (unsigned long long)lo);   // (the builtin returns a SIGNED int)
  }
  if constexpr (N <= 12) {
    PollBlock<N>::run(v, off, base, pend);
  } else {
    if (pend & 0xfffu) PollBlock<12>::run(v, off, base, pend & 0xfffu);
    if (pend >> 12) poll_pass<N - 12>(v + 12, off + 12, base, pend >> 12);
  }
}

// ---- who am I: (direction, batch tile, hidden slice) of a persistent workgroup ------------------------------------------------------------
// An exchange group = the p_gs workgroups (hidden slices) of one (direction, batch tile): they trade h_t / dGh_t with each other every time
// step and with nobody else.  Census mode (p_census; the launcher starts one workgroup per CU then): every workgroup reads the id of the
// XCD it REALLY runs on (HW_REG_XCC_ID), draws a slot on that XCD with one atomic, waits until the whole grid has arrived, and — if every
// XCD holds enough workgroups for the groups assigned to it (group g lives on XCD g % 8) — takes the role (group, slice) its slot says, so
// that a group provably shares one L2 and exchanges through it (`local`).  Workgroups left over exit.  Nothing is assumed about the
// dispatcher's placement: an uneven census just selects the placement-independent protocol (sc1 stores) with roles by workgroup id.
// All census words start at 0xffffffff (the launcher's single memset covers the exchange buffers' sentinel AND these counters).
struct PRole { int dir, bt, slice, local, active; };
__device__ __forceinline__ PRole persist_role(const RnnArgs& a, unsigned* census, int spin_limit, int kind) {
  __shared__ int s_role[4];
  const int nbt = a.p_nbt, gs = a.p_gs, ngroups = 2 * nbt, wg = blockIdx.x;
  int group = -1, slice = 0, local = 0;
  auto by_id = [&](int& g, int& sl) {
    if (wg >= ngroups * gs) { g = -1; sl = 0; }
    else if (ngroups == 8) { g = wg & 7; sl = wg >> 3; }     // 8 groups: group-major, one group per XCD under round-robin placement
    else { sl = wg % gs; g = wg / gs; }                      // slice-major (measured better for the other shapes)
  };
  if (!a.p_census) {
    by_id(group, slice);
  } else {
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xfu;
      const unsigned slot = atomicAdd(&census[xcc & 7u], 1u) + 1u;
      if (xcc > 7u) atomicAdd(&census[9], 1u);
      atomicAdd(&census[8], 1u);
      int spins = 0;
      bool all = false;
      while (!(all = __hip_atomic_load(&census[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u >= gridDim.x) && ++spins <= spin_limit)
        __builtin_amdgcn_s_sleep(1);
      int g = -2, sl = 0, loc = 0;
      if (all) {
        const int gpx = a.p_cux / gs;                                            // groups one XCD can host
        bool ok = __hip_atomic_load(&census[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu;
        for (int x = 0; x < 8; ++x) {
          const int need = (ngroups + 7 - x) / 8;
          const unsigned cnt = __hip_atomic_load(&census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
          ok = ok && need <= gpx && cnt >= (unsigned)(need * gs);
        }
        if (ok) {
          loc = 1;
          const int lg = (int)slot / gs, need_me = (ngroups + 7 - (int)xcc) / 8;
          if (lg < need_me) { g = lg * 8 + (int)xcc; sl = (int)slot % gs; }
          else g = -1;
        } else {
          by_id(g, sl);
        }
      } else if (atomicCAS(&a.status[0], 0, 3) == 0) {     // the grid never became resident: record it (the host raises), leave
        a.status[1] = wg; a.status[2] = (int)xcc; a.status[3] = kind; a.status[4] = -1; a.status[5] = 0;
        a.status[6] = (int)__hip_atomic_load(&census[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1; a.status[7] = 0;
        __threadfence_system();
      }
      s_role[0] = g; s_role[1] = sl; s_role[2] = loc;
    }
    __syncthreads();
    group = s_role[0]; slice = s_role[1]; local = s_role[2];
  }
  PRole r;
  r.active = group >= 0;
  group = __builtin_amdgcn_readfirstlane(group < 0 ? 0 : group);
  r.dir = group / nbt; r.bt = group % nbt;
  r.slice = __builtin_amdgcn_readfirstlane(slice);
  r.local = __builtin_amdgcn_readfirstlane(local);
  return r;
}

// SP (fp32 mode, DS2_F32_RNN=split): the recurrent product h W_hh^T at fp32 accuracy on the bf16 matrix cores.  Both operands travel as TWO
// bf16 planes, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (2^-18 relative): W_hh as hi / lo fragment sets in registers, h_t as a hi and a
// lo plane of every exchange buffer (4 bytes per element, as the fp32 exchange), and a chunk's product is hi.hi + lo.hi + hi.lo — three
// 16x16x32 bf16 MFMAs (48 cycles) where the fp32 path issues eight 16x16x4 fp32 MFMAs (256 cycles) for the same 32 k.  Everything else
// (state, gate math, outputs, protocol) is the BF = true data path; results are within ~1e-6 of the fp32 kernels', not bit-identical to them.
typedef float f32x2r __attribute__((ext_vector_type(2)));
template <int G, int MB, int NS, int NCW, bool BF, bool SP = false>
__global__ __launch_bounds__(NW * 64) void rnn_fwd_persistent_kernel(RnnArgs a, char* xbuf, unsigned* census, int spin_limit) {
  static_assert(MB * NS * 256 <= NW * 64, "one (row, unit) pair per thread");
  static_assert(!SP || BF, "the split form runs on the bf16 data path");
  constexpr int NPL = SP ? 2 : 1;                                             // operand planes: [hi | lo]
  __shared__ __attribute__((aligned(16))) f32x4 red[2][NW][MB * NS * G][64];       // double-buffered: ONE workgroup barrier per time step
  constexpr int KC = kchunk<BF>(), EPL = KC / 4;                              // units per chunk (32 | 16) and per 16-byte lane vector (8 | 4)
  using elem_t = typename std::conditional<BF, __bf16, float>::type;
  __shared__ __attribute__((aligned(16))) elem_t stage[NPL][NW][64];          // wave-private: a wave's 64 (row, unit) pairs = 64 / EPL complete 16-byte chunks
  // 1-D grid; (direction, batch tile, slice) from the XCD census or from the workgroup id: persist_role
  const PRole role = persist_role(a, census, spin_limit, 1);
  if (!role.active) return;
  const int dir = role.dir, bt = role.bt, slice = role.slice;
  const bool l2_local = role.local != 0;
  const int T = a.T, B = a.B, H = a.H;
  const int nsl = (H + 15) >> 4, nch = (H + KC - 1) / KC;
  const int j0 = slice * (16 * NS), b0 = bt * (16 * MB);      // slice = NS consecutive 16-unit slices
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long planebytes = (long long)2 * a.nbt16 * nch * 1024;          // one packed h plane: [dir][tile][chunk][64 lanes][16 B]
  const long long bufbytes = NPL * planebytes;                               // one exchange buffer (SP: hi plane, lo plane)
  const long long dirbase = (long long)dir * a.nbt16 * nch * 1024;

  // ---- W_hh slice -> registers (once; SP: the hi and the lo fragment set, one packed operand behind the other)
  f32x4 wreg[NPL][NCW][NS * G];
  bool cval[NCW];
  const long long wplane = (long long)2 * nsl * G * nch * 256;               // floats of one packed forward operand
#pragma unroll
  for (int k = 0; k < NCW; ++k) {
    const int c = wave + NW * k;
    cval[k] = c < nch;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int g = 0; g < NS * G; ++g)                  // (slice * NS + n, gate g') are consecutive 1 KiB-chunk rows of the packed W_hh
        wreg[pl][k][g] = cval[k] ? *reinterpret_cast<const f32x4*>(a.wp + pl * wplane + (((((long long)dir * nsl + slice * NS) * G + g) * nch + c) * 256) + lane * 4)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // lanes of a chunk whose 8 hidden units lie beyond H are never written by anybody: ignored by the poll, zero in the product
  bool lval[NCW];
#pragma unroll
  for (int k = 0; k < NCW; ++k) lval[k] = cval[k] && ((wave + NW * k) * KC + (lane >> 4) * EPL) < H;

  // ---- this thread's (batch row, hidden unit) pair: fixed for the whole layer, so the previous state stays in a register
  const int q = threadIdx.x;
  const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8, mb = sub / NS, ns = sub % NS;
  const int b = b0 + mb * 16 + brow, j = j0 + ns * 16 + jl;
  const bool pact = (sub < MB * NS) && b < B && j < H;
  const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
  // The gate math reads ONE dword of an accumulator vector: lane 16 r + jl of a wave wants dword 4 jl + r of its 64-vector row.  The LDS serves
  // a dword read 32 lanes at a time from 32 banks, and 4 jl + r hits every bank twice (jl and jl + 8) — 2 of the 4 cycles of each of the 24 reads
  // per thread and step were bank conflicts, 384 clocks per CU and step (SQ_LDS_BANK_CONFLICT, scripts/probe_lds_pair.hip).  Writer lanes with
  // bit 3 set therefore store their vector rotated by two dwords: unit jl + 8 then sits in banks 4 jl + ((r + 2) & 3), which nobody else in its
  // half-wave reads.
  const int red_sw = (lane >> 3) & 1;
  const int red_reg = (reg + 2 * ((jl >> 3) & 1)) & 3;
  const int plen = pact ? a.lens[b] : 0;
  float pb[G];
#pragma unroll
  for (int g = 0; g < G; ++g) pb[g] = pact ? a.bhh[(dir * G + g) * H + j] : 0.f;
  float pprev = 0.f;                                                         // h_{t-1} (GRU) / c_{t-1} (LSTM)
  float psum = 0.f;                                                          // sum over time of this pair's h (a.hsum)
  // A wave's 64 threads are 4 batch rows x 16 units = 64 / EPL complete 16-byte chunks of the packed buffer (chunk = one row, EPL units), so
  // every wave publishes (and resets) its own chunks — no workgroup-wide staging.  Lane p < 64 / EPL handles row (wave & 3) * 4 + (p & 3) and
  // unit group p >> 2 of this slice.
  constexpr int NPUB = 64 / EPL;
  const bool pub_lane = lane < NPUB && (wave >> 2) < MB * NS;
  long long pub_off;
  {
    const int wsub = wave >> 2, ju = j0 + (wsub % NS) * 16 + (lane >> 2) * EPL;   // first unit of this lane's group
    pub_off = dirbase + ((((long long)(bt * MB + wsub / NS) * nch + ju / KC) * 64) + ((ju % KC) / EPL) * 16 + (wave & 3) * 4 + (lane & 3)) * 16;
  }
  // element offsets of this thread's pair at the step being processed, stepped by a wave-uniform stride per time step (one 64-bit add each
  // instead of a fresh index computation with its multiply chain: the loop's VALU instructions are on the step's critical path)
  const int t0 = dir == 0 ? 0 : T - 1;
  const long long dH = (dir == 0 ? 1LL : -1LL) * B * 2 * H, dG = dH * G;
  long long eH = (((long long)t0 * B + b) * 2 + dir) * H + j, eG = (((long long)t0 * B + b) * 2 + dir) * G * H + j;
  // x-projections: fp32 (a.gx) or the bf16 tensor of ds2_gemm_bf16_nt_obf16 (a.gxb).  The NEXT step's values are requested one step ahead and
  // kept RAW (the loaded bits, untouched) until they are used: widening a bf16 right behind its load would put the load's full latency — a wait
  // for the vector-memory counter — into every time step (measured: +0.48 us per step)
  float pgx[G];
  unsigned pgx_raw[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { pgx[g] = 0.f; pgx_raw[g] = 0u; }
  // (compiled out of the split form of the fp32 mode: its instances sit at the register limit, and neither operand exists in that mode)
  constexpr bool TRAIN_OPS = BF && !SP;
  const bool gx_is_bf = TRAIN_OPS && a.gxb != nullptr;                       // (wave-uniform; BF training launches only)
  // bf16: the lane loads the aligned DWORD that holds its unit and its neighbour's (the same load instruction as the fp32 form) and keeps its
  // half: even units the low one (<< 16), odd units the high one (& 0xffff0000)
  const unsigned gx_shift = (gx_is_bf && !(j & 1)) ? 16u : 0u, gx_mask = gx_is_bf ? 0xffff0000u : 0xffffffffu;
  const unsigned* gxw = reinterpret_cast<const unsigned*>(a.gxb);           // dword view: element index e -> word e >> 1
  if (pact) {
#pragma unroll
    for (int g = 0; g < G; ++g) pgx[g] = gx_is_bf ? ldnt_bf(a.gxb + eG + g * H) : ldnt(a.gx + eG + g * H);
  }

  // saved-for-backward outputs of one time step
  auto store_outputs = [&](long long fH, long long fG, const float (&og)[4], float oaux, float oh, bool live) {
    if (!pact) return;
    if (a.gates_bf) {
      __builtin_nontemporal_store(bf16x4_{(__bf16)og[0], (__bf16)og[1], (__bf16)og[2], (__bf16)og[3]}, reinterpret_cast<bf16x4_*>(a.gates_bf) + fH);
      if (G == 4) a.aux[fH] = oaux;
    } else {
      float* gx = a.gx + fG;
      stnt(&gx[0], og[0]); stnt(&gx[H], og[1]); stnt(&gx[2 * H], og[2]);
      if (G == 4) { stnt(&gx[3 * H], og[3]); a.aux[fH] = oaux; }
      else stnt(a.aux + fH, live ? oaux : 0.f);
    }
    a.hbuf[fH] = oh;
    if (a.h_bf) __builtin_nontemporal_store((__bf16)oh, a.h_bf + fH);
  };

  // this wave's chunks of the packed exchange buffer (byte offsets from the buffer's direction base) and the all-pending mask
  static_assert(NPL * NCW * MB <= 32, "pending mask is 32 bits");
  unsigned goff[NPL * NCW * MB], pend0 = 0;
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int k = 0; k < NCW; ++k)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        goff[(pl * NCW + k) * MB + i] = (unsigned)(pl * planebytes + ((((long long)(bt * MB + i) * nch + (wave + NW * k)) * 64) + lane) * 16);
        if (cval[k]) pend0 |= 1u << ((pl * NCW + k) * MB + i);
      }

  PTRACE_DECL;
  vm_drained();                                         // prologue loads (W_hh slice, biases, lengths, first x-projections) have landed
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? s : T - 1 - s;
    PTRACE(0);
    if constexpr (TRAIN_OPS) {
      if (s > 0) {
        // x-projections of THIS step: requested one step ago, landed; widened HERE, in front of the poll (idle time), not behind it on the
        // dependent chain gather -> MFMA -> gate math
#pragma unroll
        for (int g = 0; g < G; ++g) pgx[g] = __uint_as_float((pgx_raw[g] << gx_shift) & gx_mask);
      }
    }
    f32x4 acc[MB][NS * G];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int g = 0; g < NS * G; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- gather h_{s-1}: poll this wave's chunks until none carries the sentinel (one asm statement per pass: poll_pass)
      const char* xin = xbuf + (long long)((s - 1) & 3) * bufbytes + dirbase;
      u32x4_ av[NPL * NCW * MB];
#pragma unroll
      for (int k = 0; k < NPL * NCW * MB; ++k) av[k] = u32x4_{0u, 0u, 0u, 0u};
      int spins = 0;
      unsigned pend = pend0;                              // wave-uniform: chunks of this wave that have not been seen complete yet
      while (pend) {
        // only the chunks that are still missing are read again: a blanket re-read of the whole operand by every waiting wave competes
        // with the very stores it is waiting for
#ifdef DS2_RNN_TRACE
        pt_acc[7] += 1;                                   // (trace build: poll passes, summed over the steps)
#endif
        poll_pass<NPL * NCW * MB>(av, goff, xin, pend);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int k = 0; k < NCW; ++k)
#pragma unroll
            for (int i = 0; i < MB; ++i)
              if (pend & (1u << ((pl * NCW + k) * MB + i))) {
                const u32x4_ c = av[(pl * NCW + k) * MB + i];
                const bool ok = !lval[k] || (c.x != PSENT && c.y != PSENT && c.z != PSENT && c.w != PSENT);
                if (__ballot(ok) == ~0ull) pend &= ~(1u << ((pl * NCW + k) * MB + i));
              }
        pend = __builtin_amdgcn_readfirstlane(pend);
        if (pend && ++spins > spin_limit) {
          // record who starved and on what (first failure only); the host raises at the step's sync point (no __builtin_trap: hipcc sinks
          // a trap to the kernel's common exit block, where it then fires on NORMAL completion too)
          if (lane == 0 && atomicCAS(&a.status[0], 0, 1) == 0) {
            a.status[1] = slice; a.status[2] = bt; a.status[3] = dir; a.status[4] = s; a.status[5] = wave;
            a.status[6] = (int)pend; a.status[7] = l2_local;
            __threadfence_system();
          }
          return;
        }
      }
#pragma unroll
      for (int k = 0; k < NCW; ++k)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const u32x4_ v = lval[k] ? av[k * MB + i] : u32x4_{0u, 0u, 0u, 0u};
          if constexpr (SP) {
            const u32x4_ vl = lval[k] ? av[(NCW + k) * MB + i] : u32x4_{0u, 0u, 0u, 0u};
#pragma unroll
            for (int g = 0; g < NS * G; ++g) {              // smallest terms first: lo.hi, hi.lo, then hi.hi
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vl), __builtin_bit_cast(bf16x8, wreg[0][k][g]), acc[i][g], 0, 0, 0);
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[NPL - 1][k][g]), acc[i][g], 0, 0, 0);
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[0][k][g]), acc[i][g], 0, 0, 0);
            }
          } else if constexpr (BF) {
#pragma unroll
            for (int g = 0; g < NS * G; ++g)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[0][k][g]), acc[i][g], 0, 0, 0);
          } else {
            const f32x4 vf = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)                   // same element order as the step kernel (mfma_packed): bit-identical sums
#pragma unroll
              for (int g = 0; g < NS * G; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], wreg[0][k][g][e], acc[i][g], 0, 0, 0);
          }
        }
    }
    vm_drained();                                       // (the gather has waited for everything; tell the compiler)
    PTRACE(1);                                          // gather done
#if defined(DS2_RNN_TRACE) && defined(DS2_DIAG_SECONDPASS)
    if (s > 0) {                                        // diagnostic: one more full pass over data that is certainly there -> slot 6
      u32x4_ dv[NCW * MB];
#pragma unroll
      for (int k = 0; k < NCW * MB; ++k) dv[k] = u32x4_{0u, 0u, 0u, 0u};
#if DS2_DIAG_SECONDPASS == 2                           // ... with PLAIN loads (through the L1, which the sc1 polls have not filled)
      if constexpr (NCW * MB == 4) {
        const char* dbase = xbuf + (long long)((s - 1) & 3) * bufbytes + dirbase;
        asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\tglobal_load_dwordx4 %2, %6, %8\n\t"
                     "global_load_dwordx4 %3, %7, %8\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(dv[0]), "=&v"(dv[1]), "=&v"(dv[2]), "=&v"(dv[3])
                     : "v"(goff[0]), "v"(goff[1]), "v"(goff[2]), "v"(goff[3]), "s"(dbase)
                     : "memory");
      }
#else
      poll_pass<NCW * MB>(dv, goff, xbuf + (long long)((s - 1) & 3) * bufbytes + dirbase, pend0);
#endif
#pragma unroll
      for (int k = 0; k < NCW * MB; ++k) asm volatile("" ::"v"(dv[k]));
      PTRACE(6);
    }
#endif

    if constexpr (!TRAIN_OPS) {
      if (s > 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) pgx[g] = __uint_as_float(pgx_raw[g]);  // x-projections of THIS step: loaded one step ago, landed
      }
    }
    // ---- the x-projections of the NEXT step are requested HERE: the vector-memory counter retires in order, so whatever is outstanding when the
    // next poll pass is issued delays it by its full latency; right behind the gather these loads have the whole step to land.  (The step's own
    // results go out at its end, behind the publish: nothing of the next step depends on them.)
    const bool more = s + 1 < T;
    if (more && pact) {
      if (gx_is_bf) {
#pragma unroll
        for (int g = 0; g < G; ++g) pgx_raw[g] = __builtin_nontemporal_load(gxw + ((eG + dG + g * H) >> 1));
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) pgx_raw[g] = __float_as_uint(ldnt(a.gx + eG + dG + g * H));
      }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int g = 0; g < NS * G; ++g) {
        // (two 8-byte halves, swapped in the lanes with bit 3 set: see red_reg below)
        float* rp = reinterpret_cast<float*>(&red[s & 1][wave][i * NS * G + g][lane]);
        *reinterpret_cast<f32x2r*>(rp + 2 * red_sw) = f32x2r{acc[i][g][0], acc[i][g][1]};
        *reinterpret_cast<f32x2r*>(rp + 2 - 2 * red_sw) = f32x2r{acc[i][g][2], acc[i][g][3]};
      }
    PTRACE(2);                                          // HBM section + MFMAs issued, partial sums written
    __syncthreads();
    PTRACE(3);                                          // barrier passed

    // ---- gate math (identical to the step kernel's epilogue)
    float out_g[4] = {0.f, 0.f, 0.f, 0.f}, out_aux = 0.f, hnew = 0.f;
    const bool live = pact && t < plen;
    if (live) {
      float gh[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[s & 1][w][sub * G + g][src_lane][red_reg];
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
      pprev = 0.f;                                      // beyond the sample's length: state is zero (what the step kernels re-read)
    }
    PTRACE(4);                                          // LDS sums + gate math
    // ---- publish h_s: wave-local assembly of this wave's 8 chunks, then ONE 16-byte store per publishing lane, then the reset of the same
    // chunks two buffers ahead.  "A consumer that has seen my chunk of h_s finds the sentinel, not my stale h_{s-3}, where it will poll for
    // h_{s+1}" needs my reset of buffer (s+1) & 3 — issued at step s-1 — to be visible before this publish: this step's gather has waited
    // for vmcnt(0) in between (at s = 0 the buffers still hold the launcher's fill).  The reset of buffer (s+2) & 3 is safe here: this wave
    // is past the step's barrier, so every workgroup of the group has published h_{s-1}, i.e. has finished gathering h_{s-2}.
    stage[0][wave][lane] = (elem_t)hnew;                // rows beyond B / units beyond H publish zeros: consumers wait for every chunk
    if constexpr (SP) stage[NPL - 1][wave][lane] = (elem_t)(hnew - (float)(elem_t)hnew);
    if (pub_lane) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        store16_x(xbuf + (long long)(s & 3) * bufbytes + pl * planebytes + pub_off,
                  *reinterpret_cast<const u32x4_*>(&stage[pl][wave][(lane & 3) * 16 + (lane >> 2) * EPL]), l2_local);
        store16_x(xbuf + (long long)((s + 2) & 3) * bufbytes + pl * planebytes + pub_off, u32x4_{PSENT, PSENT, PSENT, PSENT}, l2_local);
      }
    }
    PTRACE(5);                                          // publish issued
    // ---- the step's saved-for-backward outputs: last, in the shadow of the exchange
    store_outputs(eH, eG, out_g, out_aux, hnew, live);
    if constexpr (TRAIN_OPS) psum += hnew;              // (0 beyond the sample's length, exactly what hbuf holds there)
    eH += dH; eG += dG;
  }
  if (TRAIN_OPS && a.hsum) {
    // column sums of h over this tile's 16 batch rows: over the 4 rows of a wave by two xor-shuffles, over the 4 waves of a sub-tile through
    // LDS in wave order (fixed order: run-to-run identical)
    float v = psum;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    __syncthreads();                                    // (every wave is past its last read of `red`)
    float* rs = reinterpret_cast<float*>(&red[0][0][0][0]);
    if (lane < 16) rs[wave * 16 + lane] = v;
    __syncthreads();
    if ((wave & 3) == 0 && lane < 16 && (wave >> 2) < MB * NS) {
      const int sb = wave >> 2, mb_ = sb / NS, ns_ = sb % NS;
      const float tot = ((rs[wave * 16 + lane] + rs[(wave + 1) * 16 + lane]) + rs[(wave + 2) * 16 + lane]) + rs[(wave + 3) * 16 + lane];
      const int jj = j0 + ns_ * 16 + lane, tile = bt * MB + mb_;
      if (jj < H && tile < (B + 15) / 16) a.hsum[((long long)dir * ((B + 15) / 16) + tile) * H + jj] = tot;
    }
  }
  PTRACE_DUMP(0);
}

// Gate-derivative math of one (row, unit) pair, shared by the step and the persistent backward kernels.  Contraction is switched OFF:
// left to the compiler, a*b + c fuses or not depending on the surrounding code, and the two kernel families must agree to the bit.
__device__ __forceinline__ void gru_bwd_point(float dh, float r, float z, float n, float hn, float hprev, float (&dgh)[3], float& dpn, float& carry) {
#pragma clang fp contract(off)
  const float dn_ = dh * (1.f - z);
  const float dz = dh * (hprev - n);
  dpn = dn_ * (1.f - n * n);
  const float dr = dpn * hn;
  dgh[0] = dr * r * (1.f - r);
  dgh[1] = dz * z * (1.f - z);
  dgh[2] = dpn * r;                  // d(hn): the n-gate row of dGh
  carry = dh * z;
}
__device__ __forceinline__ void lstm_bwd_point(float dh, float dcar_in, float ig, float fg, float gg, float og, float c, float cprev, float (&dgh)[4],
                                               float& carry) {
#pragma clang fp contract(off)
  const float tc = tanhf_(c);
  const float dc = dcar_in + dh * og * (1.f - tc * tc);
  dgh[0] = dc * gg * ig * (1.f - ig);
  dgh[1] = dc * cprev * fg * (1.f - fg);
  dgh[2] = dc * ig * (1.f - gg * gg);
  dgh[3] = dh * tc * og * (1.f - og);
  carry = dc * fg;
}

// ------------------------------------------------------------------------------------------
// backward step (same grid mapping).  carry[b][j] = sum_k dGh[tq][b][k] * W_hh[k][j]
// ------------------------------------------------------------------------------------------
// Preloaded arguments as in the forward kernel: operand bases + the three HBM-streamed epilogue inputs (saved gates, aux, dy).
//   s_H = s | H << 16;  T_B = T | B << 16;  nbt16_dbg = tiles | flags << 16
// SP (BF = true): the split form of the fp32 mode for shapes no persistent split kernel fits (C4: LSTM H = 1280; c3 / c5: GRU H = 1024) — dGh
// travels as a hi and a lo bf16 plane of the ping-pong buffers (the second plane behind the first: together the size of the fp32 buffers),
// W_hh^T as the hi / lo fragment sets of the split operand, products on the bf16 matrix cores (mfma_packed_split); plain fp32 buffers otherwise.
template <int G, int MB, int NS, bool BF, bool SP = false>
__global__ __launch_bounds__(NW * 64) void rnn_bwd_step_kernel(const float* pk, const float* wp, float* gxbase, float* auxbase, const float* dy,
                                                               int s_H, int T_B, int nbt16_dbg, int lddy, RnnArgs a) {
  static_assert(!SP || BF, "the split form runs on the bf16 data path");
  // bit 2 of the flags: gxbase is really the packed bf16 gate-record buffer (RnnArgs::gates_bf), preloaded in gx's place
  const __bf16* gates_bf = ((nbt16_dbg >> 16) & 4) ? reinterpret_cast<const __bf16*>(gxbase) : nullptr;
  __shared__ __attribute__((aligned(16))) f32x4 red[NW][MB * NS][64];
  constexpr int NTHR = NW * 64;
  constexpr int PAIRS = (MB * NS * 256 + NTHR - 1) / NTHR;
  const int dir = blockIdx.z;
  const int slice = blockIdx.x, bt = blockIdx.y;
  const int s = s_H & 0xffff, H = (int)((unsigned)s_H >> 16);
  const int T = T_B & 0xffff, B = (int)((unsigned)T_B >> 16);
  const int nbt16 = nbt16_dbg & 0xffff, dbg = nbt16_dbg >> 16;
  const int nsl = (H + 15) >> 4;
  const int j0 = slice * (16 * NS), b0 = bt * (16 * MB);   // slice = blockIdx.x = NS consecutive 16-unit slices
  const int nchb = (G * H + kchunk<BF>() - 1) / kchunk<BF>();
  const bool has_q = s > 0;                        // a step was processed before us: its d-gates feed our carry
  const int t = dir == 0 ? T - 1 - s : s;          // reverse of the forward order
  const int tpf = dir == 0 ? t - 1 : t + 1;        // previous step in FORWARD order (h_{prev}, c_{prev})
  const bool has_pf = dir == 0 ? (t > 0) : (t < T - 1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* pk_out = const_cast<float*>(pk) + ((long long)((s & 1) * 2 + dir) * nbt16) * nchb * 256;
  const float* pk_in = pk + ((long long)(((s + 1) & 1) * 2 + dir) * nbt16) * nchb * 256;

  // ---- epilogue operands and store addresses: computed / loaded under the GEMM's operand fetch (see mfma_packed) ------------
  // (no load below depends on a loaded value: the length mask is applied after the GEMM)
  float pg[PAIRS][G], pax[PAIRS], pprev[PAIRS], pdy[PAIRS], pdc[PAIRS];
  int plen[PAIRS];
  bool pact[PAIRS];
  float* gxp[PAIRS];
  long long rowH[PAIRS], bH[PAIRS];
  auto issue_epilogue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int q = threadIdx.x + i * NTHR;
      const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8, mb = sub / NS, ns = sub % NS;
      const int b = b0 + mb * 16 + brow, j = j0 + ns * 16 + jl;
      pact[i] = (sub < MB * NS) && b < B && j < H;
      plen[i] = 0;
      pax[i] = pprev[i] = pdy[i] = pdc[i] = 0.f;
      const long long row = ((long long)t * B + b) * 2 + dir;
      gxp[i] = (gates_bf ? (float*)nullptr : gxbase) + row * G * H + j;   // with packed gate records only the offset is used
      rowH[i] = row * H + j;
      bH[i] = (long long)b * H + j;
#pragma unroll
      for (int g = 0; g < G; ++g) pg[i][g] = 0.f;
      if (pact[i]) {
        if (gates_bf) {                                   // packed record: one 8-byte load (wave-uniform choice, preloaded pointer)
          const bf16x4_ rec = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_*>(gates_bf) + rowH[i]);
          pg[i][0] = (float)rec[0]; pg[i][1] = (float)rec[1]; pg[i][2] = (float)rec[2];
          if (G == 3) pax[i] = (float)rec[3];
          else { pg[i][G - 1] = (float)rec[3]; pax[i] = ldnt(auxbase + rowH[i]); }
        } else {
#pragma unroll
          for (int g = 0; g < G; ++g) pg[i][g] = ldnt(gxp[i] + g * H);
          pax[i] = ldnt(auxbase + rowH[i]);
        }
        pdy[i] = ldnt(&dy[((long long)t * B + b) * lddy + j]);
      }
    }
    // the memory-resident arguments from here on: lengths, the carry of the previous step, h_{prev} / c_{prev}
    // (issuing these loads in FRONT of the operand fetch, as the forward kernel does — the first group alone, or all of them behind an
    // up-front kernel-argument wait — measured no gain here: this kernel is bound by its 290 KB operand fetch, not by epilogue latency)
    hoist_kernargs(a);
    const float* dcar_in = a.dcar + ((long long)(((s + 1) & 1) * 2 + dir)) * B * H;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      if (pact[i]) {
        const int q = threadIdx.x + i * NTHR;
        const int b = b0 + ((q >> 8) / NS) * 16 + ((q >> 4) & 15);
        plen[i] = a.lens[b];
        if (has_q) pdc[i] = dcar_in[bH[i]];
        if (has_pf) {
          const long long prow = ((long long)tpf * B + b) * 2 + dir;
          const int jj = j0 + ((q >> 8) % NS) * 16 + (q & 15);
          pprev[i] = (G == 3) ? a.hbuf[prow * H + jj] : auxbase[prow * H + jj];
        }
      }
    }
  };

  f32x4 acc[MB][NS];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  {
    const int nch_eff = (has_q && !DS2_ABLATE_BIT(dbg, 1)) ? nchb : 0;  // one code path, see the forward kernel
    const float* pa = pk_in + ((long long)(bt * MB) * nchb) * 256 + lane * 4;
    const float* pw = wp + (((long long)dir * nsl + slice * NS) * nchb) * 256 + lane * 4;
    if constexpr (SP)
      mfma_packed_split<MB, NS, (MB * NS > 2 ? 3 : 4)>(acc, nch_eff, wave, pa, (long long)nchb * 256, (long long)4 * nbt16 * nchb * 256, pw,
                                                       (long long)nchb * 256, (long long)2 * nsl * nchb * 256, issue_epilogue_loads);
    else
      mfma_packed<BF, MB, NS, (MB * NS > 2 ? 4 : 6)>(acc, nch_eff, wave, pa, (long long)nchb * 256, pw, (long long)nchb * 256, issue_epilogue_loads);
  }
  float* dcar_out = a.dcar + ((long long)((s & 1) * 2 + dir)) * B * H;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int n = 0; n < NS; ++n) red[wave][i * NS + n][lane] = acc[i][n];
  __syncthreads();
  if (DS2_ABLATE_BIT(dbg, 2)) return;

#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    if (!pact[i]) continue;
    const int q = threadIdx.x + i * NTHR;
    const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8;
    const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
    const int b = b0 + (sub / NS) * 16 + brow, j = j0 + (sub % NS) * 16 + jl;
    float* gx = gxp[i];
    float* ax = auxbase + rowH[i];
    float* dco = dcar_out + bH[i];
    float dgh[G];
    __bf16* gb = a.dgx_bf ? a.dgx_bf + (gxp[i] - (gates_bf ? (float*)nullptr : gxbase)) : nullptr;   // wave-uniform dGx destination
    if (!(t < plen[i])) {
#pragma unroll
      for (int g = 0; g < G; ++g) { dgx_store(gx, gb, g * H, 0.f); dgh[g] = 0.f; }
      if (G == 3) stnt(ax, 0.f);
      *dco = 0.f;
    } else {
      float carry = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) carry += red[w][sub][src_lane][reg];
      if constexpr (G == 3) {
        const float dh = pdy[i] + carry + pdc[i];
        float dpn, car;
        gru_bwd_point(dh, pg[i][0], pg[i][1], pg[i][2], pax[i], pprev[i], dgh, dpn, car);
        dgx_store(gx, gb, 0, dgh[0]); dgx_store(gx, gb, H, dgh[1]); dgx_store(gx, gb, 2 * H, dpn);
        stnt(ax, dgh[2]);
        *dco = car;
      } else {
        const float dh = pdy[i] + carry;
        float car;
        lstm_bwd_point(dh, pdc[i], pg[i][0], pg[i][1], pg[i][2], pg[i][G - 1], pax[i], pprev[i], dgh, car);
#pragma unroll
        for (int g = 0; g < G; ++g) dgx_store(gx, gb, g * H, dgh[g]);
        *dco = car;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long pi = packed_index<BF>(b, g * H + j, nchb);
      packed_store<BF>(pk_out, pi, dgh[g]);
      if constexpr (SP) packed_store<true>(pk_out + (long long)4 * nbt16 * nchb * 256, pi, dgh[g] - (float)(__bf16)dgh[g]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// PERSISTENT backward (bf16 training path: packed gate records in, bf16 dGx out): the same construction as the persistent forward
// kernel.  Per step a CU then moves only the moving operand (dGh_{t+1}: 16 rows x G*H bf16 = 98 KB at H = 1024) instead of 290 KB, the
// W_hh^T slice (32 units x G*H bf16 = 192 KB) lives in registers (96 per lane), there is no launch boundary, and the carry
// (dh*z / dc*f) never leaves its thread.  Exchange protocol, buffers, starvation handling: see rnn_fwd_persistent_kernel.
// ------------------------------------------------------------------------------------------
// SP: the split form of the fp32 mode (see rnn_fwd_persistent_kernel): dGh_t and W_hh^T as hi + lo bf16 planes, three bf16 MFMAs per product.
template <int G, int MB, int NS, int NCW, bool BF, bool SP = false>
__global__ __launch_bounds__(NW * 64) void rnn_bwd_persistent_kernel(RnnArgs a, char* xbuf, unsigned* census, int spin_limit) {
  static_assert(MB * NS * 256 <= NW * 64, "one (row, unit) pair per thread");
  static_assert(!SP || BF, "the split form runs on the bf16 data path");
  constexpr int NPL = SP ? 2 : 1;                                             // operand planes: [hi | lo]
  __shared__ __attribute__((aligned(16))) f32x4 red[2][NW][MB * NS][64];      // double-buffered: one workgroup barrier per time step
  constexpr int KC = kchunk<BF>(), EPL = KC / 4, NPUB = 64 / EPL;             // units per chunk / per 16-byte lane vector; chunks per wave and gate
  using elem_t = typename std::conditional<BF, __bf16, float>::type;
  __shared__ __attribute__((aligned(16))) elem_t stage[NPL][NW][G][64];        // wave-private: NPUB complete 16-byte chunks per gate
  const PRole role = persist_role(a, census, spin_limit, 2);                  // 1-D grid; roles from the XCD census or the workgroup id
  if (!role.active) return;
  const int dir = role.dir, bt = role.bt, slice = role.slice;
  const bool l2_local = role.local != 0;
  const int T = a.T, B = a.B, H = a.H, lddy = a.lddy;
  const int nsl = (H + 15) >> 4, nchb = (G * H + KC - 1) / KC;
  const int j0 = slice * (16 * NS), b0 = bt * (16 * MB);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long planebytes = (long long)2 * a.nbt16 * nchb * 1024;
  const long long bufbytes = NPL * planebytes;
  const long long dirbase = (long long)dir * a.nbt16 * nchb * 1024;

  f32x4 wreg[NPL][NCW][NS];
  bool cval[NCW], lval[NCW];
  const long long wplane = (long long)2 * nsl * nchb * 256;                   // floats of one packed backward operand
#pragma unroll
  for (int k = 0; k < NCW; ++k) {
    const int c = wave + NW * k;
    cval[k] = c < nchb;
    lval[k] = cval[k] && (c * KC + (lane >> 4) * EPL) < G * H;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int n = 0; n < NS; ++n)
        wreg[pl][k][n] = cval[k] ? *reinterpret_cast<const f32x4*>(a.wp + pl * wplane + ((((long long)dir * nsl + slice * NS + n) * nchb + c) * 256) + lane * 4)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int q = threadIdx.x;
  const int jl = q & 15, brow = (q >> 4) & 15, sub = q >> 8, mb = sub / NS, ns = sub % NS;
  const int b = b0 + mb * 16 + brow, j = j0 + ns * 16 + jl;
  const bool pair = sub < MB * NS;
  const bool pact = pair && b < B && j < H;
  const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
  const int red_sw = (lane >> 3) & 1, red_reg = (reg + 2 * ((jl >> 3) & 1)) & 3;      // conflict-free dword reads of the partial sums (see forward)
  const int plen = pact ? a.lens[b] : 0;
  const __bf16* gates_bf = a.gates_bf;
  float dcar = 0.f;                                                           // GRU dh*z / LSTM dc*f of the step before (own pair)

  // operands of the gate-derivative math for one time step: independent of the recurrence, so they are fetched one step ahead
  // (the packed record stays RAW until the step that uses it: converting it here would make the compiler wait for the load on the spot,
  //  i.e. put an HBM round trip into every time step — it did, 0.6 us of the first version's 3.6)
  struct Ops { bf16x4_ rec; float g0, g1, g2, g3, ax, dy, prev; };
  auto fetch = [&](int step) {
    Ops o{bf16x4_{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!pact) return o;
    const int t = dir == 0 ? T - 1 - step : step;
    const long long rowH = (((long long)t * B + b) * 2 + dir) * H + j;
    if (gates_bf) {                                       // packed 8-byte record (bf16 training path)
      o.rec = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_*>(gates_bf) + rowH);
      if (G == 4) o.ax = ldnt(a.aux + rowH);
    } else {                                              // plain buffers: gates in gx (overwritten with dGx at this row's own step), hn / c in aux
      const float* gp = a.gx + (((long long)t * B + b) * 2 + dir) * G * H + j;
      o.g0 = ldnt(gp); o.g1 = ldnt(gp + H); o.g2 = ldnt(gp + 2 * H);
      if (G == 4) { o.g3 = ldnt(gp + 3 * H); o.ax = ldnt(a.aux + rowH); }
      else o.g3 = ldnt(a.aux + rowH);
    }
    o.dy = ldnt(&a.dy[((long long)t * B + b) * lddy + j]);
    const int tpf = dir == 0 ? t - 1 : t + 1;
    if (dir == 0 ? (t > 0) : (t < T - 1)) {
      const long long prow = (((long long)tpf * B + b) * 2 + dir) * H + j;
      o.prev = (G == 3) ? a.hbuf[prow] : a.aux[prow];
    }
    return o;
  };
  Ops cur = fetch(0), nxt = cur;

  // A wave's 64 threads are 4 batch rows x 16 units: per gate NPUB complete 16-byte chunks of the packed dGh buffer.  Lane p < NPUB * G
  // publishes (and resets) chunk (gate p / NPUB, row (wave & 3) * 4 + (p & 3), unit group (p % NPUB) >> 2) of this wave's (tile, slice) = wave >> 2.
  const int pg = lane / NPUB, pp = lane % NPUB;
  const bool pub_lane = lane < NPUB * G && (wave >> 2) < MB * NS;
  long long pub_off = 0;
  {
    const int wsub = wave >> 2, ku = pg * H + j0 + (wsub % NS) * 16 + (pp >> 2) * EPL;      // first k of this lane's unit group in gate pg
    pub_off = dirbase + ((((long long)(bt * MB + wsub / NS) * nchb + ku / KC) * 64) + ((ku % KC) / EPL) * 16 + (wave & 3) * 4 + (pp & 3)) * 16;
  }

  // results of one time step (written one step late, see the loop)
  int so_t = 0;
  float so_dgx[G], so_dax = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) so_dgx[g] = 0.f;
  auto store_results = [&](int t, const float (&dgx)[G], float dax) {
    if (!pact) return;
    const long long row = ((long long)t * B + b) * 2 + dir;
    if (a.dgx_bf) {
      __bf16* gb = a.dgx_bf + row * G * H + j;
#pragma unroll
      for (int g = 0; g < G; ++g) __builtin_nontemporal_store((__bf16)dgx[g], gb + g * H);
    } else {
      float* gp = a.gx + row * G * H + j;                  // in place: the gates of this row were consumed (fetched two steps ago)
#pragma unroll
      for (int g = 0; g < G; ++g) stnt(gp + g * H, dgx[g]);
    }
    if (G == 3) {
      stnt(a.aux + row * H + j, dax);
      if (a.dhn_bf) __builtin_nontemporal_store((__bf16)dax, a.dhn_bf + row * H + j);
    }
  };
  float bs[4] = {0.f, 0.f, 0.f, 0.f};                       // this thread's (batch row, unit) sums over time: the bias gradients' partials

  static_assert(NPL * NCW * MB <= 32, "pending mask is 32 bits");
  unsigned goff[NPL * NCW * MB], pend0 = 0;                 // this wave's chunks of the packed exchange buffer, and the all-pending mask
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int k = 0; k < NCW; ++k)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        goff[(pl * NCW + k) * MB + i] = (unsigned)(pl * planebytes + ((((long long)(bt * MB + i) * nchb + (wave + NW * k)) * 64) + lane) * 16);
        if (cval[k]) pend0 |= 1u << ((pl * NCW + k) * MB + i);
      }

  PTRACE_DECL;
  vm_drained();                                         // prologue loads have landed
  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? T - 1 - s : s;
    PTRACE(0);
    f32x4 acc[MB][NS];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int n = 0; n < NS; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const char* xin = xbuf + (long long)((s - 1) & 3) * bufbytes + dirbase;
      u32x4_ av[NPL * NCW * MB];
#pragma unroll
      for (int k = 0; k < NPL * NCW * MB; ++k) av[k] = u32x4_{0u, 0u, 0u, 0u};
      int spins = 0;
      unsigned pend = pend0;                              // wave-uniform: chunks of this wave that have not been seen complete yet
      while (pend) {
        // only the chunks that are still missing are read again (one asm statement per pass: poll_pass)
        poll_pass<NPL * NCW * MB>(av, goff, xin, pend);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int k = 0; k < NCW; ++k)
#pragma unroll
            for (int i = 0; i < MB; ++i)
              if (pend & (1u << ((pl * NCW + k) * MB + i))) {
                const u32x4_ c = av[(pl * NCW + k) * MB + i];
                const bool ok = !lval[k] || (c.x != PSENT && c.y != PSENT && c.z != PSENT && c.w != PSENT);
                if (__ballot(ok) == ~0ull) pend &= ~(1u << ((pl * NCW + k) * MB + i));
              }
        pend = __builtin_amdgcn_readfirstlane(pend);
        if (pend && ++spins > spin_limit) {
          // record who starved and on what (first failure only); the host raises at the step's sync point
          if (lane == 0 && atomicCAS(&a.status[0], 0, 2) == 0) {
            a.status[1] = slice; a.status[2] = bt; a.status[3] = dir; a.status[4] = s; a.status[5] = wave;
            a.status[6] = (int)pend; a.status[7] = l2_local;
            __threadfence_system();
          }
          return;
        }
      }
#pragma unroll
      for (int k = 0; k < NCW; ++k)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const u32x4_ v = lval[k] ? av[k * MB + i] : u32x4_{0u, 0u, 0u, 0u};
          if constexpr (SP) {
            const u32x4_ vl = lval[k] ? av[(NCW + k) * MB + i] : u32x4_{0u, 0u, 0u, 0u};
#pragma unroll
            for (int n = 0; n < NS; ++n) {                  // smallest terms first: lo.hi, hi.lo, then hi.hi
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vl), __builtin_bit_cast(bf16x8, wreg[0][k][n]), acc[i][n], 0, 0, 0);
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[NPL - 1][k][n]), acc[i][n], 0, 0, 0);
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[0][k][n]), acc[i][n], 0, 0, 0);
            }
          } else if constexpr (BF) {
#pragma unroll
            for (int n = 0; n < NS; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, wreg[0][k][n]), acc[i][n], 0, 0, 0);
          } else {
            const f32x4 vf = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int n = 0; n < NS; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], wreg[0][k][n][e], acc[i][n], 0, 0, 0);
          }
        }
    }
    vm_drained();                                       // (the gather has waited for everything; tell the compiler)
    PTRACE(1);
    if (s > 0) cur = nxt;                               // operands of THIS step: fetched one step ago, landed
    if (gates_bf) { cur.g0 = (float)cur.rec[0]; cur.g1 = (float)cur.rec[1]; cur.g2 = (float)cur.rec[2]; cur.g3 = (float)cur.rec[3]; }
    // ---- HBM traffic of the step goes out HERE, right behind the gather (see the forward kernel): the previous step's results, and the
    // operands of the next step's gate-derivative math
    const bool more = s + 1 < T;
    if (s > 0) store_results(so_t, so_dgx, so_dax);
    if (more) nxt = fetch(s + 1);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int n = 0; n < NS; ++n) {                      // (halves swapped in the lanes with bit 3 set: rnn_fwd_persistent_kernel's red_reg)
        float* rp = reinterpret_cast<float*>(&red[s & 1][wave][i * NS + n][lane]);
        *reinterpret_cast<f32x2r*>(rp + 2 * red_sw) = f32x2r{acc[i][n][0], acc[i][n][1]};
        *reinterpret_cast<f32x2r*>(rp + 2 - 2 * red_sw) = f32x2r{acc[i][n][2], acc[i][n][3]};
      }
    PTRACE(2);
    __syncthreads();
    PTRACE(3);

    float dgh[G], dgx[G], dax = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) { dgh[g] = 0.f; dgx[g] = 0.f; }
    if (pact && t < plen) {
      float carry = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) carry += red[s & 1][w][sub][src_lane][red_reg];
      if constexpr (G == 3) {
        const float dh = cur.dy + carry + dcar;
        float dpn;
        gru_bwd_point(dh, cur.g0, cur.g1, cur.g2, cur.g3, cur.prev, dgh, dpn, dcar);
        dgx[0] = dgh[0]; dgx[1] = dgh[1]; dgx[2] = dpn;
        dax = dgh[2];
      } else {
        const float dh = cur.dy + carry;
        float car;
        lstm_bwd_point(dh, dcar, cur.g0, cur.g1, cur.g2, cur.g3, cur.ax, cur.prev, dgh, car);
        dcar = car;
#pragma unroll
        for (int g = 0; g < G; ++g) dgx[g] = dgh[g];
      }
    } else {
      dcar = 0.f;
    }
    PTRACE(4);
    // ---- publish dGh_s (see the forward kernel): wave-local assembly, publish, reset two steps ahead (the previous reset was acknowledged
    // inside this step's gather)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      stage[0][wave][g][lane] = (elem_t)dgh[g];
      if constexpr (SP) stage[NPL - 1][wave][g][lane] = (elem_t)(dgh[g] - (float)(elem_t)dgh[g]);
    }
    if (pub_lane) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        store16_x(xbuf + (long long)(s & 3) * bufbytes + pl * planebytes + pub_off,
                  *reinterpret_cast<const u32x4_*>(&stage[pl][wave][pg][(pp & 3) * 16 + (pp >> 2) * EPL]), l2_local);
        store16_x(xbuf + (long long)((s + 2) & 3) * bufbytes + pl * planebytes + pub_off, u32x4_{PSENT, PSENT, PSENT, PSENT}, l2_local);
      }
    }
    PTRACE(5);
    so_t = t; so_dax = dax;
#pragma unroll
    for (int g = 0; g < G; ++g) { so_dgx[g] = dgx[g]; bs[g] += dgx[g]; }
    if (G == 3) bs[3] += dax;
  }
  store_results(so_t, so_dgx, so_dax);                      // the last step's results
  if (a.bsum && pact) {
    float* o = a.bsum + (((long long)b * 2 + dir) * 4) * H + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) o[g * H] = bs[g];
  }
  PTRACE_DUMP(1);
}

// W_hh (2, G*H, H) -> fwd-packed [2][nsl][G][nch][64 lanes][16 B] and bwd-packed [2][nsl][nchb][64 lanes][16 B]
template <bool BF>
__global__ __launch_bounds__(256) void rnn_pack_kernel(const float* __restrict__ whh, void* __restrict__ wpf, void* __restrict__ wpb, int G, int H) {
  constexpr int KC = kchunk<BF>();
  constexpr int EPL = BF ? 8 : 4;                 // elements per lane = one 16-byte store per thread
  const int nsl = (H + 15) >> 4, nch = (H + KC - 1) / KC, nchb = (G * H + KC - 1) / KC;
  const long long nf = (long long)2 * nsl * G * nch * 64, nb = (long long)2 * nsl * nchb * 64;     // lane vectors
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += stride) {
    const bool fwd = i < nf;
    const long long ii = fwd ? i : i - nf;
    const int lane = (int)(ii & 63);
    long long r = ii >> 6;
    float v[EPL];
    if (fwd) {
      const int c = r % nch; r /= nch;
      const int g = r % G; r /= G;
      const int slice = r % nsl, dir = r / nsl;
      const int j = slice * 16 + (lane & 15), k0 = c * KC + (lane >> 4) * EPL;
      const float* src = whh + ((long long)dir * G * H + g * H + j) * H + k0;          // EPL consecutive k of row (g, j)
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[e] = (j < H && k0 + e < H) ? src[e] : 0.f;
    } else {
      const int c = r % nchb; r /= nchb;
      const int slice = r % nsl, dir = r / nsl;
      const int j = slice * 16 + (lane & 15), k0 = c * KC + (lane >> 4) * EPL;          // k = gate-unit row of W_hh
      const float* src = whh + ((long long)dir * G * H + k0) * H + j;                  // EPL rows, column j
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[e] = (j < H && k0 + e < G * H) ? src[(long long)e * H] : 0.f;
    }
    if constexpr (BF) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
      reinterpret_cast<bf16x8*>(fwd ? wpf : wpb)[ii] = o;
    } else {
      reinterpret_cast<f32x4*>(fwd ? wpf : wpb)[ii] = f32x4{v[0], v[1], v[2], v[3]};
    }
  }
}

// forward operand of the split persistent kernel: the bf16 fragment order of rnn_pack_kernel<true>, once for hi = bf16(w) and once for
// lo = bf16(w - hi)
__global__ __launch_bounds__(256) void rnn_pack_split_fwd_kernel(const float* __restrict__ whh, void* __restrict__ wp_hi, void* __restrict__ wp_lo, int G, int H) {
  constexpr int KC = 32;
  const int nsl = (H + 15) >> 4, nch = (H + KC - 1) / KC;
  const long long nf = (long long)2 * nsl * G * nch * 64;
  for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < nf; ii += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(ii & 63);
    long long r = ii >> 6;
    const int c = r % nch; r /= nch;
    const int g = r % G; r /= G;
    const int slice = r % nsl, dir = r / nsl;
    const int j = slice * 16 + (lane & 15), k0 = c * KC + (lane >> 4) * 8;
    const float* src = whh + ((long long)dir * G * H + g * H + j) * H + k0;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (j < H && k0 + e < H) ? src[e] : 0.f;
      hi[e] = (__bf16)v;
      lo[e] = (__bf16)(v - (float)hi[e]);
    }
    reinterpret_cast<bf16x8*>(wp_hi)[ii] = hi;
    reinterpret_cast<bf16x8*>(wp_lo)[ii] = lo;
  }
}

// backward operand of the split persistent kernel: the bf16 fragment order of rnn_pack_kernel<true>'s backward half, hi and lo
__global__ __launch_bounds__(256) void rnn_pack_split_bwd_kernel(const float* __restrict__ whh, void* __restrict__ wp_hi, void* __restrict__ wp_lo, int G, int H) {
  constexpr int KC = 32;
  const int nsl = (H + 15) >> 4, nchb = (G * H + KC - 1) / KC;
  const long long nb = (long long)2 * nsl * nchb * 64;
  for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < nb; ii += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(ii & 63);
    long long r = ii >> 6;
    const int c = r % nchb; r /= nchb;
    const int slice = r % nsl, dir = r / nsl;
    const int j = slice * 16 + (lane & 15), k0 = c * KC + (lane >> 4) * 8;
    const float* src = whh + ((long long)dir * G * H + k0) * H + j;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (j < H && k0 + e < G * H) ? src[(long long)e * H] : 0.f;
      hi[e] = (__bf16)v;
      lo[e] = (__bf16)(v - (float)hi[e]);
    }
    reinterpret_cast<bf16x8*>(wp_hi)[ii] = hi;
    reinterpret_cast<bf16x8*>(wp_lo)[ii] = lo;
  }
}

inline int pick_mb(int B, int H) {
  const int nsl = ceil_div(H, 16);
  if (B <= 16) return 1;
  // 16-row tiles that would not all be resident at once (> 256 CUs -> a second, mostly empty round doubles the step latency):
  // take 32-row tiles instead.  Otherwise 32-row tiles (half the W_hh re-reads) only when they still fill the chip.
  if ((long long)nsl * ceil_div(B, 16) * 2 > 256) return 2;
  return ((long long)nsl * ceil_div(B, 32) * 2 >= 200) ? 2 : 1;
}

// Which persistent kernels may be used, the cooldown after a starved launch, what the last call took: all of it lives in the CALLER's
// ds2_rnn_ctx (include/ds2hip.h) — the library keeps no mutable global state.  After a starved launch (ds2_rnn_persistent_status) the next
// ctx->rearm_calls recurrence calls through that context take the one-launch-per-step kernels, then the persistent kernels are armed again: a
// transient (another process or stream holding CUs for a moment) costs a few slow steps, not the rest of the run.  The backward kernel must
// be switched off (ds2_rnn_persistent_enable) by a caller that runs collectives on another stream during backward, because a persistent
// launch needs every one of its workgroups resident at once.  No context: no persistent launch.
// ds2_rnn_ctx.ws_prearmed is a one-shot promise about the workspace of the NEXT recurrence call through the context: read and cleared at entry
int take_prearmed(ds2_rnn_ctx* c) {
  if (!c) return 0;
  const int v = c->ws_prearmed;
  c->ws_prearmed = 0;
  return v;
}
bool persist_allowed(ds2_rnn_ctx* c, bool bwd) {
  if (!c || !c->status_dev) return false;
  if (c->cooldown != 0) {
    if (c->cooldown > 0) --c->cooldown;
    return false;
  }
  return bwd ? c->persist_bwd != 0 : c->persist_fwd != 0;
}
bool persist_idle(const ds2_rnn_ctx* c, bool bwd) { return c && c->status_dev && c->cooldown == 0 && (bwd ? c->persist_bwd : c->persist_fwd) != 0; }

// bytes of ONE packed h buffer of the forward recurrence ([2 dirs][tiles][chunks][1 KiB]); the persistent kernel uses four
size_t fwd_xbuf_bytes(int B, int H, int bf16) { return (size_t)2 * (ceil_div(B, 32) * 2) * ceil_div(H, bf16 ? 32 : 16) * 1024; }

int cu_count() {                                          // of the CURRENT device (cached per device id: a process may drive two GPUs)
  static int n[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!n[dev]) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) n[dev] = p.multiProcessorCount;
    if (n[dev] <= 0) n[dev] = 1;
  }
  return n[dev];
}

// XCD-local exchange (persist_role's census mode): possible when every exchange group (gs workgroups) fits one XCD and the groups fit the
// chip, on the 8 x 32-CU part this library is written for.  DS2_RNN_XCD_LOCAL=0 keeps the placement-independent sc1 protocol (A/B runs).
constexpr int XCDS = 8, CUS_PER_XCD = 32;
bool xcd_local_fits(int gs, int ngroups) {
  static const char* env = getenv("DS2_RNN_XCD_LOCAL");
  if (env && env[0] == '0') return false;
  return cu_count() == XCDS * CUS_PER_XCD && gs >= 1 && gs <= CUS_PER_XCD && ngroups <= XCDS * (CUS_PER_XCD / gs);
}
constexpr size_t CENSUS_BYTES = 64;                       // 8 per-XCD slot counters, the arrival counter, one "unexpected XCC id" flag

// Forward recurrence in one persistent launch (bf16 operands).  Returns 1 if launched, 0 if the shape / device does not qualify
// (the caller then runs the step kernels), < 0 on error.
template <int G, bool BF, bool SP = false>
int try_launch_persistent_fwd(RnnArgs a, hipStream_t st) {
  constexpr int NPL = SP ? 2 : 1;
  constexpr int RLIM = SP ? 200 : 180;                            // lane-vector budget (x 4 registers) of W_hh fragments + gathered operand
  static const char* env = getenv("DS2_RNN_PERSISTENT");          // "0" = always the step kernels (A/B runs, debugging)
  if (env && env[0] == '0') return 0;
  if (a.dbg & ~128) return 0;                                     // any selector but 128 (= no K-split backward) selects the step kernels
  if ((a.H % 16) != 0 || a.T < 2 || !persist_allowed(a.hctx, false)) return 0;
  const int nsl = a.H / 16;
  const int nch = ceil_div(a.H, kchunk<BF>());
  int ncw = ceil_div(nch, NW);
  if (!BF) ncw = ceil_div(ncw, 2) * 2;                            // fp32 instantiations: 2, 4, 6, 8 chunks per wave
  if (ncw > (BF ? 5 : 8)) return 0;
  // Tile shape.  The exchange is what a time step costs, and it is priced by the size of the gather and the number of producers
  // (64 producers / 64 KB: 2.64 us, 32 / 32 KB: 1.6-1.7 us, profiles/r01_probe_xcd_exchange.txt) — the W_hh slice sits in registers either
  // way.  So: 16 batch rows x 32 units per workgroup where the registers allow it (W_hh fragments ncw * 2G + operand ncw lane vectors of
  // 4 VGPRs) and the alternative would be a 32-row tile; else the step kernels' 16|32 rows x 16 units (with 16-row tiles the gather
  // is the same size either way and twice as many, half as big workgroups measured slightly faster).
  int mb = 1, ns = 2;
  const bool ns2_ok = (nsl % 2) == 0 && ncw * NPL * (2 * G + 1) * 4 <= RLIM && (long long)(nsl / 2) * ceil_div(a.B, 16) * 2 <= cu_count();
  // ... and 16 x 32 also where 16 x 16 would be chosen but only the wider slice lets an exchange group fit one XCD (L2-local exchange;
  // bf16 only: in fp32 the doubled MFMA instruction count per workgroup costs more than the exchange saves, c2 5.4 -> 6.1 us per step)
  const bool ns2_for_local = BF && ns2_ok && !xcd_local_fits(nsl, ceil_div(a.B, 16 * pick_mb(a.B, a.H)) * 2) && xcd_local_fits(nsl / 2, ceil_div(a.B, 16) * 2);
  if (!ns2_ok || (pick_mb(a.B, a.H) != 2 && !ns2_for_local)) {
    ns = 1;
    mb = pick_mb(a.B, a.H);
    if (ncw * NPL * (G + mb) * 4 > (SP ? RLIM : 176)) return 0;
  }
  const int nbt = ceil_div(a.B, 16 * mb);
  // every workgroup must be resident at once: one per CU (up to 160 KB of registers + up to 129 KB of LDS each)
  if ((long long)(nsl / ns) * nbt * 2 > cu_count()) return 0;
  a.nsl = nsl;
  a.nbt16 = ceil_div(a.B, 32) * 2;
  a.p_nbt = nbt; a.p_gs = nsl / ns; a.p_cux = CUS_PER_XCD;
  a.p_census = xcd_local_fits(a.p_gs, 2 * nbt) ? 1 : 0;
  char* xbuf = reinterpret_cast<char*>(a.pk);
  const size_t xbytes = 4 * NPL * fwd_xbuf_bytes(a.B, a.H, BF ? 1 : 0);
  if (!a.prearmed) DS2_HIP(hipMemsetAsync(xbuf, 0xff, xbytes + CENSUS_BYTES, st));      // every 16-byte chunk = the "not yet published" sentinel; census words = -1
  unsigned* census = reinterpret_cast<unsigned*>(xbuf + xbytes);
  // 1-D grid.  Census mode: one workgroup per CU of the whole chip, roles by real XCD id (surplus workgroups exit); else exactly the workgroups needed
  dim3 grid(a.p_census ? cu_count() : a.p_gs * nbt * 2), block(NW * 64);
  static const char* sl = getenv("DS2_RNN_SPIN_LIMIT");
  const int spin_limit = sl ? atoi(sl) : (1 << 20);              // ~1 s of polling: a missing workgroup is reported instead of hanging the queue
#define DS2_PLAUNCH(MB_, NS_, NCW_)                                                                                                   \
  do {                                                                                                                                \
    if constexpr (NCW_ * NPL * (NS_ * G + MB_) * 4 <= RLIM)                                                                           \
      hipLaunchKernelGGL((rnn_fwd_persistent_kernel<G, MB_, NS_, NCW_, BF, SP>), grid, block, 0, st, a, xbuf, census, spin_limit);             \
    else                                                                                                                              \
      return 0;                                                                                                                       \
  } while (0)
#define DS2_PCASE(NCW_) case NCW_: if (ns == 2) DS2_PLAUNCH(1, 2, NCW_); else if (mb == 2) DS2_PLAUNCH(2, 1, NCW_); else DS2_PLAUNCH(1, 1, NCW_); break;
  if constexpr (BF) {
    switch (ncw) {
      DS2_PCASE(1) DS2_PCASE(2) DS2_PCASE(3) DS2_PCASE(4) DS2_PCASE(5)
      default: return 0;
    }
  } else {
    switch (ncw) {
      DS2_PCASE(2) DS2_PCASE(4) DS2_PCASE(6) DS2_PCASE(8)
      default: return 0;
    }
  }
#undef DS2_PCASE
#undef DS2_PLAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn persistent launch failed: %s", hipGetErrorString(e));
  return 1;
}

size_t bwd_xbuf_bytes(int gates, int B, int H, int bf16) { return (size_t)2 * (ceil_div(B, 32) * 2) * ceil_div(gates * H, bf16 ? 32 : 16) * 1024; }

// Backward recurrence in one persistent launch (bf16 training path).  1 = launched, 0 = not eligible, < 0 = error.
template <int G, bool BF, bool SP = false>
int try_launch_persistent_bwd(RnnArgs a, hipStream_t st) {
  constexpr int NPL = SP ? 2 : 1;
  static const char* env = getenv("DS2_RNN_PERSISTENT");
  if ((env && env[0] == '0') || (a.dbg & ~128)) return 0;
  // buffers: either the bf16 training path's (packed gate records in, bf16 dGx out) or the plain ones (gates in gx, dGx in place)
  if (!((a.gates_bf && a.dgx_bf) || (!a.gates_bf && !a.dgx_bf && a.gx))) return 0;
  if ((a.H % 16) != 0 || a.T < 2 || !persist_allowed(a.hctx, true)) return 0;
  int mb = pick_mb(a.B, a.H);
  const int nsl = a.H / 16;
  const int nchb = ceil_div(G * a.H, kchunk<BF>());
  const int q = BF ? 3 : 6;                                           // instantiated chunks per wave: bf16 3, 6, 9, 12 ; fp32 6, 12, 18
  const int ncw = ceil_div(ceil_div(nchb, NW), q) * q;
  int ns = (mb == 2 && (nsl % 2) == 0) ? 2 : 1;                       // same tile choice as the step kernels ...
  // ... and 16 rows x 32 units also where that is what lets an exchange group fit one XCD (L2-local exchange)
  if (BF && ns == 1 && mb == 1 && (nsl % 2) == 0 && ncw * NPL * (2 + 1) * 4 <= (SP ? 160 : 192) && !xcd_local_fits(nsl, ceil_div(a.B, 16) * 2) &&
      xcd_local_fits(nsl / 2, ceil_div(a.B, 16) * 2))
    ns = 2;
  if (ns == 2) mb = 1;
  const int nbt = ceil_div(a.B, 16 * mb);
  if (ncw > (BF ? 12 : 18) || ncw * NPL * (ns + mb) * 4 > (SP ? 160 : 192) || NPL * ncw * mb > 32) return 0;   // W_hh^T fragments + operand lane vectors must fit the registers
  if ((long long)(nsl / ns) * nbt * 2 > cu_count()) return 0;
  a.nsl = nsl;
  a.nbt16 = ceil_div(a.B, 32) * 2;
  a.p_nbt = nbt; a.p_gs = nsl / ns; a.p_cux = CUS_PER_XCD;
  a.p_census = xcd_local_fits(a.p_gs, 2 * nbt) ? 1 : 0;
  char* xbuf = reinterpret_cast<char*>(a.pk);
  const size_t xbytes = 4 * NPL * bwd_xbuf_bytes(G, a.B, a.H, BF ? 1 : 0);
  if (!a.prearmed) DS2_HIP(hipMemsetAsync(xbuf, 0xff, xbytes + CENSUS_BYTES, st));
  unsigned* census = reinterpret_cast<unsigned*>(xbuf + xbytes);
  dim3 grid(a.p_census ? cu_count() : a.p_gs * nbt * 2), block(NW * 64);   // see the forward launcher
  static const char* sl = getenv("DS2_RNN_SPIN_LIMIT");
  const int spin_limit = sl ? atoi(sl) : (1 << 20);
#define DS2_PB(MB_, NS_, NCW_)                                                                                                      \
  do {                                                                                                                              \
    if constexpr (NCW_ * NPL * (NS_ + MB_) * 4 <= (SP ? 160 : 192) && NPL * NCW_ * MB_ <= 32)                                                   \
      hipLaunchKernelGGL((rnn_bwd_persistent_kernel<G, MB_, NS_, NCW_, BF, SP>), grid, block, 0, st, a, xbuf, census, spin_limit);           \
    else                                                                                                                            \
      return 0;                                                                                                                     \
  } while (0)
#define DS2_PBCASE(NCW_) case NCW_: if (ns == 2) DS2_PB(1, 2, NCW_); else if (mb == 2) DS2_PB(2, 1, NCW_); else DS2_PB(1, 1, NCW_); break;
  if constexpr (BF) {
    switch (ncw) {
      DS2_PBCASE(3) DS2_PBCASE(6) DS2_PBCASE(9) DS2_PBCASE(12)
      default: return 0;
    }
  } else {
    switch (ncw) {
      DS2_PBCASE(6) DS2_PBCASE(12) DS2_PBCASE(18)
      default: return 0;
    }
  }
#undef DS2_PBCASE
#undef DS2_PB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn persistent backward launch failed: %s", hipGetErrorString(e));
  return 1;
}

// 16-byte exchange store with a wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit offset: no 64-bit vector address arithmetic per store
// (s_nop: see store16_sc1)
template <bool L2_LOCAL>
__device__ __forceinline__ void store16_base(const char* base, unsigned off, u32x4_ v) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(base);
  const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)), blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned long long sb = ((unsigned long long)bhi << 32) | (unsigned long long)blo;
  if (L2_LOCAL) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 2" ::"v"(off), "v"(v), "s"(sb) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 2" ::"v"(off), "v"(v), "s"(sb) : "memory");
}

#include "rnn_bwd_ksplit.h"
#include "rnn_fwd_u10.h"

template <int G, bool BF, bool SP = false>
int launch_steps(bool bwd, RnnArgs a, hipStream_t st) {
  a.dbg = a.hctx ? a.hctx->debug_flags : 0;
  int mb = pick_mb(a.B, a.H);
  a.nsl = ceil_div(a.H, 16);
  // backward: the moving operand (dGh, G*H wide, FRESH from the previous launch = through the fabric) costs about twice as much
  // per byte as the L2-resident W_hh^T slice (profiles/r01_probe_l2_residency.txt), so where the forward kernel takes 32 batch
  // rows x 16 units per workgroup, the backward kernel takes 16 rows x 32 units: half the fresh bytes, twice the cached ones.
  const int ns = (mb == 2 && (a.nsl % 2) == 0 && ((bwd && !(a.dbg & 8)) || (!bwd && (a.dbg & 16)))) ? 2 : 1;
  if (ns == 2) mb = 1;
  const int nbt = ceil_div(a.B, 16 * mb);
  a.nbt16 = ceil_div(a.B, 32) * 2;                                   // tile rows of the packed exchange buffers (pk_floats)
  dim3 grid(a.nsl / ns, nbt, 2), block(NW * 64);
  if (a.nbt16 > 0xffff || a.T > 0xffff || a.B > 0xffff || a.H > 0xffff)
    return ds2_set_error("rnn: T, B, H and the tile count must fit 16 bits (T=%d B=%d H=%d)", a.T, a.B, a.H);
  const int packed = a.nbt16 | ((a.dbg | ((bwd && a.gates_bf) ? 4 : 0)) << 16);   // preloaded dwords: tile count + flags, T | B, s | H
  float* gx_or_rec = (bwd && a.gates_bf) ? reinterpret_cast<float*>(a.gates_bf) : a.gx;
  const int T_B = a.T | (a.B << 16);
  const float* pk = a.pk;
  const float* wp = a.wp;
  const float* prev = G == 3 ? a.hbuf : a.aux;                      // previous hidden (GRU) / cell (LSTM) state
  for (int s = 0; s < a.T; ++s) {
    const int s_H = s | (a.H << 16);
    if (!bwd) {
      if (ns == 2) hipLaunchKernelGGL((rnn_fwd_step_kernel<G, 1, 2, BF>), grid, block, 0, st, pk, wp, a.gx, prev, a.bhh, s_H, T_B, packed, a);
      else if (mb == 2) hipLaunchKernelGGL((rnn_fwd_step_kernel<G, 2, 1, BF>), grid, block, 0, st, pk, wp, a.gx, prev, a.bhh, s_H, T_B, packed, a);
      else hipLaunchKernelGGL((rnn_fwd_step_kernel<G, 1, 1, BF>), grid, block, 0, st, pk, wp, a.gx, prev, a.bhh, s_H, T_B, packed, a);
    } else {
      if (ns == 2) hipLaunchKernelGGL((rnn_bwd_step_kernel<G, 1, 2, BF, SP>), grid, block, 0, st, pk, wp, gx_or_rec, a.aux, a.dy, s_H, T_B, packed, a.lddy, a);
      else if (mb == 2) hipLaunchKernelGGL((rnn_bwd_step_kernel<G, 2, 1, BF, SP>), grid, block, 0, st, pk, wp, gx_or_rec, a.aux, a.dy, s_H, T_B, packed, a.lddy, a);
      else hipLaunchKernelGGL((rnn_bwd_step_kernel<G, 1, 1, BF, SP>), grid, block, 0, st, pk, wp, gx_or_rec, a.aux, a.dy, s_H, T_B, packed, a.lddy, a);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn step launch failed: %s", hipGetErrorString(e));
  return 0;
}

size_t pk_floats(int B, int H, int kdim, int bf16) {   // size in 4-byte units (1 chunk = 1 KiB = 256 units in both precisions)
  const int nbt16 = ceil_div(B, 32) * 2;
  return (size_t)2 * 2 * nbt16 * ceil_div(kdim, bf16 ? 32 : 16) * 256;
}

template <bool BF>
int dispatch(int gates, bool bwd, const RnnArgs& a, hipStream_t st) {
  return gates == 3 ? launch_steps<3, BF>(bwd, a, st) : launch_steps<4, BF>(bwd, a, st);
}

}  // namespace

// packed-weight sizes in BYTES: which = 0 forward operand, 1 backward operand; bf16 = 0 | 1
// bf16 = 2 (fp32 mode with the split recurrences): each operand = [fp32 fragments | bf16 hi fragments | bf16 lo fragments]
extern "C" size_t ds2_rnn_packed_bytes(int gates, int H, int which, int bf16) {
  const size_t nsl = (size_t)ceil_div(H, 16);
  // bf16 = 2 (fp32 mode, split recurrences): [fp32 | hi | lo] and, forward operand of shapes the 10-unit-slice kernel exists for, [u10 hi | u10 lo]
  if (bf16 == 2)
    return ds2_rnn_packed_bytes(gates, H, which, 0) + 2 * ds2_rnn_packed_bytes(gates, H, which, 1) +
           ((which == 0 && u10_shape_ok(H)) ? 2 * u10_plane_bytes(gates, H) : 0);
  const int kc = bf16 ? 32 : 16;
  return (which == 0 ? 2 * nsl * gates * ceil_div(H, kc) : 2 * nsl * ceil_div(gates * H, kc)) * 1024;
}

// Re-pack W_hh = [weight_hh_l0 ; weight_hh_l0_reverse] (2, G*H, H) fp32 into MFMA-fragment order, as fp32 (bf16 = 0) or
// bf16 (bf16 = 1) fragments (call after every optimizer step / load_state_dict).
extern "C" int ds2_rnn_pack_whh(int gates, const float* whh, void* wp_fwd, void* wp_bwd, int H, int bf16, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_pack_whh: gates must be 3 or 4");
  DS2_REQUIRE(whh && wp_fwd && wp_bwd && H > 0 && (H % 4) == 0, "ds2_rnn_pack_whh: bad args");
  if (bf16 == 1) hipLaunchKernelGGL(rnn_pack_kernel<true>, dim3(2048), dim3(256), 0, (hipStream_t)stream, whh, wp_fwd, wp_bwd, gates, H);
  else hipLaunchKernelGGL(rnn_pack_kernel<false>, dim3(2048), dim3(256), 0, (hipStream_t)stream, whh, wp_fwd, wp_bwd, gates, H);
  DS2_LAUNCH_CHECK("rnn_pack_kernel");
  if (bf16 == 2) {                                               // + the split forward operand behind the fp32 one
    char* hi = (char*)wp_fwd + ds2_rnn_packed_bytes(gates, H, 0, 0);
    hipLaunchKernelGGL(rnn_pack_split_fwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, whh, (void*)hi,
                       (void*)(hi + ds2_rnn_packed_bytes(gates, H, 0, 1)), gates, H);
    DS2_LAUNCH_CHECK("rnn_pack_split_fwd_kernel");
    if (u10_shape_ok(H)) {                                       // + the 10-unit-slice operand (rnn_fwd_u10.h) behind those
      char* uh = hi + 2 * ds2_rnn_packed_bytes(gates, H, 0, 1);
      hipLaunchKernelGGL(rnn_pack_u10_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, whh, (void*)uh, (void*)(uh + u10_plane_bytes(gates, H)), gates, H);
      DS2_LAUNCH_CHECK("rnn_pack_u10_kernel");
    }
    char* bh = (char*)wp_bwd + ds2_rnn_packed_bytes(gates, H, 1, 0);
    hipLaunchKernelGGL(rnn_pack_split_bwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, whh, (void*)bh,
                       (void*)(bh + ds2_rnn_packed_bytes(gates, H, 1, 1)), gates, H);
    DS2_LAUNCH_CHECK("rnn_pack_split_bwd_kernel");
  }
  return 0;
}

// status of the persistent forward kernel since the last call: 8 ints {starved, block x, y, z, step, wave, ok-mask lo, hi}; out8[0] != 0
// if a wave ever gave up polling for its operand (results of that launch are then invalid).  Synchronises the device, clears the record.
extern "C" int ds2_rnn_persistent_status(ds2_rnn_ctx* ctx, int* out8) {
  DS2_REQUIRE(ctx && ctx->status_dev && out8, "ds2_rnn_persistent_status: null pointer");
  DS2_HIP(hipDeviceSynchronize());
  DS2_HIP(hipMemcpy(out8, ctx->status_dev, 8 * sizeof(int), hipMemcpyDeviceToHost));
  if (out8[0]) {
    // a launch starved (not every workgroup could be resident, or something else held CUs): the step that starved is invalid and must be
    // reported as failed; the next calls through this context take the one-launch-per-step kernels so that the caller's retry / next steps
    // work, then the persistent kernels are armed again (ctx->rearm_calls)
    ctx->cooldown = ctx->rearm_calls > 0 ? ctx->rearm_calls : -1;
    ++ctx->starved_total;
    DS2_HIP(hipMemset(ctx->status_dev, 0, 8 * sizeof(int)));
  }
  if (ctx->poison_host) *(volatile int*)ctx->poison_host = 0;      // (the device is idle: no poison kernel is in flight)
  return 0;
}

// bf16: 0 fp32, 1 bf16 operands, 2 fp32 mode with the split (hi + lo bf16) persistent forward kernel and the fp32 kernels as fallback
extern "C" size_t ds2_rnn_fwd_workspace_bytes(int B, int H, int bf16) {
  const size_t step = pk_floats(B, H, H, bf16 == 1) * sizeof(float);            // two ping-pong buffers of the step kernels
  size_t pers = 4 * fwd_xbuf_bytes(B, H, bf16 == 1) + 64;                        // four round-robin buffers of the persistent kernel + its census words
  if (bf16 == 2) { const size_t sp = 8 * fwd_xbuf_bytes(B, H, 1) + 64; pers = sp > pers ? sp : pers; }
  return step > pers ? step : pers;
}

// gates: 3 = GRU (r,z,n), 4 = LSTM (i,f,g,o).  bf16 = 1: the h W_hh^T product uses bf16 MFMA operands (fp32 accumulate;
// h, c, gates stay fp32) and wp_fwd must have been packed with bf16 = 1.
//   gx     (T,B,2,G*H)  in: X W_ih^T + b_ih for [fwd | reverse] ; out: activated gates (saved for backward)
//   wp_fwd packed W_hh (ds2_rnn_pack_whh), bhh (2,G*H) = [bias_hh_l0, bias_hh_l0_reverse]
//   hbuf   (T,B,2,H) out: h per direction (0 beyond each sample's length)
//   aux    (T,B,2,H) out: GRU W_hn h + b_hn ; LSTM cell state
//   h_bf16 optional (T,B,2,H) bf16: a bf16 copy of hbuf, written by a PERSISTENT launch only (ds2_rnn_last_path() & 1 after the call)
extern "C" int ds2_rnn_fwd_ex(ds2_rnn_ctx* ctx, int gates, float* gx, const void* wp_fwd, const float* bhh, float* hbuf, float* aux, const int* lens_dev, int T,
                              int B, int H, int bf16, void* gates_bf16, void* h_bf16, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_fwd: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE(gx && wp_fwd && bhh && hbuf && aux && lens_dev, "ds2_rnn_fwd: null pointer");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_fwd: need H %% 4 == 0 (H=%d)", H);
  DS2_REQUIRE(ws && ws_bytes >= ds2_rnn_fwd_workspace_bytes(B, H, bf16), "ds2_rnn_fwd: workspace too small");
  RnnArgs a{};
  a.gx = gx; a.aux = aux; a.hbuf = hbuf; a.wp = (const float*)wp_fwd; a.bhh = bhh; a.pk = (float*)ws; a.lens = lens_dev;
  a.T = T; a.B = B; a.H = H; a.gates_bf = (__bf16*)gates_bf16; a.h_bf = (__bf16*)h_bf16;
  a.hctx = ctx; a.status = ctx ? ctx->status_dev : nullptr;
  a.prearmed = take_prearmed(ctx);
  int scratch_path = 0;
  int& last_path = ctx ? ctx->last_path : scratch_path;
  {
    a.dbg = ctx ? ctx->debug_flags : 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    if (bf16 == 2) {
      // fp32 mode, split forward recurrence: operands behind the fp32 fragments of wp_fwd (ds2_rnn_packed_bytes(.., 0, 2)); where its shape
      // does not qualify (or during a cooldown, which the fp32 attempt below counts) the fp32 kernels take the call with the fp32 fragments
      static const char* env = getenv("DS2_F32_RNN");              // "f32": never the split kernel (A/B runs)
      if (persist_idle(ctx, false) && !(env && env[0] == 'f') && !a.gates_bf) {
        RnnArgs b = a;
        b.wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wp_fwd) + ds2_rnn_packed_bytes(gates, H, 0, 0));
        b.dbg &= ~256;
        last_path &= ~256;
        if (!(a.dbg & 256))                                         // (debug flag 256: prefer the 10-unit kernel where its shape qualifies)
          rc = gates == 3 ? try_launch_persistent_fwd<3, true, true>(b, st) : try_launch_persistent_fwd<4, true, true>(b, st);
        if (rc == 0 && u10_shape_ok(H)) {
          // the 16-unit split kernel does not fit (LSTM H = 1280 at B = 32: BASELINE C4): ten-unit slices, rnn_fwd_u10.h
          b.wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wp_fwd) + ds2_rnn_packed_bytes(gates, H, 0, 0) +
                                                2 * ds2_rnn_packed_bytes(gates, H, 0, 1));
          rc = gates == 3 ? try_launch_fwd_u10<3>(b, st) : try_launch_fwd_u10<4>(b, st);
          if (rc == 1) last_path |= 256;                          // bit 8: the 10-unit kernel
        }
        if (rc == 1) last_path |= 32;                             // bit 5: a split kernel took the call
      }
      if (rc == 0) { last_path &= ~32; a.dbg &= ~256; rc = gates == 3 ? try_launch_persistent_fwd<3, false>(a, st) : try_launch_persistent_fwd<4, false>(a, st); }
    } else {
      last_path &= ~(32 | 256);
      rc = bf16 ? (gates == 3 ? try_launch_persistent_fwd<3, true>(a, st) : try_launch_persistent_fwd<4, true>(a, st))
                : (gates == 3 ? try_launch_persistent_fwd<3, false>(a, st) : try_launch_persistent_fwd<4, false>(a, st));
    }
    last_path = (last_path & ~1) | (rc == 1 ? 1 : 0);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  // step kernels: zero padding rows / columns of the ping-pong buffers (the persistent path has filled its own with the sentinel)
  DS2_HIP(hipMemsetAsync(ws, 0, ds2_rnn_fwd_workspace_bytes(B, H, bf16), (hipStream_t)stream));
  return bf16 == 1 ? dispatch<true>(gates, false, a, (hipStream_t)stream) : dispatch<false>(gates, false, a, (hipStream_t)stream);
}

// ds2_rnn_fwd_ex for the bf16 TRAINING mode (bf16 == 1, packed gate records), with two more optional operands:
//   gx_bf16  the x-projections as the bf16 tensor ds2_gemm_bf16_nt_obf16 wrote (then gx may be NULL).  Persistent kernels only: returns 1 —
//            nothing launched, nothing counted — when the call cannot run as a persistent launch (cooldown, forward kernel switched off, no
//            persistent kernel for the shape); the caller widens (ds2_cast_f32_from_bf16) and calls again with gx.
//   hsum     (2, ceil(B/16), H) fp32: per direction and 16-row batch tile the sums of h over time — written by a persistent launch only
//            (ds2_rnn_last_path() & 1), the input of ds2_center_colstats.
// Returns 0 = done, 1 = see gx_bf16, < 0 error.
extern "C" int ds2_rnn_fwd_x(ds2_rnn_ctx* ctx, int gates, float* gx, const void* gx_bf16, const void* wp_fwd, const float* bhh, float* hbuf, float* aux,
                             const int* lens_dev, int T, int B, int H, void* gates_bf16, void* h_bf16, float* hsum, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_fwd_x: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE((gx || gx_bf16) && wp_fwd && bhh && hbuf && aux && lens_dev && gates_bf16, "ds2_rnn_fwd_x: null pointer");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_fwd_x: need H %% 4 == 0 (H=%d)", H);
  DS2_REQUIRE(ws && ws_bytes >= ds2_rnn_fwd_workspace_bytes(B, H, 1), "ds2_rnn_fwd_x: workspace too small");
  if (!gx && !persist_idle(ctx, false)) return 1;
  RnnArgs a{};
  a.gx = gx_bf16 ? nullptr : gx; a.gxb = (const __bf16*)gx_bf16; a.aux = aux; a.hbuf = hbuf; a.wp = (const float*)wp_fwd; a.bhh = bhh; a.pk = (float*)ws;
  a.lens = lens_dev; a.T = T; a.B = B; a.H = H; a.gates_bf = (__bf16*)gates_bf16; a.h_bf = (__bf16*)h_bf16; a.hsum = hsum;
  a.hctx = ctx; a.status = ctx ? ctx->status_dev : nullptr;
  a.prearmed = take_prearmed(ctx);
  a.dbg = ctx ? ctx->debug_flags : 0;
  int scratch_path = 0;
  int& last_path = ctx ? ctx->last_path : scratch_path;
  const int rc = gates == 3 ? try_launch_persistent_fwd<3, true>(a, (hipStream_t)stream) : try_launch_persistent_fwd<4, true>(a, (hipStream_t)stream);
  if (rc < 0) return rc;
  last_path = (last_path & ~(1 | 32 | 256)) | (rc == 1 ? 1 : 0);
  if (rc == 1) return 0;
  if (!gx) return 1;
  a.gx = gx; a.gxb = nullptr; a.hsum = nullptr;
  DS2_HIP(hipMemsetAsync(ws, 0, ds2_rnn_fwd_workspace_bytes(B, H, 1), (hipStream_t)stream));
  return dispatch<true>(gates, false, a, (hipStream_t)stream);
}

extern "C" int ds2_rnn_fwd(ds2_rnn_ctx* ctx, int gates, float* gx, const void* wp_fwd, const float* bhh, float* hbuf, float* aux, const int* lens_dev, int T,
                           int B, int H, int bf16, void* gates_bf16, void* ws, size_t ws_bytes, void* stream) {
  return ds2_rnn_fwd_ex(ctx, gates, gx, wp_fwd, bhh, hbuf, aux, lens_dev, T, B, H, bf16, gates_bf16, nullptr, ws, ws_bytes, stream);
}

// bias gradients from the per-batch-row sums of the persistent backward kernel: part (B,2,4,H) -> db_ih (2,G*H), db_hh (2,G*H).
// GRU slots [d r, d z, d n, d(hn)]: db_ih = [r, z, n], db_hh = [r, z, hn].  LSTM: both = [i, f, g, o].  Rows summed in index order.
__global__ void rnn_bias_finalize_kernel(const float* __restrict__ part, int B, int H, int G, float* __restrict__ dbih, float* __restrict__ dbhh) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (dir, slot, j)
  if (idx >= 2 * 4 * H) return;
  const int j = idx % H, slot = (idx / H) & 3, dir = idx / (4 * H);
  // eight independent loads in flight, added in row order (one dependent load per row made this 64 x 0.45 us = 29 us per call)
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(long long)(b + k) * 8 * H + idx];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; b < B; ++b) s += part[(long long)b * 8 * H + idx];
  const long long o = (long long)dir * G * H + j;
  if (G == 4) { dbih[o + slot * H] = s; dbhh[o + slot * H] = s; }
  else if (slot < 2) { dbih[o + slot * H] = s; dbhh[o + slot * H] = s; }
  else if (slot == 2) dbih[o + 2 * H] = s;
  else dbhh[o + 2 * H] = s;
}

extern "C" int ds2_rnn_bias_grads(int gates, const float* bias_part, int B, int H, float* dbih, float* dbhh, void* stream) {
  DS2_REQUIRE((gates == 3 || gates == 4) && bias_part && dbih && dbhh && B > 0 && H > 0, "ds2_rnn_bias_grads: bad args");
  hipLaunchKernelGGL(rnn_bias_finalize_kernel, dim3(ceil_div(8 * H, 256)), dim3(256), 0, (hipStream_t)stream, bias_part, B, H, gates, dbih, dbhh);
  DS2_LAUNCH_CHECK("rnn_bias_finalize_kernel");
  return 0;
}

extern "C" size_t ds2_rnn_bwd_workspace_bytes(int gates, int B, int H, int bf16) {
  const size_t step = pk_floats(B, H, gates * H, bf16 == 1) * sizeof(float);     // two ping-pong buffers of the step kernels
  size_t pers = 4 * bwd_xbuf_bytes(gates, B, H, bf16 == 1) + 64;                  // four round-robin buffers of the persistent kernel + its census words
  if (bf16 == 2) pers = std::max(pers, 8 * bwd_xbuf_bytes(gates, B, H, 1) + 64);  // the split form: hi and lo plane per buffer
  if (bf16 == 2 && ksplit_shape_ok(H)) pers = std::max(pers, 2 * ksplit_xbuf_bytes(B, H) + 64);   // ... of the K-split kernel: two planes of two slots
  if (bf16 == 1 && ksplit_shape_ok(H)) pers = std::max(pers, ksplit_xbuf_bytes(B, H) + 64);   // two slots of the K-split kernel + census
  return (size_t)4 * B * H * sizeof(float) + (step > pers ? step : pers);
}

// Which recurrences may run as ONE persistent launch (bf16 mode; default: both).  A persistent launch needs all of its workgroups resident
// at once, so a caller that runs other kernels concurrently on the device during backward (collectives on a communication stream)
// must switch the backward one off.  DS2_RNN_PERSISTENT=0 in the environment switches both off.
// which kernel family the last recurrence calls used: bit 0 = ds2_rnn_fwd, bit 1 = ds2_rnn_bwd took the persistent kernel (reporting only)
extern "C" int ds2_rnn_last_path(const ds2_rnn_ctx* ctx) { return ctx ? ctx->last_path : 0; }

namespace {
__global__ void step_gate_kernel(const float* __restrict__ loss, int* __restrict__ flag, const int* __restrict__ status) {
  const float l = *loss;
  // 1 = apply the update; 0 = this rank's loss is not valid; -1 = a persistent recurrence launch of this rank starved.  The MIN over the
  // ranks then tells EVERY rank which of the two happened somewhere (the trainer restores BatchNorm statistics on all ranks after a -1).
  *flag = (status && status[0] != 0) ? -1 : ((l == l && l != __builtin_inff() && l != -__builtin_inff() && l >= 0.f) ? 1 : 0);
}
}  // namespace

// flag[0] = 1 if the train step enqueued so far on `stream` is valid when this kernel RUNS: the loss (device scalar) is finite and
// non-negative (functional.py:45-61) and no persistent recurrence launch has recorded starvation; 0 for an invalid loss, -1 for starvation.  Consumed on the device by
// ds2_adamw_gated_f32 (and, under data parallelism, all-reduced with MIN first), read back by the host whenever convenient.
extern "C" int ds2_rnn_step_gate(const ds2_rnn_ctx* ctx, const float* loss_dev, int* flag_dev, void* stream) {
  DS2_REQUIRE(loss_dev && flag_dev, "ds2_rnn_step_gate: null pointer");
  hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_dev, flag_dev, ctx ? (const int*)ctx->status_dev : (const int*)nullptr);
  DS2_LAUNCH_CHECK("step_gate_kernel");
  return 0;
}

namespace {
__global__ void poison_if_starved_kernel(float* __restrict__ buf, long long n, const int* __restrict__ status, int* __restrict__ seen) {
  if (status[0] == 0) return;
  if (seen && blockIdx.x == 0 && threadIdx.x == 0) { *seen = 1; __threadfence_system(); }
  const float qnan = __builtin_nanf("");
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) buf[i] = qnan;
}
}  // namespace

// Inference without a host synchronisation: if a persistent recurrence launch enqueued before this call on `stream` has recorded
// starvation (its activations are invalid), overwrite buf[0..n) with NaN — in stream order, without reading or clearing the record.  The
// caller's next ds2_rnn_persistent_status (at a point where it synchronises anyway) raises; until then nothing plausible-looking leaves.
extern "C" int ds2_rnn_poison_if_starved(ds2_rnn_ctx* ctx, float* buf, size_t n, void* stream) {
  DS2_REQUIRE(ctx && ctx->status_dev && (buf || n == 0), "ds2_rnn_poison_if_starved: null pointer");
  if (n == 0) return 0;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(poison_if_starved_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, (long long)n, (const int*)ctx->status_dev, ctx->poison_dev);
  DS2_LAUNCH_CHECK("poison_if_starved_kernel");
  return 0;
}

// 1 if a ds2_rnn_poison_if_starved kernel has overwritten a buffer since the last ds2_rnn_persistent_status (a host memory read: no
// synchronisation, no device call); the caller should then call ds2_rnn_persistent_status, which reports, clears and starts the cooldown.
extern "C" int ds2_rnn_poison_seen(const ds2_rnn_ctx* ctx) { return (ctx && ctx->poison_host) ? *(volatile int*)ctx->poison_host : 0; }

// What a workgroup of the K-split persistent backward recurrence occupies, from the binary that is loaded: out3 = {registers per lane
// (hipFuncGetAttributes), static LDS bytes, threads}.  Returns 1 if this (gates, H) shape has a K-split instance, 0 if not.  The host
// uses it to decide whether the co-resident weight-gradient kernel (ds2_gemm_bf16_tn_group: 4 waves x 128 registers) fits beside it.
extern "C" int ds2_rnn_bwd_ksplit_footprint(int gates, int H, int* out3) {
  DS2_REQUIRE(out3 && (gates == 3 || gates == 4), "ds2_rnn_bwd_ksplit_footprint: bad args");
  out3[0] = out3[1] = out3[2] = 0;
  if (!ksplit_shape_ok(H)) return 0;
  const int nt = H / 128;
  if (nt * gates * 4 > 176) return 0;
  const void* fn = nullptr;
#define DS2_KSF(G_, NT_)                                                           \
  if (gates == G_ && nt == NT_) {                                                  \
    if constexpr (NT_ * G_ * 4 <= 176) fn = G_ == 3 ? (const void*)rnn_bwd_ksplit_kernel<3, NT_, true> : (const void*)rnn_bwd_ksplit_kernel<G_, NT_, false>; \
  }
  DS2_KSF(3, 2) DS2_KSF(3, 4) DS2_KSF(3, 6) DS2_KSF(3, 8) DS2_KSF(3, 10)
  DS2_KSF(4, 2) DS2_KSF(4, 4) DS2_KSF(4, 6) DS2_KSF(4, 8) DS2_KSF(4, 10)
#undef DS2_KSF
  if (!fn) return 0;
  hipFuncAttributes at;
  DS2_HIP(hipFuncGetAttributes(&at, fn));
  out3[0] = at.numRegs; out3[1] = (int)at.sharedSizeBytes; out3[2] = NW * 64;
  return 1;
}

extern "C" int ds2_rnn_persistent_enable(ds2_rnn_ctx* ctx, int forward, int backward) {
  DS2_REQUIRE(ctx, "ds2_rnn_persistent_enable: null context");
  ctx->persist_fwd = forward != 0;
  ctx->persist_bwd = backward != 0;
  return 0;
}

// {launches through this context that starved, recurrence calls left on the step kernels before the persistent ones are armed again (-1: never)}
extern "C" int ds2_rnn_persistent_counters(const ds2_rnn_ctx* ctx, int* out2) {
  DS2_REQUIRE(ctx && out2, "ds2_rnn_persistent_counters: null pointer");
  out2[0] = ctx->starved_total;
  out2[1] = ctx->cooldown;
  return 0;
}

//   dy     (T,B,H) pitch lddy: grad wrt y = h_fwd + h_bwd
//   gx     in: gates from fwd ; out: grad wrt the x-projections (T,B,2,G*H)  (= dGx, feeds dW_ih, db_ih, dX)
//   dgx_bf16  optional (T,B,2,G*H) bf16: when given, dGx is written THERE (rounded to bf16) and gx keeps the gates
//   gates_bf16 optional (T,B,2,H,4) bf16 packed gate records written by ds2_rnn_fwd(gates_bf16): read instead of gx (+ GRU aux)
//   aux    GRU: in hn, out d(hn) [so that dGh = (dGx_r, dGx_z, aux)] ; LSTM: cell state (unchanged; dGh = dGx)
//   wp_bwd packed W_hh^T (ds2_rnn_pack_whh)
//   dhn_bf16  optional (T,B,2,H) bf16 (GRU): a bf16 copy of the d(hn) written into aux          } written by a PERSISTENT launch only
//   bias_part optional (B,2,4,H) fp32: per-batch-row sums over time of [d r, d z, d n, d(hn)] (GRU)  } (ds2_rnn_last_path() & 2 after the
//             / [d i, d f, d g, d o] (LSTM); their column sums over B are db_ih / db_hh             } call); untouched otherwise
extern "C" int ds2_rnn_bwd_ex(ds2_rnn_ctx* ctx, int gates, const float* dy, int lddy, float* gx, float* aux, const float* hbuf, const void* wp_bwd,
                              const int* lens_dev, int T, int B, int H, int bf16, void* dgx_bf16, const void* gates_bf16, void* dhn_bf16,
                              float* bias_part, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_bwd: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE(dy && aux && hbuf && wp_bwd && lens_dev, "ds2_rnn_bwd: null pointer");
  DS2_REQUIRE(gx || (gates_bf16 && dgx_bf16), "ds2_rnn_bwd: gx may only be NULL with both gates_bf16 and dgx_bf16 given");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_bwd: need H %% 4 == 0 (H=%d)", H);
  DS2_REQUIRE(ws && ws_bytes >= ds2_rnn_bwd_workspace_bytes(gates, B, H, bf16), "ds2_rnn_bwd: workspace too small");
  RnnArgs a{};
  a.gx = gx; a.aux = aux; a.hbuf = const_cast<float*>(hbuf); a.wp = (const float*)wp_bwd; a.dy = dy; a.lddy = lddy;
  a.dcar = (float*)ws; a.pk = (float*)ws + (size_t)4 * B * H;
  a.dgx_bf = (__bf16*)dgx_bf16;
  a.gates_bf = (__bf16*)const_cast<void*>(gates_bf16);
  a.dhn_bf = (__bf16*)dhn_bf16; a.bsum = bias_part;
  a.lens = lens_dev; a.T = T; a.B = B; a.H = H;
  a.hctx = ctx; a.status = ctx ? ctx->status_dev : nullptr;
  a.prearmed = take_prearmed(ctx);
  int scratch_path = 0, scratch_kind = 0;
  int& last_path = ctx ? ctx->last_path : scratch_path;
  int& last_bwd_kind = ctx ? ctx->last_bwd_kind : scratch_kind;
  {
    a.dbg = ctx ? ctx->debug_flags : 0;
    hipStream_t st = (hipStream_t)stream;
    // bf16: the K-split kernel where the shape qualifies (2 = it does, but a starved launch's cooldown is running: step kernels)
    int rc = bf16 == 1 ? (gates == 3 ? try_launch_ksplit_bwd<3>(a, st) : try_launch_ksplit_bwd<4>(a, st)) : 0;
    last_bwd_kind = rc == 1 ? 2 : 0;
    bool split = false;
    if (rc == 0 && bf16 == 2) {
      // fp32 mode, split backward recurrence (operands behind the fp32 fragments of wp_bwd); else / during a cooldown the fp32 kernels
      static const char* env = getenv("DS2_F32_RNN");
      if (persist_idle(ctx, true) && !(env && env[0] == 'f') && !a.gates_bf && !a.dgx_bf) {
        RnnArgs b = a;
        b.wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wp_bwd) + ds2_rnn_packed_bytes(gates, H, 1, 0));
        static const char* envk = ds2_exp_getenv("DS2_RNN_KSPLIT");
        const int rk = (envk && envk[0] == '0') ? 0 : (gates == 3 ? try_launch_ksplit_bwd<3, true>(b, st) : try_launch_ksplit_bwd<4, true>(b, st));
        if (rk < 0) return rk;
        if (rk == 1) { rc = 1; split = true; last_bwd_kind = 2; }
        else {
          rc = gates == 3 ? try_launch_persistent_bwd<3, true, true>(b, st) : try_launch_persistent_bwd<4, true, true>(b, st);
          split = rc == 1;
          if (split) last_bwd_kind = 1;
        }
      }
    }
    if (rc == 0) {
      rc = bf16 == 1 ? (gates == 3 ? try_launch_persistent_bwd<3, true>(a, st) : try_launch_persistent_bwd<4, true>(a, st))
                     : (gates == 3 ? try_launch_persistent_bwd<3, false>(a, st) : try_launch_persistent_bwd<4, false>(a, st));
      last_bwd_kind = rc == 1 ? 1 : 0;
    }
    last_path = (last_path & ~(6 | 16 | 64)) | (rc == 1 ? 2 : 0) | (last_bwd_kind == 2 ? 4 : 0) | (split ? 64 : 0);   // bit 6: the split kernel
    if (rc != 0 && rc != 2) return rc < 0 ? rc : 0;
  }
  DS2_HIP(hipMemsetAsync(ws, 0, ds2_rnn_bwd_workspace_bytes(gates, B, H, bf16), (hipStream_t)stream));   // step kernels: zero carry + padding
  if (bf16 == 2 && (H % 32) == 0 && !a.gates_bf && !a.dgx_bf) {
    // fp32 mode, no persistent kernel took the call (shape, cooldown): the step kernels in split form — unless DS2_F32_RNN=f32 / debug selectors
    static const char* env = getenv("DS2_F32_RNN");
    if (!(env && env[0] == 'f') && !(a.dbg & ~128)) {
      a.wp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wp_bwd) + ds2_rnn_packed_bytes(gates, H, 1, 0));
      last_path |= 64;                                            // bit 6 without bit 1: split step kernels
      return gates == 3 ? launch_steps<3, true, true>(true, a, (hipStream_t)stream) : launch_steps<4, true, true>(true, a, (hipStream_t)stream);
    }
  }
  return bf16 == 1 ? dispatch<true>(gates, true, a, (hipStream_t)stream) : dispatch<false>(gates, true, a, (hipStream_t)stream);
}

// ds2_rnn_bwd_ex for a layer whose output feeds a BatchNorm1d (every recurrent layer of the model: SequenceWise(BatchNorm1d) of the next
// layer, blocks.py:75,85-86, or of the fc block, deepspeech.py:104).  dyn = gradient wrt that BatchNorm's OUTPUT (pitch lddyn), bn_x = its
// input = this layer's y (pitch ldx), bn_mean / bn_var / bn_gamma its batch statistics and weight, bn_s0 / bn_s1 the column sums of dyn
// and of dyn * xhat (= dbeta / dgamma, from ds2_bn1d_bwd_f32 with dX = NULL).  Where the K-split persistent kernel takes the call, the
// elementwise half of the BatchNorm backward is applied on the fly to the one value a (row, unit) pair needs per step (bit 16 of
// ds2_rnn_last_path()) — no pass over (T*B, H); otherwise it is materialised into dy_scratch (T*B, H) first and the call proceeds as
// ds2_rnn_bwd_ex(dy = dy_scratch).
static int rnn_bwd_bn_impl(ds2_rnn_ctx* ctx, int gates, const float* dyn, int lddyn, const float* bn_x, int bn_x_bf16, int ldx, const float* bn_mean, const float* bn_var,
                              const float* bn_gamma, const float* bn_s0, const float* bn_s1, float bn_eps, float* dy_scratch, float* gx,
                              float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev, int T, int B, int H, int bf16,
                              void* dgx_bf16, const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws, size_t ws_bytes,
                              void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_bwd_bn: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE(dyn && bn_x && bn_mean && bn_var && bn_gamma && bn_s0 && bn_s1 && aux && hbuf && wp_bwd && lens_dev,
              "ds2_rnn_bwd_bn: null pointer");
  DS2_REQUIRE(gx || (gates_bf16 && dgx_bf16), "ds2_rnn_bwd_bn: gx may only be NULL with both gates_bf16 and dgx_bf16 given");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_bwd_bn: need H %% 4 == 0 (H=%d)", H);
  DS2_REQUIRE(ws && ws_bytes >= ds2_rnn_bwd_workspace_bytes(gates, B, H, bf16), "ds2_rnn_bwd_bn: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (bf16 == 1) {
    RnnArgs a{};
    a.gx = gx; a.aux = aux; a.hbuf = const_cast<float*>(hbuf); a.wp = (const float*)wp_bwd; a.dy = dyn; a.lddy = lddyn;
    a.dcar = (float*)ws; a.pk = (float*)ws + (size_t)4 * B * H;
    a.dgx_bf = (__bf16*)dgx_bf16; a.gates_bf = (__bf16*)const_cast<void*>(gates_bf16);
    a.dhn_bf = (__bf16*)dhn_bf16; a.bsum = bias_part;
    a.lens = lens_dev; a.T = T; a.B = B; a.H = H;
    a.bn_x = bn_x; a.bn_x_bf16 = bn_x_bf16; a.ldbnx = ldx; a.bn_mean = bn_mean; a.bn_var = bn_var; a.bn_gamma = bn_gamma; a.bn_s0 = bn_s0; a.bn_s1 = bn_s1; a.bn_eps = bn_eps;
    a.hctx = ctx; a.status = ctx ? ctx->status_dev : nullptr;
    a.prearmed = take_prearmed(ctx);
    a.dbg = ctx ? ctx->debug_flags : 0;
    // (only when the persistent backward is armed: a cool-down call is counted once, by the un-fused call below)
    const int rc = persist_idle(ctx, true) ? (gates == 3 ? try_launch_ksplit_bwd<3>(a, st) : try_launch_ksplit_bwd<4>(a, st)) : 0;
    if (rc < 0) return rc;
    if (rc == 1) {
      ctx->last_bwd_kind = 2;
      ctx->last_path = (ctx->last_path & ~(2 | 4 | 16)) | 2 | 4 | 16;
      return 0;
    }
  }
  // not taken by the K-split kernel: materialise dy, then the ordinary path.  dy_scratch == NULL: nothing has been launched or counted
  // yet — return 1 so that the caller allocates the (T*B, H) buffer only when it is really needed, and calls again
  if (!dy_scratch) return 1;
  int rc = bn_x_bf16 ? ds2i_bn1d_bwd_apply_xbf(dyn, lddyn, bn_x, ldx, dy_scratch, H, T * B, H, bn_mean, bn_var, bn_gamma, bn_s0, bn_s1, bn_eps, st)
                     : ds2i_bn1d_bwd_apply(dyn, lddyn, bn_x, ldx, dy_scratch, H, T * B, H, bn_mean, bn_var, bn_gamma, bn_s0, bn_s1, bn_eps, st);
  if (rc) return rc;
  rc = ds2_rnn_bwd_ex(ctx, gates, dy_scratch, H, gx, aux, hbuf, wp_bwd, lens_dev, T, B, H, bf16, dgx_bf16, gates_bf16, dhn_bf16, bias_part, ws, ws_bytes,
                      stream);
  if (ctx) ctx->last_path &= ~16;
  return rc;
}

extern "C" int ds2_rnn_bwd_bn(ds2_rnn_ctx* ctx, int gates, const float* dyn, int lddyn, const float* bn_x, int ldx, const float* bn_mean, const float* bn_var,
                              const float* bn_gamma, const float* bn_s0, const float* bn_s1, float bn_eps, float* dy_scratch, float* gx,
                              float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev, int T, int B, int H, int bf16,
                              void* dgx_bf16, const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws, size_t ws_bytes,
                              void* stream) {
  return rnn_bwd_bn_impl(ctx, gates, dyn, lddyn, bn_x, 0, ldx, bn_mean, bn_var, bn_gamma, bn_s0, bn_s1, bn_eps, dy_scratch, gx, aux, hbuf, wp_bwd, lens_dev,
                         T, B, H, bf16, dgx_bf16, gates_bf16, dhn_bf16, bias_part, ws, ws_bytes, stream);
}

// ds2_rnn_bwd_bn with the BatchNorm's input given as bf16 (pitch ldx, even; H % 4 == 0): the centred operand of ds2_center_colstats, bn_mean its
// delta.  Everything else as ds2_rnn_bwd_bn.
extern "C" int ds2_rnn_bwd_bn_xbf16(ds2_rnn_ctx* ctx, int gates, const float* dyn, int lddyn, const void* bn_x_bf16, int ldx, const float* bn_mean,
                                    const float* bn_var, const float* bn_gamma, const float* bn_s0, const float* bn_s1, float bn_eps, float* dy_scratch,
                                    float* gx, float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev, int T, int B, int H, int bf16,
                                    void* dgx_bf16, const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE((ldx % 2) == 0 && (H % 4) == 0 && ((uintptr_t)bn_x_bf16 % 4) == 0, "ds2_rnn_bwd_bn_xbf16: even pitch, H %% 4 == 0, 4-byte aligned base");
  return rnn_bwd_bn_impl(ctx, gates, dyn, lddyn, (const float*)bn_x_bf16, 1, ldx, bn_mean, bn_var, bn_gamma, bn_s0, bn_s1, bn_eps, dy_scratch, gx, aux, hbuf,
                         wp_bwd, lens_dev, T, B, H, bf16, dgx_bf16, gates_bf16, dhn_bf16, bias_part, ws, ws_bytes, stream);
}

extern "C" int ds2_rnn_bwd(ds2_rnn_ctx* ctx, int gates, const float* dy, int lddy, float* gx, float* aux, const float* hbuf, const void* wp_bwd,
                           const int* lens_dev, int T, int B, int H, int bf16, void* dgx_bf16, const void* gates_bf16, void* ws, size_t ws_bytes,
                           void* stream) {
  return ds2_rnn_bwd_ex(ctx, gates, dy, lddy, gx, aux, hbuf, wp_bwd, lens_dev, T, B, H, bf16, dgx_bf16, gates_bf16, nullptr, nullptr, ws, ws_bytes,
                        stream);
}
