// Fused log-softmax + CTC loss + gradient (fp32, log space), blank = 0, reduction = per-utterance nll.
// Replaces `out.float().log_softmax(2)` + `torch.nn.CTCLoss(reduction="sum")` forward AND their
// autograd backward (deepspeech_trainer.py:108-112, trainers/__main__.py:53).
//
//   input  logits (T,B,C) row pitch ld      (need not be normalised: the row log-sum-exp is computed
//                                            here; feeding log-probs gives lse = 0 -> same result)
//   output nll[b] = -log p(target_b | logits[:T_b, b])           (+inf if no valid alignment)
//          grad[t,b,c] = scale * (softmax(logits)[t,b,c] - occupancy[t,b,c]) for t < T_b, else 0
//          (= d(sum_b nll_b)/d logits * scale; rows of an infeasible utterance get 0)
//
// Kernels: (1) one wavefront per (t,b) row: max/sum shuffles -> lse.  (2) one workgroup per
// (utterance, direction): the 2U+1 lattice row lives in LDS, one barrier per frame, emission
// gathers prefetched 4 frames ahead.  (3) occupancy + gradient, deterministic (no float atomics):
// repeated labels are summed by the thread owning the first occurrence walking a next-same chain.
#include "common.h"
#include <math.h>

namespace {

constexpr float NEG_INF = -INFINITY;

// log-sum-exp of the lattice recurrences on the hardware exp2 / log2 (v_exp_f32, v_log_f32: 1 ulp each): every argument is <= 0 and
// the sum lies in [1, 3], so the only extra error over libm's expf / logf is the fp32 scaling of the argument (|x| 2^-24 in the
// exponent: < 5e-6 relative, and only on terms that are themselves < e^-80 of the sum).  The lattice is ONE dependent lse3 per frame per
// lane: with libm's range-reduced expf / logf (~25 instructions each) a frame cost 0.56 us, with these 0.45 us (c3: 278 -> 227 us per step;
// the rest is the LDS round trip + barrier + the dependent exp2 / log2 chain).  A one-wave-per-lattice form (K states per lane, DPP wave
// shifts instead of LDS + barrier, bit-identical) was built and is SLOWER: K dependent chains per lane per frame.
__device__ __forceinline__ float fast_exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log_(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + fast_log_(fast_exp_(a - m) + fast_exp_(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == NEG_INF) return NEG_INF;
  return m + fast_log_(fast_exp_(a - m) + fast_exp_(b - m) + fast_exp_(c - m));
}

// one wave per row
__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ logits, int ld, int T, int Bn, int C,
                                                      const int* __restrict__ in_lens, float* __restrict__ lse) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T * Bn) return;
  const int t = row / Bn, b = row % Bn;
  if (t >= in_lens[b]) return;
  const int lane = threadIdx.x & 63;
  const float* p = logits + (long long)row * ld;
  float m = NEG_INF;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(p[c] - m);
  s = wave_sum(s);
  if (lane == 0) lse[row] = m + logf(s);
}

// blockIdx.x = utterance, blockIdx.y = 0: alpha (forward in t), 1: beta (backward in t).
// ab layout: [2][B][T][Smax]
__global__ __launch_bounds__(1024) void ctc_lattice_kernel(const float* __restrict__ logits, int ld, int T, int Bn, int C,
                                                           const int* __restrict__ targets, const int* __restrict__ tgt_off,
                                                           const int* __restrict__ in_lens, const int* __restrict__ tgt_lens,
                                                           const float* __restrict__ lse, float* __restrict__ ab, int Smax,
                                                           float* __restrict__ nll) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][S] rows
  const int b = blockIdx.x, dirn = blockIdx.y;
  const int Tb = min(in_lens[b], T), U = tgt_lens[b];
  const int S = 2 * U + 1;
  float* out = ab + (((long long)dirn * Bn + b) * T) * Smax;
  if (Tb <= 0) {
    if (dirn == 0 && threadIdx.x == 0) nll[b] = (U == 0) ? 0.f : INFINITY;
    return;
  }
  const int* lab = targets + tgt_off[b];
  float* row0 = smem;
  float* row1 = smem + Smax;
  const int nthr = blockDim.x;
  // this thread owns states s = tid, tid + nthr, ... (normally exactly one)
  // generic loop for very long targets; the common case S <= blockDim.x runs one iteration.
  const int step = dirn == 0 ? 1 : -1;
  const int tfirst = dirn == 0 ? 0 : Tb - 1;

  // init row
  for (int s = threadIdx.x; s < S; s += nthr) {
    const int cls = (s & 1) ? lab[s >> 1] : 0;
    const float lp = logits[((long long)tfirst * Bn + b) * ld + cls] - lse[tfirst * Bn + b];
    float v = NEG_INF;
    if (dirn == 0) { if (s == 0 || s == 1) v = lp; }
    else { if (s == S - 1 || s == S - 2) v = lp; }
    row0[s] = v;
    out[(long long)tfirst * Smax + s] = v;
  }
  __syncthreads();
  float* prev = row0;
  float* cur = row1;
  if (S <= nthr) {
    // Hot loop, written for instruction count (one dependent lattice row per frame: every instruction here is on the critical path).
    // Per-thread pointers advance by constant strides; emissions are loaded unconditionally 4 frames ahead from CLAMPED frame indices
    // (no exec-mask branches, no per-frame index arithmetic); only the stores are predicated.
    const int s = threadIdx.x;
    const bool act = s < S;
    int cls = 0;
    bool skip = false;
    if (act) {
      cls = (s & 1) ? lab[s >> 1] : 0;
      if (dirn == 0) skip = (s >= 2) && (s & 1) && (lab[s >> 1] != lab[(s >> 1) - 1]);
      else skip = (s + 2 < S) && (s & 1) && (lab[s >> 1] != lab[(s >> 1) + 1]);
    }
    const long long lg_stride = (long long)step * Bn * ld;
    const int ls_stride = step * Bn, out_stride = step * Smax;
    const float* lgp = logits + ((long long)tfirst * Bn + b) * ld + cls;        // frame tfirst, this state's class
    const float* lsp = lse + (long long)tfirst * Bn + b;
    float* op = out + (long long)tfirst * Smax + (act ? s : 0);
    const int s1 = dirn == 0 ? (s >= 1 ? s - 1 : 0) : (s + 1 < S ? s + 1 : 0);    // neighbour indices, clamped into the row
    const int s2 = dirn == 0 ? (s >= 2 ? s - 2 : 0) : (s + 2 < S ? s + 2 : 0);
    const bool has1 = dirn == 0 ? (s >= 1) : (s + 1 < S);
    const int last = Tb - 1;
    auto emission = [&](int i) {                                                  // frame index clamped: always a valid address
      const int ii = i < last ? i : last;
      return lgp[(long long)ii * lg_stride] - lsp[(long long)ii * ls_stride];
    };
    float nx0 = emission(1), nx1 = emission(2), nx2 = emission(3), nx3 = emission(4);
    const int sa = act ? s : 0;
    for (int i = 1; i < Tb; ++i) {
      const float lp = nx0;
      nx0 = nx1; nx1 = nx2; nx2 = nx3;
      nx3 = emission(i + 4);
      op += out_stride;
      const float a0 = prev[sa];
      const float a1 = has1 ? prev[s1] : NEG_INF;
      const float a2 = skip ? prev[s2] : NEG_INF;
      const float m = lse3(a0, a1, a2);
      const float v = (m == NEG_INF) ? NEG_INF : m + lp;
      if (act) {
        cur[s] = v;
        *op = v;
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
  } else {
    for (int i = 1; i < Tb; ++i) {
      const int t = tfirst + step * i;
      for (int s = threadIdx.x; s < S; s += nthr) {
        const int cls = (s & 1) ? lab[s >> 1] : 0;
        const float lp = logits[((long long)t * Bn + b) * ld + cls] - lse[t * Bn + b];
        float a0 = prev[s], a1, a2;
        if (dirn == 0) {
          const bool skip = (s >= 2) && (s & 1) && (lab[s >> 1] != lab[(s >> 1) - 1]);
          a1 = (s >= 1) ? prev[s - 1] : NEG_INF;
          a2 = skip ? prev[s - 2] : NEG_INF;
        } else {
          const bool skip = (s + 2 < S) && (s & 1) && (lab[s >> 1] != lab[(s >> 1) + 1]);
          a1 = (s + 1 < S) ? prev[s + 1] : NEG_INF;
          a2 = skip ? prev[s + 2] : NEG_INF;
        }
        const float m = lse3(a0, a1, a2);
        const float v = (m == NEG_INF) ? NEG_INF : m + lp;
        cur[s] = v;
        out[(long long)t * Smax + s] = v;
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
  }
  if (dirn == 0 && threadIdx.x == 0) {
    const float l1 = prev[S - 1];
    const float l2 = (S >= 2) ? prev[S - 2] : NEG_INF;
    nll[b] = -lse2(l1, l2);
  }
}

// grid = (ceil(T / TCH), B); block = 128 threads; dynamic LDS: q[C] + val[S] + nxt[U] (ints) + lab[U]
constexpr int TCH = 8;
__global__ __launch_bounds__(128) void ctc_grad_kernel(const float* __restrict__ logits, int ld, float* __restrict__ grad, int ldg, int T,
                                                       int Bn, int C, const int* __restrict__ targets,
                                                       const int* __restrict__ tgt_off, const int* __restrict__ in_lens,
                                                       const int* __restrict__ tgt_lens, const float* __restrict__ lse,
                                                       const float* __restrict__ ab, int Smax, const float* __restrict__ nll,
                                                       float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float bred[2];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TCH;
  const int Tb = min(in_lens[b], T), U = tgt_lens[b];
  const int S = 2 * U + 1;
  const float nl = nll[b];
  const bool feasible = (nl != INFINITY) && (nl == nl);
  float* q = smem;                      // [C]
  float* val = q + C;                   // [S]
  int* nxt = (int*)(val + Smax);        // [U]
  int* labs = nxt + (Smax / 2 + 1);     // [U]
  int* hd = labs + (Smax / 2 + 1);      // [U] 1 = first occurrence of its label (the thread that sums the chain)
  const int* lab = targets + tgt_off[b];
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int u = tid; u < U; u += nthr) labs[u] = lab[u];
  __syncthreads();
  // nxt[u] = next index with the same label, or -1 ; head flag encoded as nxt sign trick: store separately
  for (int u = tid; u < U; u += nthr) {
    const int l = labs[u];
    int n = -1;
    for (int v = u + 1; v < U; ++v)
      if (labs[v] == l) { n = v; break; }
    nxt[u] = n;
    int head = 1;                        // (was re-derived for every frame)
    for (int v = 0; v < u; ++v)
      if (labs[v] == l) { head = 0; break; }
    hd[u] = head;
  }
  __syncthreads();
  const float* alpha = ab + (((long long)0 * Bn + b) * T) * Smax;
  const float* beta = ab + (((long long)1 * Bn + b) * T) * Smax;
  for (int tt = 0; tt < TCH; ++tt) {
    const int t = t0 + tt;
    if (t >= T) break;
    float* g = grad + ((long long)t * Bn + b) * ldg;
    if (t >= Tb || !feasible) {
      for (int c = tid; c < C; c += nthr) g[c] = 0.f;
      continue;  // uniform across the block
    }
    const float* lg = logits + ((long long)t * Bn + b) * ld;
    const float ls = lse[t * Bn + b];
    for (int c = tid; c < C; c += nthr) q[c] = 0.f;
    float bsum = 0.f;
    for (int s = tid; s < S; s += nthr) {
      const int cls = (s & 1) ? labs[s >> 1] : 0;
      const float a = alpha[(long long)t * Smax + s] + beta[(long long)t * Smax + s];
      float v = 0.f;
      if (a != NEG_INF) v = expf(a - (lg[cls] - ls) + nl);
      if (s & 1) val[s >> 1] = v;
      else bsum += v;
    }
    bsum = wave_sum(bsum);
    if ((tid & 63) == 0) bred[tid >> 6] = bsum;
    __syncthreads();
    // heads: u is a head if no earlier index has the same label
    for (int u = tid; u < U; u += nthr) {
      const int l = labs[u];
      if (hd[u]) {
        float sacc = 0.f;
        for (int v = u; v >= 0; v = nxt[v]) sacc += val[v];
        q[l] += sacc;  // single writer per class (l != 0 guaranteed for labels; blank handled below)
      }
    }
    if (tid == 0) q[0] += bred[0] + bred[1];
    __syncthreads();
    for (int c = tid; c < C; c += nthr) g[c] = scale * (expf(lg[c] - ls) - q[c]);
    __syncthreads();
  }
}

// softmax over the last dim, one wave per row (InferenceBatchSoftmax, modules/blocks.py:59-64)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows,
                                                           int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + (long long)row * ldx;
  float* q = y + (long long)row * ldy;
  float m = NEG_INF;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(p[c] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < C; c += 64) q[c] = expf(p[c] - m) * inv;
}

}  // namespace

extern "C" int ds2_softmax_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int C, void* stream) {
  DS2_REQUIRE(x && y && rows > 0 && C > 0, "ds2_softmax_rows_f32: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C);
  DS2_LAUNCH_CHECK("softmax_rows_kernel");
  return 0;
}

namespace {
// loss = sum_b nll[b] / B in a fixed order (lane-strided partial sums, then a shuffle tree): one wave
__global__ __launch_bounds__(64) void batch_mean_kernel(const float* __restrict__ nll, int B, float* __restrict__ out) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += nll[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = s / (float)B;
}
}  // namespace

// out[0] = (sum of the per-utterance losses) / B: `loss = criterion(...) / inputs.size(0)` (trainers/deepspeech_trainer.py:110-112) on the device
extern "C" int ds2_ctc_batch_mean_f32(const float* nll_dev, int B, float* out_dev, void* stream) {
  DS2_REQUIRE(nll_dev && out_dev && B > 0, "ds2_ctc_batch_mean_f32: bad args");
  hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nll_dev, B, out_dev);
  DS2_LAUNCH_CHECK("batch_mean_kernel");
  return 0;
}

extern "C" size_t ds2_ctc_workspace_bytes(int T, int B, int max_target_len) {
  const size_t Smax = 2 * (size_t)max_target_len + 1;
  return align_up((size_t)T * B * sizeof(float), 256) + 2 * (size_t)B * T * Smax * sizeof(float);
}

// targets: flat labels (device int32), tgt_off[b] = start of utterance b in `targets` (device int32).
extern "C" int ds2_ctc_loss_f32(const float* logits, int ld, int T, int B, int C, const int* targets_dev, const int* tgt_off_dev,
                                const int* in_lens_dev, const int* tgt_lens_dev, int max_target_len, float* nll_dev,
                                float* grad, int ldg, float grad_scale, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(logits && targets_dev && tgt_off_dev && in_lens_dev && tgt_lens_dev && nll_dev, "ds2_ctc_loss_f32: null pointer");
  DS2_REQUIRE(T > 0 && B > 0 && C > 0 && max_target_len >= 0, "ds2_ctc_loss_f32: bad dims");
  DS2_REQUIRE(ws && ws_bytes >= ds2_ctc_workspace_bytes(T, B, max_target_len), "ds2_ctc_loss_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int Smax = 2 * max_target_len + 1;
  float* lse = (float*)ws;
  float* ab = (float*)((char*)ws + align_up((size_t)T * B * sizeof(float), 256));
  hipLaunchKernelGGL(ctc_lse_kernel, dim3(ceil_div(T * B, 4)), dim3(256), 0, s, logits, ld, T, B, C, in_lens_dev, lse);
  DS2_LAUNCH_CHECK("ctc_lse_kernel");
  int threads = ceil_div(Smax, 64) * 64;
  if (threads > 1024) threads = 1024;
  const size_t lds = (size_t)2 * Smax * sizeof(float);
  DS2_REQUIRE(lds <= 64 * 1024, "ds2_ctc_loss_f32: target too long for LDS lattice rows (Smax=%d)", Smax);
  hipLaunchKernelGGL(ctc_lattice_kernel, dim3(B, 2), dim3(threads), lds, s, logits, ld, T, B, C, targets_dev, tgt_off_dev,
                     in_lens_dev, tgt_lens_dev, (const float*)lse, ab, Smax, nll_dev);
  DS2_LAUNCH_CHECK("ctc_lattice_kernel");
  if (grad) {
    const size_t lds2 = ((size_t)C + Smax) * sizeof(float) + 3 * ((size_t)Smax / 2 + 1) * sizeof(int);
    DS2_REQUIRE(lds2 <= 64 * 1024, "ds2_ctc_loss_f32: C/S too large for LDS (C=%d Smax=%d)", C, Smax);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(ceil_div(T, TCH), B), dim3(128), lds2, s, logits, ld, grad, ldg, T, B, C, targets_dev,
                       tgt_off_dev, in_lens_dev, tgt_lens_dev, (const float*)lse, (const float*)ab, Smax, (const float*)nll_dev,
                       grad_scale);
    DS2_LAUNCH_CHECK("ctc_grad_kernel");
  }
  return 0;
}
