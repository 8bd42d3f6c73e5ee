// conv2 forward / data-gradient with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulate and output.
// Used when the model runs with precision = "bf16" (BASELINE configs[2],[4]); the fp32 kernels in conv.hip stay the
// parity path.  Conv2d(32,32,(21,11),s=(2,1),p=(10,5)), deepspeech.py:64, and its dX.
//
// Operand layout: the GEMM K dimension is the INPUT CHANNEL (32 = 2 MFMA k-steps), so the activation tensor is read
// channels-last, (B, D, T, 32) bf16: an MFMA B fragment (lane = time step, 8 consecutive channels) is one aligned
// 16-byte read for every tap — the kernel-tap shift moves whole 64-byte pixels, never sub-fragment offsets.
// `ds2_nhwc_bf16_f32` produces that layout from the (B,32,D,T) fp32 tensors of the rest of the stack in one
// LDS-transposed pass.  Weights are pre-packed per kernel row as [kd][kt][kk][kgroup][m][8] bf16, i.e. already in
// A-fragment order: staging is a straight copy and fragment reads are conflict-free linear 512-byte runs.
//
//   block = (b, R output rows, 128 time steps); per kernel row kd: stage one input row (138 pixels x 64 B, 80-B pitch)
//   and that row's 11x2 weight fragments in LDS, then 22 R MFMAs per wave (R tiles of 32 co x 32 t).  Loads of row kd+1 are
//   parked in registers while row kd is multiplied.  (conv2_bf16_rows_kernel; the one-row conv2_bf16_kernel is the A/B reference.)
// dgrad = the same kernel on dY (channels-last) with per-parity re-packed weights (as in conv.hip).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TT = 128, KT = 11, PT = 5, CH = 32;
constexpr int NPIX = TT + KT - 1;        // 138 pixels per staged row
constexpr int IPITCH = 80;               // bytes per pixel in LDS (64 + 16 pad: conflict-free ds_read_b128 at pixel stride)
constexpr int WROW = KT * 2 * 2 * 32 * 16;   // bytes of packed weights per kernel row: [kt][kk][kgroup][m][16 B] = 22528

__device__ __attribute__((aligned(16))) float g_zero_cb[4] = {0.f, 0.f, 0.f, 0.f};

struct CArgs {
  const __bf16* in;   // (B, Din, T, 32) channels-last
  const __bf16* wpk;  // [KD][WROW bytes]
  const float* bias;  // [32] or null
  float* out;         // (B, 32, Dtot, T) fp32
  const int* lens;    // (B) or null
  int B, Din, T, Dtot, KD, SD, PD, OS, OO;
  int gx, gy, gz;     // conv2_bf16_rows_kernel: logical grid (time tiles, output-row groups, utterances) behind its 1-D launch
};

constexpr int IN_CHUNKS = NPIX * 4;                 // 16-byte chunks of one staged row (552)
constexpr int IN_IT = (IN_CHUNKS + 255) / 256;      // 3
constexpr int W_CHUNKS = WROW / 16;                 // 1408
constexpr int W_IT = (W_CHUNKS + 255) / 256;        // 6

__device__ __forceinline__ void load_row(const CArgs& a, int b, int f, int t0, int kd, f32x4 (&ri)[IN_IT], f32x4 (&rw)[W_IT]) {
  const bool rowok = f >= 0 && f < a.Din;
  const char* rowp = reinterpret_cast<const char*>(a.in + (((long long)b * a.Din + (rowok ? f : 0)) * a.T) * CH);
#pragma unroll
  for (int u = 0; u < IN_IT; ++u) {
    const int c = threadIdx.x + 256 * u;
    const int pix = c >> 2, q = c & 3;
    const int t = t0 - PT + pix;
    const bool ok = rowok && c < IN_CHUNKS && t >= 0 && t < a.T;
    const void* p = ok ? (const void*)(rowp + ((long long)t * CH) * 2 + q * 16) : (const void*)g_zero_cb;
    ri[u] = *reinterpret_cast<const f32x4*>(p);
  }
  const char* wp = reinterpret_cast<const char*>(a.wpk) + (long long)kd * WROW;
#pragma unroll
  for (int u = 0; u < W_IT; ++u) {
    const int c = threadIdx.x + 256 * u;
    rw[u] = *reinterpret_cast<const f32x4*>(wp + (c < W_CHUNKS ? c : 0) * 16);
  }
}
__device__ __forceinline__ void store_row(char* __restrict__ in_lds, char* __restrict__ w_lds, const f32x4 (&ri)[IN_IT], const f32x4 (&rw)[W_IT]) {
#pragma unroll
  for (int u = 0; u < IN_IT; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < IN_CHUNKS) *reinterpret_cast<f32x4*>(in_lds + (c >> 2) * IPITCH + (c & 3) * 16) = ri[u];
  }
#pragma unroll
  for (int u = 0; u < W_IT; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < W_CHUNKS) *reinterpret_cast<f32x4*>(w_lds + c * 16) = rw[u];
  }
}

__global__ __launch_bounds__(256) void conv2_bf16_kernel(CArgs a) {
  __shared__ __attribute__((aligned(16))) char in_lds[NPIX * IPITCH];    // 11040
  __shared__ __attribute__((aligned(16))) char w_lds[WROW];              // 22528
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int t0 = blockIdx.x * TT, o = blockIdx.y, b = blockIdx.z;
  const int len = a.lens ? min(a.lens[b], a.T) : a.T;
  const int orow = a.OS * o + a.OO;
  const int t = t0 + wave * 32 + l31;
  if (t0 >= len) {   // whole tile masked (MaskConv): zeros
    if (t < a.T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        a.out[(((long long)b * CH + co) * a.Dtot + orow) * a.T + t] = 0.f;
      }
    }
    return;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  f32x4 ri[IN_IT], rw[W_IT];
  load_row(a, b, a.SD * o - a.PD, t0, 0, ri, rw);
  // this lane's fragment bases: B operand (input): pixel (wave*32 + l31 + kt), channel group (kk*16 + half*8)
  //                              A operand (weights): [kt][kk][kgroup = half][m = l31]
  const char* bbase = in_lds + (wave * 32 + l31) * IPITCH + half * 16;
  const char* abase = w_lds + half * (32 * 16) + l31 * 16;
  for (int kd = 0; kd < a.KD; ++kd) {
    __syncthreads();
    store_row(in_lds, w_lds, ri, rw);
    __syncthreads();
    if (kd + 1 < a.KD) load_row(a, b, a.SD * o + (kd + 1) - a.PD, t0, kd + 1, ri, rw);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(abase + (kt * 2 + kk) * (2 * 32 * 16));
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bbase + kt * IPITCH + kk * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
      }
    }
  }
  // biases first, all 16 before any store: a load inside the store loop would put an s_waitcnt vmcnt(0) — which also waits for
  // every earlier store — in front of each store (one in-order memory counter), i.e. 16 serialised HBM write round trips
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[(r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  if (t < a.T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] + bv[r];
      if (t >= len) v = 0.f;
      a.out[(((long long)b * CH + co) * a.Dtot + orow) * a.T + t] = v;
    }
  }
}

// R output rows (o0 .. o0 + R - 1) per block: they read the same kernel-row weights and input rows SD apart, so the input rows live in a
// ring of (R - 1) SD + 1 LDS slots and every staged (input row, weight row) pair feeds 22 R MFMAs per wave instead of 22 - 1/R of the
// weight staging (22.5 KB per kernel row was 2/3 of what a block moved into LDS), of the L2 reads and of the barriers per MFMA.  Same
// accumulation order per output as the one-row kernel: bit-identical.
template <int SD, int R>
__global__ __launch_bounds__(256) void conv2_bf16_rows_kernel(CArgs a, int n_out, float* __restrict__ stat_part) {
  constexpr int RING = (R - 1) * SD + 1;                   // rows f0 + kd .. f0 + kd + (R - 1) SD are live at kernel row kd
  constexpr int SLOT = NPIX * IPITCH;
  __shared__ __attribute__((aligned(16))) char in_lds[RING * SLOT];
  __shared__ __attribute__((aligned(16))) char w_lds[WROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // 1-D launch, workgroup ids go round the 8 XCDs: XCD x takes a contiguous eighth of the (utterance, time tile, row group) list, row group
  // fastest — neighbouring row groups of one (utterance, time tile) read 19 of their 25 input rows in common and now meet in ONE L2 (the 3-D
  // launch put them on different XCDs: every XCD fetched nearly the whole input, 4.4 x the operand bytes on the fabric)
  int bx, by, bz;
  {
    const int total = a.gx * a.gy * a.gz, id = blockIdx.x;
    const int lin = (total & 7) == 0 ? (id & 7) * (total >> 3) + (id >> 3) : id;
    by = lin % a.gy;
    const int pair = lin / a.gy;
    bx = pair % a.gx; bz = pair / a.gx;
  }
  const int t0 = bx * TT, o0 = R * by, b = bz;
  const int len = a.lens ? min(a.lens[b], a.T) : a.T;
  const int t = t0 + wave * 32 + l31;
  auto out_ptr = [&](int o, int co) { return a.out + (((long long)b * CH + co) * a.Dtot + (a.OS * o + a.OO)) * a.T + t; };
  if (t0 >= len) {   // whole tile masked (MaskConv): zeros
    if (t < a.T) {
#pragma unroll
      for (int q = 0; q < R; ++q)
        if (o0 + q < n_out) {
#pragma unroll
          for (int r = 0; r < 16; ++r) *out_ptr(o0 + q, (r & 3) + 8 * (r >> 2) + 4 * half) = 0.f;
        }
    }
    return;
  }
  f32x16 acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int f0 = SD * o0 - a.PD;                          // input row of (o0, kd = 0); (o0 + q, kd) reads row f0 + kd + q SD
  f32x4 ri[IN_IT], rw[W_IT];
  auto load_in = [&](int f) {
    const bool rowok = f >= 0 && f < a.Din;
    const char* rowp = reinterpret_cast<const char*>(a.in + (((long long)b * a.Din + (rowok ? f : 0)) * a.T) * CH);
#pragma unroll
    for (int u = 0; u < IN_IT; ++u) {
      const int c = tid + 256 * u;
      const int pix = c >> 2, q = c & 3;
      const int tt = t0 - PT + pix;
      const bool ok = rowok && c < IN_CHUNKS && tt >= 0 && tt < a.T;
      const void* p = ok ? (const void*)(rowp + ((long long)tt * CH) * 2 + q * 16) : (const void*)g_zero_cb;
      ri[u] = *reinterpret_cast<const f32x4*>(p);
    }
  };
  auto store_in = [&](int slot) {
#pragma unroll
    for (int u = 0; u < IN_IT; ++u) {
      const int c = tid + 256 * u;
      if (c < IN_CHUNKS) *reinterpret_cast<f32x4*>(in_lds + slot * SLOT + (c >> 2) * IPITCH + (c & 3) * 16) = ri[u];
    }
  };
  auto load_w = [&](int kd) {
    const char* wp = reinterpret_cast<const char*>(a.wpk) + (long long)kd * WROW;
#pragma unroll
    for (int u = 0; u < W_IT; ++u) {
      const int c = tid + 256 * u;
      rw[u] = *reinterpret_cast<const f32x4*>(wp + (c < W_CHUNKS ? c : 0) * 16);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int u = 0; u < W_IT; ++u) {
      const int c = tid + 256 * u;
      if (c < W_CHUNKS) *reinterpret_cast<f32x4*>(w_lds + c * 16) = rw[u];
    }
  };
#pragma unroll
  for (int j = 0; j < RING - 1; ++j) {                     // the rows that are older than the one kernel row 0 brings in
    load_in(f0 + j);
    store_in(j);
  }
  load_in(f0 + RING - 1);
  load_w(0);
  const int pix_off = (wave * 32 + l31) * IPITCH + half * 16;
  const char* abase = w_lds + half * (32 * 16) + l31 * 16;
  int slot0 = 0;                                           // ring slot of row f0 + kd
  for (int kd = 0; kd < a.KD; ++kd) {
    __syncthreads();
    {
      int snew = slot0 + RING - 1;
      if (snew >= RING) snew -= RING;
      store_in(snew);
    }
    store_w();
    __syncthreads();
    if (kd + 1 < a.KD) {
      load_in(f0 + kd + RING);
      load_w(kd + 1);
    }
    const char* bq[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      int sl = slot0 + q * SD;
      if (sl >= RING) sl -= RING;
      bq[q] = in_lds + sl * SLOT + pix_off;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(abase + (kt * 2 + kk) * (2 * 32 * 16));
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(bq[q] + kt * IPITCH + kk * 32);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, x, acc[q], 0, 0, 0);
        }
      }
    }
    slot0 = slot0 + 1 == RING ? 0 : slot0 + 1;
  }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[(r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  float s1[16], s2[16];                                    // BatchNorm statistics of what this lane stores (stat_part != NULL)
#pragma unroll
  for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
  for (int q = 0; q < R; ++q)
    if (o0 + q < n_out) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (t >= len || t >= a.T) ? 0.f : acc[q][r] + bv[r];
        if (t < a.T) *out_ptr(o0 + q, (r & 3) + 8 * (r >> 2) + 4 * half) = v;
        s1[r] += v;
        s2[r] += v * v;
      }
    }
  if (stat_part) {
    // per-channel sums of the block: over the 32 lanes of a half-wave (xor shuffles stay inside it), then over the 4 waves in order
    __shared__ float red[4][32][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) { s1[r] += __shfl_xor(s1[r], m); s2[r] += __shfl_xor(s2[r], m); }
      if (l31 == 0) { const int co = (r & 3) + 8 * (r >> 2) + 4 * half; red[wave][co][0] = s1[r]; red[wave][co][1] = s2[r]; }
    }
    __syncthreads();
    if (tid < 64) {
      const int co = tid >> 1, w = tid & 1;
      const float sum = ((red[0][co][w] + red[1][co][w]) + red[2][co][w]) + red[3][co][w];
      const long long blk = ((long long)bz * a.gy + by) * a.gx + bx;
      stat_part[(blk * 32 + co) * 2 + w] = sum;
    }
  }
}

// (B, 32, D, T) fp32 -> (B, D, T, 32) bf16, 32 channels x 64 time steps per block through LDS
__global__ __launch_bounds__(256) void nhwc_cast_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int Bn, int D, int T) {
  __shared__ float tile[CH][65];
  const int t0 = blockIdx.x * 64, d = blockIdx.y, b = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = ty + 4 * i, t = t0 + tx;
    tile[c][tx] = (t < T) ? src[(((long long)b * CH + c) * D + d) * T + t] : 0.f;
  }
  __syncthreads();
  const int c = threadIdx.x & 31, tq = threadIdx.x >> 5;   // 8 time steps per pass
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int tl = tq + 8 * i, t = t0 + tl;
    if (t < T) dst[(((long long)b * D + d) * T + t) * CH + c] = (__bf16)tile[c][tl];
  }
}

// forward weights: W (32 co, 32 ci, 21, 11) fp32 -> [kd][kt][kk][kgroup][m = co][8: ci = kk*16 + kgroup*8 + j] bf16
// dgrad parity p : [e'][kt'][kk][kgroup][m = ci][8: co = kk*16 + kgroup*8 + j] = W[co][ci][2*(KDe-1-e') + p][10 - kt'],  e' < KDe
__global__ void pack_conv2_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ wf, __bf16* __restrict__ wd0, __bf16* __restrict__ wd1) {
  const int per_row = WROW / 2;   // bf16 elements per kernel row
  const int nf = 21 * per_row, n0 = 11 * per_row, n1 = 10 * per_row;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nf + n0 + n1) return;
  int which, i;
  if (idx < nf) { which = 0; i = idx; } else if (idx < nf + n0) { which = 1; i = idx - nf; } else { which = 2; i = idx - nf - n0; }
  const int j = i & 7, m = (i >> 3) & 31, kg = (i >> 8) & 1, kk = (i >> 9) & 1;
  const int rest = i >> 10;          // kdrow * 11 + kt
  const int kt = rest % KT, kdr = rest / KT;
  const int kch = kk * 16 + kg * 8 + j;
  float v;
  if (which == 0) {
    v = W[((m * 32 + kch) * 21 + kdr) * 11 + kt];                                  // m = co, kch = ci
    wf[i] = (__bf16)v;
  } else {
    const int p = which - 1, KDe = (21 - p + 1) / 2;
    v = W[((kch * 32 + m) * 21 + (2 * (KDe - 1 - kdr) + p)) * 11 + (10 - kt)];     // m = ci, kch = co
    (which == 1 ? wd0 : wd1)[i] = (__bf16)v;
  }
}

}  // namespace

// bytes of the three packed bf16 weight sets: [0] forward (21 rows), [1] dgrad even rows (11), [2] dgrad odd rows (10)
extern "C" size_t ds2_conv2_bf16_packed_bytes(int which) { return (size_t)(which == 0 ? 21 : which == 1 ? 11 : 10) * WROW; }

extern "C" int ds2_conv2_pack_bf16(const float* w2, void* wf, void* wd0, void* wd1, void* stream) {
  DS2_REQUIRE(w2 && wf && wd0 && wd1, "ds2_conv2_pack_bf16: null pointer");
  const int total = (21 + 11 + 10) * (WROW / 2);
  hipLaunchKernelGGL(pack_conv2_bf16_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w2, (__bf16*)wf, (__bf16*)wd0,
                     (__bf16*)wd1);
  DS2_LAUNCH_CHECK("pack_conv2_bf16_kernel");
  return 0;
}

// (B,32,D,T) fp32 -> (B,D,T,32) bf16 channels-last
extern "C" int ds2_nhwc_bf16_f32(const float* src, void* dst, int B, int D, int T, void* stream) {
  DS2_REQUIRE(src && dst && B > 0 && D > 0 && T > 0, "ds2_nhwc_bf16_f32: bad args");
  hipLaunchKernelGGL(nhwc_cast_kernel, dim3(ceil_div(T, 64), D, B), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, B, D, T);
  DS2_LAUNCH_CHECK("nhwc_cast_kernel");
  return 0;
}

// y2 (B,32,D2,T) fp32 = mask(conv2(a1) + b2), a1 given channels-last bf16 (B,D1,T,32)
extern "C" int ds2_conv2_fwd_bf16_stat_blocks(int B, int D1, int T) {
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  return ceil_div(T, TT) * ceil_div(D2, 3) * B;
}

// stat_part: NULL, or ds2_conv2_fwd_bf16_stat_blocks() x 32 x 2 floats: per-block (sum, sum of squares) of every output channel over
// what the block stored (masked frames count as zeros, exactly what BatchNorm2d sees) - ds2_chanstats_from_partials finishes them
extern "C" int ds2_conv2_fwd_bf16_stats(const void* a1_nhwc, const void* wf, const float* bias, const int* lens_dev, float* y2, int B, int D1,
                                        int T, float* stat_part, void* stream) {
  DS2_REQUIRE(a1_nhwc && wf && y2, "ds2_conv2_fwd_bf16: null pointer");
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  CArgs a{};
  a.in = (const __bf16*)a1_nhwc; a.wpk = (const __bf16*)wf; a.bias = bias; a.out = y2; a.lens = lens_dev;
  a.B = B; a.Din = D1; a.T = T; a.Dtot = D2; a.KD = 21; a.SD = 2; a.PD = 10; a.OS = 1; a.OO = 0;
  // three output rows per block (5 input-row slots + the weight row = 77 KB of LDS, two blocks per CU): 655 -> 561 us at c3; two rows
  // 614, four rows (one block per CU) 767.  DS2_CONV2_ROWS=1: the one-row kernel (A/B switch)
  static const char* rows_env = ds2_exp_getenv("DS2_CONV2_ROWS");
  DS2_REQUIRE(!stat_part || !(rows_env && rows_env[0] == '1'), "ds2_conv2_fwd_bf16_stats: the one-row kernel (DS2_CONV2_ROWS=1) has no statistics epilogue");
  // blocks of a fully masked tile return early: their slots must read as zeros
  if (stat_part) DS2_HIP(hipMemsetAsync(stat_part, 0, (size_t)ds2_conv2_fwd_bf16_stat_blocks(B, D1, T) * 64 * sizeof(float), (hipStream_t)stream));
  if (rows_env && rows_env[0] == '1') hipLaunchKernelGGL(conv2_bf16_kernel, dim3(ceil_div(T, TT), D2, B), dim3(256), 0, (hipStream_t)stream, a);
  else {
    a.gx = ceil_div(T, TT); a.gy = ceil_div(D2, 3); a.gz = B;
    hipLaunchKernelGGL((conv2_bf16_rows_kernel<2, 3>), dim3(a.gx * a.gy * a.gz), dim3(256), 0, (hipStream_t)stream, a, D2, stat_part);
  }
  DS2_LAUNCH_CHECK("conv2_bf16_kernel fwd");
  return 0;
}

extern "C" int ds2_conv2_fwd_bf16(const void* a1_nhwc, const void* wf, const float* bias, const int* lens_dev, float* y2, int B, int D1,
                                  int T, void* stream) {
  return ds2_conv2_fwd_bf16_stats(a1_nhwc, wf, bias, lens_dev, y2, B, D1, T, nullptr, stream);
}

// da1 (B,32,D1,T) fp32 = conv2^T(dy2), dy2 given channels-last bf16 (B,D2,T,32)
extern "C" int ds2_conv2_dgrad_bf16(const void* dy2_nhwc, const void* wd0, const void* wd1, float* da1, int B, int D1, int T, void* stream) {
  DS2_REQUIRE(dy2_nhwc && wd0 && wd1 && da1, "ds2_conv2_dgrad_bf16: null pointer");
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  for (int p = 0; p < 2; ++p) {
    const int KDe = (21 - p + 1) / 2;
    const int n_o = (D1 - p + 1) / 2;
    CArgs a{};
    a.in = (const __bf16*)dy2_nhwc; a.wpk = (const __bf16*)(p == 0 ? wd0 : wd1); a.bias = nullptr; a.out = da1; a.lens = nullptr;
    a.B = B; a.Din = D2; a.T = T; a.Dtot = D1; a.KD = KDe; a.SD = 1; a.PD = KDe - 6; a.OS = 2; a.OO = p;
    // four output rows per block here (rows one apart: 4 input-row slots): 2 x 347 -> 2 x 290 us
    static const char* rows_env = ds2_exp_getenv("DS2_CONV2_ROWS");
    if (rows_env && rows_env[0] == '1') hipLaunchKernelGGL(conv2_bf16_kernel, dim3(ceil_div(T, TT), n_o, B), dim3(256), 0, (hipStream_t)stream, a);
    else {
      a.gx = ceil_div(T, TT); a.gy = ceil_div(n_o, 4); a.gz = B;
      hipLaunchKernelGGL((conv2_bf16_rows_kernel<1, 4>), dim3(a.gx * a.gy * a.gz), dim3(256), 0, (hipStream_t)stream, a, n_o, (float*)nullptr);
    }
  }
  DS2_LAUNCH_CHECK("conv2_bf16_kernel dgrad");
  return 0;
}

// =============================================================================================================
// conv2 weight gradient with bf16 MFMA operands.
//   dW[co][ci][kd][kt] = sum_{b,o,t} dY[b,co,o,t] * A1[b,ci,2o+kd-10,t+kt-5]
// GEMM view per tap: M = co (32), N = ci (32), K = time.  Both tensors are read in their natural (B,C,D,T) layout as
// zero-padded bf16 copies (row pitch Tp = 8-aligned, 8 leading zeros: x = t + 8), so time is contiguous and an MFMA
// fragment is 8 consecutive time steps.  The tap shift s = kt-5 would make the dY fragment start at an arbitrary
// 2-byte offset; instead the dY tile is staged in LDS as 8 copies pre-shifted by r = 0..7 elements, and tap s reads
// copy (s mod 8) — every ds_read_b128 stays 16-byte aligned, no funnel shifts in the MFMA loop.
//   K runs over the INPUT time t' in [t0-8, t0+72):  D[co][ci] += dYcopy_r[co][t' - s] * A1[ci][t']
// block = (group of 3 kernel rows, chunk of (b,o) pairs); 4 waves split the 11 time taps (x 3 kernel rows each); accumulators live in registers
// across the whole (b,o,t) loop; ordered reduction over chunks afterwards (deterministic).
// =============================================================================================================
namespace {

constexpr int WT = 64;                  // output time steps per tile
constexpr int WK = WT + 16;             // K extent per tile (t' in [t0-8, t0+72)) = 5 MFMA k-steps of 16
constexpr int WPITCH = 176;             // bytes per LDS row (88 bf16): conflict-free ds_read_b128 at row stride
constexpr int KDG = 3;                  // kernel rows per block
constexpr int TAPS = KDG * KT;          // 33
constexpr int TPW = (TAPS + 3) / 4;     // taps per wave (9)

struct WArgs {
  const __bf16* a1p;   // (B, 32, D1, Tp) padded bf16
  const __bf16* dyp;   // (B, 32, D2, Tp)
  float* part;         // [chunks][7 groups][33 taps][32 co][32 ci]
  const int* lens;
  int B, D1, D2, T, Tp, pairs_per_chunk;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// The 8 pre-shifted copies of one 8-element dY chunk, built in registers: copy_r chunk m = elements [8 - r, 16 - r) of the
// 16-element run (S[m-1] | S[m]) — a word select for even r, one v_alignbit per word for odd r — and stored with ds_write_b128.
__device__ __forceinline__ void store_shifted_copies(const u32x4 lo, const u32x4 hi, char* dst /* copy 0, row co, chunk m */) {
  const unsigned W[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int w0 = (8 - r) >> 1;
    u32x4 o;
    if ((r & 1) == 0) {
      o = u32x4{W[w0], W[w0 + 1], W[w0 + 2], W[w0 + 3]};
    } else {
      o = u32x4{__builtin_amdgcn_alignbit(W[w0 + 1], W[w0], 16), __builtin_amdgcn_alignbit(W[w0 + 2], W[w0 + 1], 16),
                __builtin_amdgcn_alignbit(W[w0 + 3], W[w0 + 2], 16), __builtin_amdgcn_alignbit(W[w0 + 4], W[w0 + 3], 16)};
    }
    *reinterpret_cast<u32x4*>(dst + r * 32 * WPITCH) = o;
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv2_wgrad_bf16_kernel(WArgs a) {
  __shared__ __attribute__((aligned(16))) char dy_lds[8 * 32 * WPITCH];      // 45056: [r][co][88]
  __shared__ __attribute__((aligned(16))) char a1_lds[KDG * 32 * WPITCH];    // 16896: [kdl][ci][88] (80 used)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int grp = blockIdx.x, chunk = blockIdx.y;
  const int kd0 = grp * KDG;

  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // per-thread staging items: dY (co, m) for idx = tid, tid + 256 (< 32 * 11); A1 (kdl, ci, ch) for idx = tid + 256 k (< KDG * 32 * 10)
  constexpr int NDY = 2, NA1 = (KDG * 32 * 10 + 255) / 256;
  u32x4 dlo[NDY], dhi[NDY], av[NA1];

  const int npairs = a.B * a.D2;
  const int pend = min(npairs, (chunk + 1) * a.pairs_per_chunk);
  int p = chunk * a.pairs_per_chunk, t0 = 0, len = 0;
  auto settle = [&]() {                         // move (p, t0) to the next existing tile; false when the chunk is exhausted
    while (p < pend) {
      len = a.lens ? min(a.lens[p / a.D2], a.T) : a.T;
      if (t0 < len) return true;
      ++p;
      t0 = 0;
    }
    return false;
  };
  // dY source chunk c of row co covers x in [t0 - 8 + 8c, +8) (x = t + 8); only THIS tile's output steps (chunks 2..9) are
  // non-zero so that neighbouring tiles' contributions are not counted twice
  auto load_tile = [&]() {
    const int b = p / a.D2, o = p % a.D2;
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int idx = tid + 256 * k;
      const int co = idx / 11, m = idx % 11;
      const __bf16* row = a.dyp + (((long long)b * 32 + min(co, 31)) * a.D2 + o) * a.Tp;
      const int xl = t0 - 8 + m * 8, xh = xl + 8;           // copy_r[8m + e] = dY(x = t0 + 8m + e - r): source chunks m and m + 1
      const bool okl = idx < 352 && m >= 2 && m < 2 + WT / 8 && xl + 8 <= a.Tp;
      const bool okh = idx < 352 && m + 1 >= 2 && m + 1 < 2 + WT / 8 && xh + 8 <= a.Tp;
      dlo[k] = *reinterpret_cast<const u32x4*>(okl ? (const void*)(row + xl) : (const void*)g_zero_cb);
      dhi[k] = *reinterpret_cast<const u32x4*>(okh ? (const void*)(row + xh) : (const void*)g_zero_cb);
    }
#pragma unroll
    for (int k = 0; k < NA1; ++k) {
      const int c = tid + 256 * k;
      const int ch = c % 10, ci = (c / 10) % 32, kdl = c / 320;
      const int f = 2 * o + (kd0 + kdl) - 10;
      const int x0 = t0 + ch * 8;
      const bool ok = c < KDG * 320 && (kd0 + kdl) < 21 && f >= 0 && f < a.D1 && x0 + 8 <= a.Tp;
      av[k] = *reinterpret_cast<const u32x4*>(ok ? (const void*)(a.a1p + (((long long)b * 32 + ci) * a.D1 + f) * a.Tp + x0)
                                                 : (const void*)g_zero_cb);
    }
  };

  bool have = settle();
  if (have) load_tile();
  while (have) {
    __syncthreads();                            // previous tile's fragment reads are done
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int idx = tid + 256 * k;
      if (idx < 352) store_shifted_copies(dlo[k], dhi[k], dy_lds + (idx / 11) * WPITCH + (idx % 11) * 16);
    }
#pragma unroll
    for (int k = 0; k < NA1; ++k) {
      const int c = tid + 256 * k;
      if (c < KDG * 320) *reinterpret_cast<u32x4*>(a1_lds + ((c / 320) * 32 + (c / 10) % 32) * WPITCH + (c % 10) * 16) = av[k];
    }
    __syncthreads();
    t0 += WT;
    have = settle();
    if (have) load_tile();                      // next tile's global loads fly behind this tile's MFMAs
    // ---- MFMA: wave w owns the time taps kt = w, w+4, w+8 (< 11) for all three kernel rows of the group: 9 accumulators
    // (slot i = q * 3 + kdl).  Per k-step the dY fragment of a tap is shared by its 3 kernel rows and the A1 fragment of a kernel
    // row by the wave's taps: 6 LDS reads feed 9 MFMAs (18 before); no control flow in the loop (the missing 12th tap of wave 3
    // multiplies a valid-but-unused copy and is never stored).
    const char* ap[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int kt = min(wave + 4 * q, KT - 1);
      const int s = kt - PT;                              // dY index = t' - s
      const int r = s & 7;                                // copy
      const int joff = (s < 0) ? 8 : 0;                   // j = 16*ks + 8*half + joff
      ap[q] = dy_lds + (r * 32 + l31) * WPITCH + (8 * half + joff) * 2;
    }
    const char* bp = a1_lds + l31 * WPITCH + (8 * half) * 2;
#pragma unroll
    for (int ks = 0; ks < WK / 16; ++ks) {
      bf16x8 af[3], bfr[KDG];
#pragma unroll
      for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const bf16x8*>(ap[q] + ks * 32);
#pragma unroll
      for (int kdl = 0; kdl < KDG; ++kdl) bfr[kdl] = *reinterpret_cast<const bf16x8*>(bp + kdl * 32 * WPITCH + ks * 32);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int kdl = 0; kdl < KDG; ++kdl) acc[q * KDG + kdl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bfr[kdl], acc[q * KDG + kdl], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + 4 * (i / KDG), kdl = i % KDG;
    if (kt < KT) {
      const int tap = kdl * KT + kt;
      float* out = a.part + ((((long long)chunk * gridDim.x + grp) * TAPS + tap) * 32) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        out[co * 32 + l31] = acc[i][r];
      }
    }
  }
}

// Threads follow the PARTIAL layout ([group][tap][co][ci], ci fastest): every wave-load is one contiguous 256-byte piece of a slab and
// the scattered side is the 0.9 MB result.  (Indexed by the dW layout - kt fastest - each lane of a load hit its own cache line:
// 73 slabs x 236 k lines, 100 us.)  Chunks are added in index order, eight loads in flight.
__global__ void conv2_wgrad_bf16_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int chunks, int ngroups) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;         // ((grp * TAPS + tap) * 32 + co) * 32 + ci
  const int per_chunk = ngroups * TAPS * 32 * 32;
  if (j >= per_chunk) return;
  const int ci = j & 31, co = (j >> 5) & 31, tap = (j >> 10) % TAPS, grp = (j >> 10) / TAPS;
  const int kd = grp * KDG + tap / KT, kt = tap % KT;
  float s = 0.f;
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(long long)(c + k) * per_chunk + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; c < chunks; ++c) s += part[(long long)c * per_chunk + j];
  if (kd < 21) dW[((co * 32 + ci) * 21 + kd) * 11 + kt] = s;
}

// dst[r][x] = bf16(src[r][x - 8]) for 8 <= x < T + 8, else 0 ; dst pitch Tp (multiple of 8, >= T + 16)
__global__ __launch_bounds__(256) void padcast_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long R, int T, int Tp) {
  const int cq = Tp / 8;
  const long long total = R * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cq;
    const int x0 = (int)(i % cq) * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = x0 + e - 8;
      o[e] = (__bf16)((t >= 0 && t < T) ? src[r * T + t] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(dst + r * Tp + x0) = o;
  }
}

}  // namespace


// =============================================================================================================
// conv2 weight gradient from the CHANNELS-LAST operands the rest of the bf16 conv stack already has (round 5).
//   dW[co][ci][kd][kt] = sum_{b,o,t} dY[b,o,t][co] * A1[b, 2o+kd-10, t+kt-5][ci]       dY (B,D2,T,32), A1 (B,D1,T,32) bf16
// With time as the ROW index of both LDS images ([t][32 channels], 64-byte rows) the tap shift is a row offset — always aligned — so the
// eight pre-shifted dY copies of the kernel above (80 KB of ds_write_b128 per 64 output steps: as much LDS time as the MFMAs) are not
// needed: rows go global -> LDS by DMA (global_load_lds, one 1 KiB instruction = 16 time steps, no VGPR staging, no write pass) and an
// MFMA operand (32 channels x 16 steps, 8 consecutive steps per lane) is gathered by two ds_read_b64_tr_b16 — a 32-lane half of such a
// read covers 4 consecutive rows = 256 contiguous bytes = every bank once, no swizzle.  K runs over the OUTPUT time of a 128-step segment
// (no halo steps multiplied: 8 k-steps of 16, against 5 for 64 outputs above); the A1 rows are staged with 8 halo steps on either side.
// block = (group of 3 kernel rows, chunk of (b,o) pairs) as above — but all 7 groups of a chunk on ONE XCD and the chunk walked segment-major
// (L2 hit rate 10 -> 79 %, 2.2 GB -> 0.43 GB of fabric reads per launch): 4 waves split the 11 time taps, 9 accumulators per wave live in registers
// across the chunk; per k-step one dY fragment feeds all 9 MFMAs of the wave, each (kernel row, tap) has its own A1 fragment.
// Segments are double-buffered: the DMA of segment i+1 is issued behind the barrier that ends segment i-1's reads.
// Same partial layout as above (ordered reduction by conv2_wgrad_bf16_reduce_kernel: deterministic).
// =============================================================================================================
namespace {
constexpr int NS = 128;                          // output time steps per segment
constexpr int NA1 = NS + 16;                     // staged steps per A1 row: t in [t0 - 8, t0 + 136)
constexpr int SEG_DY = NS * 64;                  // 8192
constexpr int SEG_A1 = NA1 * 64;                 // 9216 per kernel row
constexpr int SEG_BYTES = SEG_DY + KDG * SEG_A1; // 35840
typedef __attribute__((address_space(3))) void lds_void_c;
typedef const __attribute__((address_space(1))) void gbl_void_c;
typedef float f32x2c __attribute__((ext_vector_type(2)));

struct W2Args {
  const __bf16* a1n;   // (B, D1, T, 32)
  const __bf16* dyn;   // (B, D2, T, 32)
  float* part;         // [chunks][7 groups][33 taps][32 co][32 ci]
  const int* lens;
  int B, D1, D2, T, pairs_per_chunk, nchunks, chunks_per_xcd;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv2_wgrad_nhwc_kernel(W2Args a) {
  extern __shared__ __attribute__((aligned(1024))) char seg[];      // [2][dY 128 x 64 B | A1 3 x 144 x 64 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // workgroup ids go round the 8 XCDs: XCD x hosts the chunks x * cpx .. + cpx - 1, ALL SEVEN kernel-row groups of each.  The groups of a
  // chunk walk the same (pair, segment) sequence in step, so the dY segment is fetched into that XCD's L2 once for the seven of them, and an A1
  // row (f = 2 o + kd - 10: group g needs at pair o what group g + 2 needed at pair o - 3) is still there when the next group asks for it.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = xcd * a.chunks_per_xcd + slot / 7, grp = slot % 7;
  if (slot >= 7 * a.chunks_per_xcd || chunk >= a.nchunks) return;
  const int kd0 = grp * KDG;

  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int npairs = a.B * a.D2;
  const int pbeg = chunk * a.pairs_per_chunk, pend = min(npairs, pbeg + a.pairs_per_chunk);
  int p = pbeg, t0 = 0;
  // SEGMENT-major walk (all pairs of the chunk at one t0, then the next t0): consecutive pairs are consecutive output rows o of one utterance,
  // whose A1 rows overlap (21 kernel rows, stride 2) - the window that has to stay in the L2 is 21 rows x one segment, not x the utterance
  auto settle = [&]() {                         // move (p, t0) to the next existing segment; false when the chunk is exhausted
    while (t0 < a.T) {
      while (p < pend) {
        const int len = a.lens ? min(a.lens[p / a.D2], a.T) : a.T;
        if (t0 < len) return true;
        ++p;
      }
      p = pbeg;
      t0 += NS;
    }
    return false;
  };
  // DMA of one segment: wave w moves dY pieces w, w + 4 (16 steps each) and A1 pieces w, w + 4, .. (< 27 = 3 rows x 9); a lane fetches the
  // 16-byte quarter (lane & 3) of step (lane >> 2) of its piece; steps outside the tensor read the zero page
  const int prow = lane >> 2, pq = (lane & 3) * 8;
  auto stage = [&](int buf, int sp, int st0) {
    const int b = sp / a.D2, o = sp % a.D2;
    char* base = seg + buf * SEG_BYTES;
    const __bf16* dyrow = a.dyn + ((long long)(b * a.D2 + o) * a.T) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave + 4 * i;
      const int t = st0 + piece * 16 + prow;
      const void* src = t < a.T ? (const void*)(dyrow + (long long)t * 32 + pq) : (const void*)g_zero_cb;
      __builtin_amdgcn_global_load_lds((gbl_void_c*)src, (lds_void_c*)(base + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int idx = wave + 4 * i;
      if (idx < KDG * 9) {
        const int kdl = idx / 9, piece = idx % 9;
        const int f = 2 * o + (kd0 + kdl) - 10;
        const int t = st0 - 8 + piece * 16 + prow;
        const bool ok = (kd0 + kdl) < 21 && f >= 0 && f < a.D1 && t >= 0 && t < a.T;
        const void* src = ok ? (const void*)(a.a1n + ((long long)(b * a.D1 + f) * a.T + t) * 32 + pq) : (const void*)g_zero_cb;
        __builtin_amdgcn_global_load_lds((gbl_void_c*)src, (lds_void_c*)(base + SEG_DY + kdl * SEG_A1 + piece * 1024), 16, 0, 0);
      }
    }
  };

  // fragment addresses (buffer 0, k-step 0, first of the two tr-reads): lane p of a 16-lane group addresses row (p >> 2), channels
  // 4 (p & 3) .. + 3 of the group's 16 channels; the second read is 4 rows (256 B) on, a k-step 16 rows (1024 B)
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_c*)seg;
  const int p16 = lane & 15, g16 = (lane >> 4) & 1;
  const unsigned lanepart = (unsigned)((half * 8 + (p16 >> 2)) * 64 + (g16 * 16 + 4 * (p16 & 3)) * 2);
  const unsigned dyaddr = lds0 + lanepart;
  unsigned a1addr[3];                            // per tap of this wave: the staged A1 row image starts at t0 - 8, tap kt reads step t + kt - 5
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int kt = min(wave + 4 * q, KT - 1);
    a1addr[q] = lds0 + SEG_DY + (unsigned)((kt - PT + 8) * 64) + lanepart;
  }

  bool have = settle();
  int cur = 0;
  int np = p, nt0 = t0;
  if (have) stage(0, p, t0);
  while (have) {
    // next segment of the chunk (its DMA goes out behind the barrier below)
    nt0 = t0; np = p + 1;
    { const int sp = p, st = t0; p = np; t0 = nt0; const bool more = settle(); np = p; nt0 = t0; p = sp; t0 = st; have = more; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of the current segment have landed
    __syncthreads();                                          // everybody's have; everybody is done reading the other buffer
    if (have) stage(cur ^ 1, np, nt0);
    const unsigned boff = (unsigned)(cur * SEG_BYTES);
    // 8 k-steps x 3 tap groups, software-pipelined by one group: the reads of group (ks, q + 1) — for q = 2: the next k-step's dY fragment
    // and its first group — are issued BEFORE the three MFMAs of group (ks, q), whose own fragments a counted lgkmcnt retires (LDS returns in
    // order; at most 8 + 6 = 14 reads in flight, the counter holds 15).  Fragment registers: two dY slots, two group slots.
    f32x4 fd[2], fg[2][KDG];
#define W2_RD(dst, addr, off)                                                                                                         \
    do {                                                                                                                              \
      f32x2c lo_, hi_;                                                                                                                \
      asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                        \
                   : "=&v"(lo_), "=&v"(hi_)                                                                                           \
                   : "v"(addr), "n"(off), "n"((off) + 256));                                                                          \
      dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);                                                                            \
    } while (0)
#define W2_GROUP_RD(slot, ks, q)                                                                                                      \
    do {                                                                                                                              \
      W2_RD(fg[slot][0], a1addr[q] + boff, (ks) * 1024);                                                                              \
      W2_RD(fg[slot][1], a1addr[q] + boff, (ks) * 1024 + SEG_A1);                                                                     \
      W2_RD(fg[slot][2], a1addr[q] + boff, (ks) * 1024 + 2 * SEG_A1);                                                                 \
    } while (0)
#define W2_GROUP_MMA(slot, dslot, q, NWAIT)                                                                                           \
    do {                                                                                                                              \
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fd[dslot]), "+v"(fg[slot][0]), "+v"(fg[slot][1]), "+v"(fg[slot][2]) : "n"(NWAIT) : "memory"); \
      _Pragma("unroll") for (int kdl = 0; kdl < KDG; ++kdl)                                                                           \
        acc[(q) * KDG + kdl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fd[dslot]), __builtin_bit_cast(bf16x8, fg[slot][kdl]), \
                                                                      acc[(q) * KDG + kdl], 0, 0, 0);                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                              \
    } while (0)
    W2_RD(fd[0], dyaddr + boff, 0);
    W2_GROUP_RD(0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < NS / 16; ++ks) {
      const int s0 = (3 * ks) & 1, d = ks & 1;               // slot of group (ks, 0); groups alternate slots
      W2_GROUP_RD(s0 ^ 1, ks, 1);
      __builtin_amdgcn_sched_barrier(0);
      W2_GROUP_MMA(s0, d, 0, 6);
      W2_GROUP_RD(s0, ks, 2);
      __builtin_amdgcn_sched_barrier(0);
      W2_GROUP_MMA(s0 ^ 1, d, 1, 6);
      if (ks + 1 < NS / 16) {
        W2_RD(fd[d ^ 1], dyaddr + boff, (ks + 1) * 1024);
        W2_GROUP_RD(s0 ^ 1, ks + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        W2_GROUP_MMA(s0, d, 2, 8);
      } else {
        W2_GROUP_MMA(s0, d, 2, 0);
      }
    }
#undef W2_RD
#undef W2_GROUP_RD
#undef W2_GROUP_MMA
    p = np; t0 = nt0; cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int kt = wave + 4 * (i / KDG), kdl = i % KDG;
    if (kt < KT) {
      const int tap = kdl * KT + kt;
      float* out = a.part + ((((long long)chunk * 7 + grp) * TAPS + tap) * 32) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        out[co * 32 + l31] = acc[i][r];
      }
    }
  }
}
}  // namespace

extern "C" int ds2_conv_padded_pitch(int T) { return (T + 16 + 7) / 8 * 8; }

// (R, T) fp32 -> (R, Tp) bf16 with 8 leading zeros and zero tail, Tp = ds2_conv_padded_pitch(T)
extern "C" int ds2_padcast_bf16(const float* src, void* dst, long long R, int T, void* stream) {
  DS2_REQUIRE(src && dst && R > 0 && T > 0, "ds2_padcast_bf16: bad args");
  const int Tp = ds2_conv_padded_pitch(T);
  long long blocks = (R * (Tp / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(padcast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, R, T, Tp);
  DS2_LAUNCH_CHECK("padcast_kernel");
  return 0;
}

extern "C" size_t ds2_conv2_wgrad_bf16_workspace_bytes(int B, int D1) {
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  const int pairs = B * D2;
  const int chunks = pairs < 73 ? pairs : 73;
  return (size_t)chunks * 7 * TAPS * 32 * 32 * sizeof(float);
}

// dW2 (32,32,21,11) fp32 from zero-padded bf16 copies a1p (B,32,D1,Tp), dy2p (B,32,D2,Tp)  [ds2_padcast_bf16]
extern "C" int ds2_conv2_wgrad_bf16(const void* a1p, const void* dy2p, const int* lens_dev, float* dW2, int B, int D1, int T, void* ws,
                                    size_t ws_bytes, void* stream) {
  DS2_REQUIRE(a1p && dy2p && dW2 && ws, "ds2_conv2_wgrad_bf16: null pointer");
  DS2_REQUIRE(ws_bytes >= ds2_conv2_wgrad_bf16_workspace_bytes(B, D1), "ds2_conv2_wgrad_bf16: workspace too small");
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  const int pairs = B * D2;
  int chunks = pairs < 73 ? pairs : 73;
  const int ppc = ceil_div(pairs, chunks);
  chunks = ceil_div(pairs, ppc);
  WArgs a{};
  a.a1p = (const __bf16*)a1p; a.dyp = (const __bf16*)dy2p; a.part = (float*)ws; a.lens = lens_dev;
  a.B = B; a.D1 = D1; a.D2 = D2; a.T = T; a.Tp = ds2_conv_padded_pitch(T); a.pairs_per_chunk = ppc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv2_wgrad_bf16_kernel, dim3(7, chunks), dim3(256), 0, s, a);
  DS2_LAUNCH_CHECK("conv2_wgrad_bf16_kernel");
  hipLaunchKernelGGL(conv2_wgrad_bf16_reduce_kernel, dim3(ceil_div(7 * TAPS * 32 * 32, 256)), dim3(256), 0, s, (const float*)ws, dW2, chunks, 7);
  DS2_LAUNCH_CHECK("conv2_wgrad_bf16_reduce_kernel");
  return 0;
}

// dW2 (32,32,21,11) fp32 from the channels-last bf16 operands a1 (B,D1,T,32) and dy2 (B,D2,T,32) [ds2_nhwc_bf16_f32 / the fused BatchNorm2d
// kernels' nhwc outputs]: conv2_wgrad_nhwc_kernel.  Same workspace as ds2_conv2_wgrad_bf16.  Replaces Conv2d's weight gradient under
// loss.backward() (asr_deepspeech/modules/deepspeech.py:64, trainers/deepspeech_trainer.py:87).
extern "C" int ds2_conv2_wgrad_nhwc_bf16(const void* a1_nhwc, const void* dy2_nhwc, const int* lens_dev, float* dW2, int B, int D1, int T, void* ws,
                                         size_t ws_bytes, void* stream) {
  DS2_REQUIRE(a1_nhwc && dy2_nhwc && dW2 && ws, "ds2_conv2_wgrad_nhwc_bf16: null pointer");
  DS2_REQUIRE(ws_bytes >= ds2_conv2_wgrad_bf16_workspace_bytes(B, D1), "ds2_conv2_wgrad_nhwc_bf16: workspace too small");
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  const int pairs = B * D2;
  // two workgroups per CU, 64 per XCD = 9 chunks x 7 kernel-row groups (+ one idle slot): at most 72 chunks
  constexpr int CPX = 9;
  int chunks = pairs < 8 * CPX ? pairs : 8 * CPX;
  const int ppc = ceil_div(pairs, chunks);
  chunks = ceil_div(pairs, ppc);
  W2Args a{};
  a.a1n = (const __bf16*)a1_nhwc; a.dyn = (const __bf16*)dy2_nhwc; a.part = (float*)ws; a.lens = lens_dev;
  a.B = B; a.D1 = D1; a.D2 = D2; a.T = T; a.pairs_per_chunk = ppc; a.nchunks = chunks; a.chunks_per_xcd = ceil_div(chunks, 8);
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    DS2_HIP(hipFuncSetAttribute((const void*)conv2_wgrad_nhwc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SEG_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL(conv2_wgrad_nhwc_kernel, dim3(8 * (7 * a.chunks_per_xcd + 1)), dim3(256), 2 * SEG_BYTES, s, a);
  DS2_LAUNCH_CHECK("conv2_wgrad_nhwc_kernel");
  hipLaunchKernelGGL(conv2_wgrad_bf16_reduce_kernel, dim3(ceil_div(7 * TAPS * 32 * 32, 256)), dim3(256), 0, s, (const float*)ws, dW2, chunks, 7);
  DS2_LAUNCH_CHECK("conv2_wgrad_bf16_reduce_kernel");
  return 0;
}
