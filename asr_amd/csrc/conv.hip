// DeepSpeech2 conv front-end on the CDNA4 f32 matrix cores (v_mfma_f32_32x32x2_f32), no im2col.
//   conv1: Conv2d(1,32,(41,11),s=(2,2),p=(20,5))    deepspeech.py:61
//   conv2: Conv2d(32,32,(21,11),s=(2,1),p=(10,5))   deepspeech.py:64
// forward, data-gradient (conv2 only: the spectrogram needs no grad) and weight-gradient, with the
// bias add and the MaskConv time mask (blocks.py:48-55) fused into the forward epilogue.
//
// Layout: activations (B, C, D, T) fp32 with T contiguous (the reference's own layout,
// functional.py:18).  The time axis is the GEMM N (fwd) / K (wgrad) dimension, so every global
// access is a unit-stride run along T.
//
// Implicit GEMM without im2col: for one (b, output row o) and one input channel, the KD input rows
// the taps touch are staged in LDS once; the MFMA B operand  B[k=(kd,kt)][n=t] = row_kd[ST*t + kt]
// is a *shifted read* of those rows (Toeplitz structure), i.e. a unit-stride ds_read_b32 per
// fragment.  The A operand is the weight slab  A[m=co][k] pre-packed as [stage][k][co].
//   fwd   : D[co][t]        = sum_k  Wpk[k][co] * row[k](t)             M=32, N=128 t / block
//   dgrad : same kernel on dY with per-parity re-packed (transposed, flipped) weights; the
//           stride-2 frequency axis splits the output rows into even/odd sub-problems.
//   wgrad : D[co][(kd,kt)] += sum_t dY[co][t] * row[kd](ST*t + kt)      M=32, N=taps, K=t
// A "stage" = (input channel, chunk of <= KDC kernel rows); conv1's 41 rows run as 2 stages of 21.
#include "common.h"

namespace {

constexpr int TT = 128;   // time tile per block (4 waves x 32)
constexpr int CO = 32;    // output channels of both convs

struct ConvArgs {
  const float* in;    // (B, Cin, Din, Tin)
  const float* wpk;   // [NS][KK2][32]
  const float* bias;  // [32] or null
  float* out;         // (B, 32, Dtot, Tout)
  const int* lens;    // (B) or null : zero output for t >= lens[b]
  int B, Cin, Din, Tin, Dtot, Tout;
  int KD;             // real kernel rows
  int SPC;            // stages per input channel = ceil(KD / KDC)
  int SD, PD, PT;     // in_row = SD*o + kd - PD ; in_t = ST*t + kt - PT
  int OS, OO;         // out_row = OS*o + OO
};

__device__ __attribute__((aligned(16))) float g_zero_conv[4] = {0.f, 0.f, 0.f, 0.f};   // global-address-space zero page

// Stage `NROWS` input rows of ROWLEN floats (rows kd0.. of channel ci, window starting at ST*t0 - PT) into LDS.
// Loads are issued in unrolled batches with out-of-range elements redirected to a zero page (pointer select, no
// branch), so each batch costs ONE memory round trip instead of one per element.
template <int NROWS, int ROWLEN, int ROWP, int ST>
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ in_bc /* in + (b*Cin+ci)*Din*Tin */,
                                           int Din, int Tin, int frow0 /* input row of local row 0 */, int nvalid_rows, int tstart) {
  constexpr int TOTAL = NROWS * ROWLEN;
  constexpr int NIT = (TOTAL + 255) / 256;
  constexpr int BATCH = 12;
#pragma unroll
  for (int n0 = 0; n0 < NIT; n0 += BATCH) {
    float v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = threadIdx.x + 256 * (n0 + u);
      const int kdl = idx / ROWLEN, i = idx - kdl * ROWLEN;
      const int f = frow0 + kdl, ti = tstart + i;
      const bool ok = (n0 + u < NIT) && idx < TOTAL && kdl < nvalid_rows && f >= 0 && f < Din && ti >= 0 && ti < Tin;
      const float* p = ok ? in_bc + (long long)f * Tin + ti : g_zero_conv;
      v[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = threadIdx.x + 256 * (n0 + u);
      const int kdl = idx / ROWLEN, i = idx - kdl * ROWLEN;
      if ((n0 + u < NIT) && idx < TOTAL) lds[kdl * ROWP + i] = v[u];
    }
  }
}

template <int KDC, int KT, int ST>
struct ConvGeom {
  static constexpr int ROWLEN = ST * TT + KT - 1;
  static constexpr int ROWP = (ROWLEN + 3) & ~3;
  static constexpr int KK = KDC * KT;
  static constexpr int KK2 = (KK + 1) & ~1;
};

template <int KDC, int KT, int ST>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
  using G = ConvGeom<KDC, KT, ST>;
  __shared__ __attribute__((aligned(16))) float in_lds[KDC * G::ROWP];
  __shared__ __attribute__((aligned(16))) float w_lds[G::KK2 * CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int t0 = blockIdx.x * TT, o = blockIdx.y, b = blockIdx.z;
  const int len = a.lens ? min(a.lens[b], a.Tout) : a.Tout;
  const int orow = a.OS * o + a.OO;
  const int t = t0 + wave * 32 + l31;

  if (t0 >= len) {  // whole tile masked: MaskConv zeroes it (bias included)
    if (t < a.Tout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
        a.out[(((long long)b * CO + co) * a.Dtot + orow) * a.Tout + t] = 0.f;
      }
    }
    return;
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int NS = a.Cin * a.SPC;
  const int bbase = ST * (wave * 32 + l31);
  for (int st = 0; st < NS; ++st) {
    const int ci = st / a.SPC, kd0 = (st % a.SPC) * KDC;
    __syncthreads();
    stage_rows<KDC, G::ROWLEN, G::ROWP, ST>(in_lds, a.in + ((long long)b * a.Cin + ci) * a.Din * a.Tin, a.Din, a.Tin,
                                            a.SD * o + kd0 - a.PD, a.KD - kd0, ST * t0 - a.PT);
    {
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + (long long)st * G::KK2 * CO);
      constexpr int NW4 = G::KK2 * CO / 4, WIT = (NW4 + 255) / 256;
      f32x4 wv[WIT];
#pragma unroll
      for (int u = 0; u < WIT; ++u) {
        const int idx = tid + 256 * u;
        wv[u] = wsrc[idx < NW4 ? idx : 0];
      }
#pragma unroll
      for (int u = 0; u < WIT; ++u) {
        const int idx = tid + 256 * u;
        if (idx < NW4) reinterpret_cast<f32x4*>(w_lds)[idx] = wv[u];
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < G::KK2 / 2; ++s) {
      const int klo = 2 * s, khi = 2 * s + 1;
      const int off_lo = (klo / KT) * G::ROWP + (klo % KT);
      const int off_hi = (khi < G::KK) ? (khi / KT) * G::ROWP + (khi % KT) : 0;
      const int off = half ? off_hi : off_lo;
      const float av = w_lds[(2 * s + half) * CO + l31];
      const float bv = in_lds[bbase + off];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  float bias_r[16];                               // all bias loads before the first store (one in-order memory counter)
#pragma unroll
  for (int r = 0; r < 16; ++r) bias_r[r] = a.bias ? a.bias[(r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  if (t < a.Tout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] + bias_r[r];
      if (t >= len) v = 0.f;
      a.out[(((long long)b * CO + co) * a.Dtot + orow) * a.Tout + t] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* in;    // (B, Cin, Din, Tin)
  const float* dy;    // (B, 32, Dy, Tout)   (time-masked)
  float* part;        // [chunks][32][NS][KK]
  const int* lens;    // (B) or null: tiles with t0 >= lens[b] are skipped (dy is zero there)
  int B, Cin, Din, Tin, Dy, Tout;
  int KD, SPC, SD, PD, PT;
  int pairs_per_chunk;
};

template <int KDC, int KT, int ST, int WPC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  using G = ConvGeom<KDC, KT, ST>;
  constexpr int NT = (G::KK + 31) / 32;
  constexpr int NTW = (NT + WPC - 1) / WPC;
  constexpr int SPB = 4 / WPC;
  constexpr int LDY = TT + 1;
  __shared__ float dy_lds[CO * LDY];
  __shared__ float in_lds[SPB * KDC * G::ROWP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int grp = blockIdx.x, chunk = blockIdx.y;
  const int NS = a.Cin * a.SPC;
  const int sl_mine = wave / WPC;
  const int nt0 = (wave % WPC) * NTW;

  f32x16 acc[NTW];
  int off[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int n = (nt0 + j) * 32 + l31;
    off[j] = (n < G::KK) ? (n / KT) * G::ROWP + (n % KT) : 0;
  }

  const int npairs = a.B * a.Dy;
  const int pbeg = chunk * a.pairs_per_chunk, pend = min(npairs, pbeg + a.pairs_per_chunk);
  const float* myin = in_lds + sl_mine * KDC * G::ROWP;
  for (int p = pbeg; p < pend; ++p) {
    const int b = p / a.Dy, o = p % a.Dy;
    const int len = a.lens ? min(a.lens[b], a.Tout) : a.Tout;
    for (int t0 = 0; t0 < len; t0 += TT) {
      __syncthreads();
      {   // dy tile: 32 rows (co) x TT, rows are D*T apart
        constexpr int NIT = CO * TT / 256;   // 16
        float v[NIT];
        const float* dyb = a.dy + (((long long)b * CO) * a.Dy + o) * a.Tout;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
          const int idx = tid + 256 * u;
          const int co = idx / TT, i = idx % TT;
          const int t = t0 + i;
          const float* p = (t < a.Tout) ? dyb + (long long)co * a.Dy * a.Tout + t : g_zero_conv;
          v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
          const int idx = tid + 256 * u;
          dy_lds[(idx / TT) * LDY + (idx % TT)] = v[u];
        }
      }
#pragma unroll
      for (int sl = 0; sl < SPB; ++sl) {
        const int st = grp * SPB + sl;
        const int ci = (st < NS) ? st / a.SPC : 0, kd0 = (st < NS) ? (st % a.SPC) * KDC : 0;
        stage_rows<KDC, G::ROWLEN, G::ROWP, ST>(in_lds + sl * KDC * G::ROWP, a.in + ((long long)b * a.Cin + ci) * a.Din * a.Tin, a.Din, a.Tin,
                                                a.SD * o + kd0 - a.PD, (st < NS) ? a.KD - kd0 : 0, ST * t0 - a.PT);
      }
      __syncthreads();
#pragma unroll 4
      for (int s = 0; s < TT / 2; ++s) {
        const int k = 2 * s + half;
        const float av = dy_lds[l31 * LDY + k];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const float bv = myin[off[j] + ST * k];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  const int st = grp * SPB + sl_mine;
  if (st < NS) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = (nt0 + j) * 32 + l31;
      if (nt0 + j < NT && n < G::KK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          a.part[(((long long)chunk * CO + co) * NS + st) * G::KK + n] = acc[j][r];
        }
      }
    }
  }
}

// dW[co][ci][kd][kt] = sum_chunks part[chunk][co][st][n]   (ordered => deterministic)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int chunks, int Cin, int KD, int KT,
                                    int KDC, int SPC, int accumulate) {
  const int total = CO * Cin * KD * KT;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kt = idx % KT, kd = (idx / KT) % KD, ci = (idx / (KT * KD)) % Cin, co = idx / (KT * KD * Cin);
  const int NS = Cin * SPC, KK = KDC * KT;
  const int st = ci * SPC + kd / KDC, n = (kd % KDC) * KT + kt;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(((long long)c * CO + co) * NS + st) * KK + n];
  if (accumulate) s += dW[idx];
  dW[idx] = s;
}

// W (32, Cin, KD, KT) -> wpk[st=(ci,chunk)][kidx=kdl*KT+kt][co], zero padded
__global__ void pack_fwd_kernel(const float* __restrict__ W, float* __restrict__ wpk, int Cin, int KD, int KT, int KDC, int SPC) {
  const int KK = KDC * KT, KK2 = (KK + 1) & ~1;
  const int total = Cin * SPC * KK2 * CO;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int co = idx % CO, kidx = (idx / CO) % KK2, st = idx / (CO * KK2);
  const int ci = st / SPC, kd = (st % SPC) * KDC + kidx / KT, kt = kidx % KT;
  float v = 0.f;
  if (kidx < KK && kd < KD) v = W[((co * Cin + ci) * KD + kd) * KT + kt];
  wpk[idx] = v;
}

// conv2 data-gradient weights for output-row parity p (f = 2*o + p):
//   wpk[st=co][kidx = e'*KT + kt'][ci] = W[co][ci][2*(KDe-1-e') + p][KT-1-kt'],  e' < KDe = (KD - p + 1)/2
__global__ void pack_dgrad_kernel(const float* __restrict__ W, float* __restrict__ wpk, int Cin, int KD, int KT, int KDC, int p) {
  const int KK = KDC * KT, KK2 = (KK + 1) & ~1;
  const int KDe = (KD - p + 1) / 2;
  const int total = CO * KK2 * Cin;  // Cin == 32 here (the MFMA M dimension)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % Cin, kidx = (idx / Cin) % KK2, co = idx / (Cin * KK2);
  const int e = kidx / KT, ktp = kidx % KT;
  float v = 0.f;
  if (kidx < KK && e < KDe) v = W[((co * Cin + ci) * KD + (2 * (KDe - 1 - e) + p)) * KT + (KT - 1 - ktp)];
  wpk[idx] = v;
}

constexpr int K1D = 41, K2D = 21, KTT = 11, KDC1 = 21, KDC2 = 21, KDCD = 11;

}  // namespace

static inline int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

extern "C" void ds2_conv_dims(int F, int Tin, int* D1, int* D2, int* T) {
  const int d1 = conv_out(F, 41, 2, 20);
  if (D1) *D1 = d1;
  if (D2) *D2 = conv_out(d1, 21, 2, 10);
  if (T) *T = conv_out(Tin, 11, 2, 5);
}

// packed-weight sizes (floats): [0] conv1 fwd, [1] conv2 fwd, [2] conv2 dgrad (both parities)
extern "C" size_t ds2_conv_packed_floats(int which) {
  const size_t kk2a = (size_t)((KDC1 * KTT + 1) & ~1), kk2d = (size_t)((KDCD * KTT + 1) & ~1);
  if (which == 0) return 2 * kk2a * CO;
  if (which == 1) return 32 * kk2a * CO;
  return 2 * 32 * kk2d * CO;
}

// Re-pack conv weights (call after every optimizer step / load_state_dict).
//   w1 (32,1,41,11) -> wpk1 ; w2 (32,32,21,11) -> wpk2 (fwd) and wpk2d (dgrad, 2 parities)
extern "C" int ds2_conv_pack_f32(const float* w1, const float* w2, float* wpk1, float* wpk2, float* wpk2d, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (w1 && wpk1) {
    const int total = (int)ds2_conv_packed_floats(0);
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, w1, wpk1, 1, K1D, KTT, KDC1, 2);
  }
  if (w2 && wpk2) {
    const int total = (int)ds2_conv_packed_floats(1);
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, w2, wpk2, 32, K2D, KTT, KDC2, 1);
  }
  if (w2 && wpk2d) {
    const int half = (int)ds2_conv_packed_floats(2) / 2;
    for (int p = 0; p < 2; ++p)
      hipLaunchKernelGGL(pack_dgrad_kernel, dim3(ceil_div(half, 256)), dim3(256), 0, s, w2, wpk2d + (size_t)p * half, 32, K2D, KTT,
                         KDCD, p);
  }
  DS2_LAUNCH_CHECK("conv pack kernels");
  return 0;
}

// y1 (B,32,D1,T) = mask(conv1(x (B,1,F,Tin)) + b1)
extern "C" int ds2_conv1_fwd_f32(const float* x, const float* wpk1, const float* bias, const int* lens_dev, float* y1, int B, int F,
                                 int Tin, void* stream) {
  DS2_REQUIRE(x && wpk1 && y1, "ds2_conv1_fwd_f32: null pointer");
  int D1, D2, T;
  ds2_conv_dims(F, Tin, &D1, &D2, &T);
  ConvArgs a{};
  a.in = x; a.wpk = wpk1; a.bias = bias; a.out = y1; a.lens = lens_dev;
  a.B = B; a.Cin = 1; a.Din = F; a.Tin = Tin; a.Dtot = D1; a.Tout = T;
  a.KD = K1D; a.SPC = 2; a.SD = 2; a.PD = 20; a.PT = 5; a.OS = 1; a.OO = 0;
  dim3 grid(ceil_div(T, TT), D1, B);
  hipLaunchKernelGGL((conv_fwd_kernel<KDC1, KTT, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DS2_LAUNCH_CHECK("conv1 fwd");
  return 0;
}

// y2 (B,32,D2,T) = mask(conv2(a1 (B,32,D1,T)) + b2)
extern "C" int ds2_conv2_fwd_f32(const float* a1, const float* wpk2, const float* bias, const int* lens_dev, float* y2, int B, int D1,
                                 int T, void* stream) {
  DS2_REQUIRE(a1 && wpk2 && y2, "ds2_conv2_fwd_f32: null pointer");
  const int D2 = conv_out(D1, 21, 2, 10);
  ConvArgs a{};
  a.in = a1; a.wpk = wpk2; a.bias = bias; a.out = y2; a.lens = lens_dev;
  a.B = B; a.Cin = 32; a.Din = D1; a.Tin = T; a.Dtot = D2; a.Tout = T;
  a.KD = K2D; a.SPC = 1; a.SD = 2; a.PD = 10; a.PT = 5; a.OS = 1; a.OO = 0;
  dim3 grid(ceil_div(T, TT), D2, B);
  hipLaunchKernelGGL((conv_fwd_kernel<KDC2, KTT, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DS2_LAUNCH_CHECK("conv2 fwd");
  return 0;
}

// da1 (B,32,D1,T) = conv2^T(dy2 (B,32,D2,T))   (no mask: the BN/Hardtanh backward masks)
extern "C" int ds2_conv2_dgrad_f32(const float* dy2, const float* wpk2d, float* da1, int B, int D1, int T, void* stream) {
  DS2_REQUIRE(dy2 && wpk2d && da1, "ds2_conv2_dgrad_f32: null pointer");
  const int D2 = conv_out(D1, 21, 2, 10);
  const size_t half = ds2_conv_packed_floats(2) / 2;
  for (int p = 0; p < 2; ++p) {
    const int KDe = (K2D - p + 1) / 2;
    const int n_o = (D1 - p + 1) / 2;  // rows f = 2*o + p < D1
    ConvArgs a{};
    a.in = dy2; a.wpk = wpk2d + p * half; a.bias = nullptr; a.out = da1; a.lens = nullptr;
    a.B = B; a.Cin = 32; a.Din = D2; a.Tin = T; a.Dtot = D1; a.Tout = T;
    a.KD = KDe; a.SPC = 1; a.SD = 1; a.PD = KDe - 6; a.PT = 5; a.OS = 2; a.OO = p;
    dim3 grid(ceil_div(T, TT), n_o, B);
    hipLaunchKernelGGL((conv_fwd_kernel<KDCD, KTT, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  DS2_LAUNCH_CHECK("conv2 dgrad");
  return 0;
}

extern "C" size_t ds2_conv_wgrad_workspace_bytes(int which, int B, int F) {
  int D1, D2, T;
  ds2_conv_dims(F, 64, &D1, &D2, &T);
  const size_t kk = (size_t)KDC1 * KTT;
  if (which == 0) {
    const int pairs = B * D1;
    const int chunks = pairs < 512 ? pairs : 512;
    return (size_t)chunks * CO * 2 * kk * sizeof(float);
  }
  const int pairs = B * D2;
  const int chunks = pairs < 64 ? pairs : 64;
  return (size_t)chunks * CO * 32 * kk * sizeof(float);
}

// dW1 (32,1,41,11) from x (B,1,F,Tin) and dy1 (B,32,D1,T) [masked]
extern "C" int ds2_conv1_wgrad_f32(const float* x, const float* dy1, const int* lens_dev, float* dW1, int B, int F, int Tin,
                                   int accumulate, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(x && dy1 && dW1 && ws, "ds2_conv1_wgrad_f32: null pointer");
  DS2_REQUIRE(ws_bytes >= ds2_conv_wgrad_workspace_bytes(0, B, F), "ds2_conv1_wgrad_f32: workspace too small");
  int D1, D2, T;
  ds2_conv_dims(F, Tin, &D1, &D2, &T);
  const int pairs = B * D1;
  int chunks = pairs < 512 ? pairs : 512;
  const int ppc = ceil_div(pairs, chunks);
  chunks = ceil_div(pairs, ppc);
  WgradArgs a{};
  a.in = x; a.dy = dy1; a.part = (float*)ws; a.lens = lens_dev;
  a.B = B; a.Cin = 1; a.Din = F; a.Tin = Tin; a.Dy = D1; a.Tout = T;
  a.KD = K1D; a.SPC = 2; a.SD = 2; a.PD = 20; a.PT = 5; a.pairs_per_chunk = ppc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((conv_wgrad_kernel<KDC1, KTT, 2, 2>), dim3(1, chunks), dim3(256), 0, s, a);
  DS2_LAUNCH_CHECK("conv1 wgrad");
  const int total = CO * 1 * K1D * KTT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, dW1, chunks, 1, K1D, KTT,
                     KDC1, 2, accumulate);
  DS2_LAUNCH_CHECK("conv1 wgrad reduce");
  return 0;
}

// dW2 (32,32,21,11) from a1 (B,32,D1,T) and dy2 (B,32,D2,T) [masked]
extern "C" int ds2_conv2_wgrad_f32(const float* a1, const float* dy2, const int* lens_dev, float* dW2, int B, int D1, int T,
                                   int accumulate, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(a1 && dy2 && dW2 && ws, "ds2_conv2_wgrad_f32: null pointer");
  const int D2 = conv_out(D1, 21, 2, 10);
  const int pairs = B * D2;
  int chunks = pairs < 64 ? pairs : 64;
  const int ppc = ceil_div(pairs, chunks);
  chunks = ceil_div(pairs, ppc);
  DS2_REQUIRE(ws_bytes >= (size_t)chunks * CO * 32 * KDC2 * KTT * sizeof(float), "ds2_conv2_wgrad_f32: workspace too small");
  WgradArgs a{};
  a.in = a1; a.dy = dy2; a.part = (float*)ws; a.lens = lens_dev;
  a.B = B; a.Cin = 32; a.Din = D1; a.Tin = T; a.Dy = D2; a.Tout = T;
  a.KD = K2D; a.SPC = 1; a.SD = 2; a.PD = 10; a.PT = 5; a.pairs_per_chunk = ppc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((conv_wgrad_kernel<KDC2, KTT, 1, 1>), dim3(8, chunks), dim3(256), 0, s, a);
  DS2_LAUNCH_CHECK("conv2 wgrad");
  const int total = CO * 32 * K2D * KTT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, dW2, chunks, 32, K2D, KTT,
                     KDC2, 1, accumulate);
  DS2_LAUNCH_CHECK("conv2 wgrad reduce");
  return 0;
}
