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
//   block = (b, output row o, 128 time steps); per kernel row kd: stage one input row (138 pixels x 64 B, 80-B pitch)
//   and that row's 11x2 weight fragments in LDS, then 22 MFMAs per wave (32 co x 32 t tile).  Loads of row kd+1 are
//   parked in registers while row kd is multiplied.
// dgrad = the same kernel on dY (channels-last) with per-parity re-packed weights (as in conv.hip).
#include "common.h"

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
  if (t < a.T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] + (a.bias ? a.bias[co] : 0.f);
      if (t >= len) v = 0.f;
      a.out[(((long long)b * CH + co) * a.Dtot + orow) * a.T + t] = v;
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
extern "C" int ds2_conv2_fwd_bf16(const void* a1_nhwc, const void* wf, const float* bias, const int* lens_dev, float* y2, int B, int D1,
                                  int T, void* stream) {
  DS2_REQUIRE(a1_nhwc && wf && y2, "ds2_conv2_fwd_bf16: null pointer");
  const int D2 = (D1 + 2 * 10 - 21) / 2 + 1;
  CArgs a{};
  a.in = (const __bf16*)a1_nhwc; a.wpk = (const __bf16*)wf; a.bias = bias; a.out = y2; a.lens = lens_dev;
  a.B = B; a.Din = D1; a.T = T; a.Dtot = D2; a.KD = 21; a.SD = 2; a.PD = 10; a.OS = 1; a.OO = 0;
  hipLaunchKernelGGL(conv2_bf16_kernel, dim3(ceil_div(T, TT), D2, B), dim3(256), 0, (hipStream_t)stream, a);
  DS2_LAUNCH_CHECK("conv2_bf16_kernel fwd");
  return 0;
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
    hipLaunchKernelGGL(conv2_bf16_kernel, dim3(ceil_div(T, TT), n_o, B), dim3(256), 0, (hipStream_t)stream, a);
  }
  DS2_LAUNCH_CHECK("conv2_bf16_kernel dgrad");
  return 0;
}
