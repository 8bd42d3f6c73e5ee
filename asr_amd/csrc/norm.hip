// HBM-bound normalisation / reduction kernels of the DeepSpeech2 train step (fp32).
//
//   BatchNorm1d over (T*B, H) rows incl. padding rows   blocks.py:75,85-86 ; deepspeech.py:104
//   BatchNorm2d + Hardtanh(0,20) + MaskConv time mask   deepspeech.py:62-66 ; blocks.py:48-55
//   direction sum y = h_fwd + h_bwd                      blocks.py:92
//   (B,C*D,T) <-> (T,B,C*D) collapse/transposes          deepspeech.py:135-137
//
// All reductions are two-stage and ORDERED (per-chunk fp32 partials -> fp64 finalize): results
// are bit-reproducible run to run, no float atomics.  Every kernel is a coalesced stream:
// 16-byte loads along the contiguous axis, wavefront shuffles + LDS for the in-block reduction.
#include "common.h"
#include <algorithm>

namespace {

// (rows that are only 4-byte aligned — T = 501-style lengths — still move as ONE dword-aligned dwordx4 per lane: the hardware takes it, the
// aligned(4) vector type says so to the compiler; rounds 1-4 fell back to four scalar accesses there)
typedef float f32x4_dw __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
constexpr int BN_ROWS_PER_BLOCK = 32;
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p, int valid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else if (valid >= 4) {
    const f32x4_dw q = *reinterpret_cast<const f32x4_dw*>(p);
    v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
  }
  return v;
}
__device__ __forceinline__ void st4(float* __restrict__ p, f32x4 v, int valid, bool vec) {
  if (valid >= 4 && vec) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else if (valid >= 4) {
    *reinterpret_cast<f32x4_dw*>(p) = f32x4_dw{v.x, v.y, v.z, v.w};
  } else {
    if (valid > 0) p[0] = v.x;
    if (valid > 1) p[1] = v.y;
    if (valid > 2) p[2] = v.z;
    if (valid > 3) p[3] = v.w;
  }
}

// ------------------------------------------------------------------------------------------
// Column reductions over a row-major (M, H) matrix: two sums per column.
//   MODE 0: x = X            ; s0 = sum x, s1 = sum x^2
//   MODE 1: x = X + X2 -> Y  ; s0 = sum x, s1 = sum x^2            (direction sum + BN stats)
//   MODE 2: s0 = sum dY, s1 = sum dY * (X - mean) * rstd            (BN backward sums; X2 = dY)
// block = 256 threads: 16 column-quads (64 columns) x 16 row groups.  grid = (ceil(H/64), chunks)
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ X2,
                                                         int ldx2, float* __restrict__ Y, int ldy, int M, int H,
                                                         int rows_per_chunk, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps, float* __restrict__ part,
                                                         int vec) {
  __shared__ float red[16][64][2];
  const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cq * 4;
  const int valid = H - c0;
  const int rbeg = blockIdx.y * rows_per_chunk;
  const int rend = min(M, rbeg + rows_per_chunk);
  f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
  f32x4 mu = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
  if (MODE == 2 && valid > 0) {
    mu = ld4(mean + c0, valid, false);
    f32x4 vv = ld4(var + c0, valid, false);
    rs.x = rsqrtf(vv.x + eps); rs.y = rsqrtf(vv.y + eps); rs.z = rsqrtf(vv.z + eps); rs.w = rsqrtf(vv.w + eps);
  }
  if (valid > 0) {
    for (int r = rbeg + rg; r < rend; r += 16) {
      f32x4 x = ld4(X + (long long)r * ldx + c0, valid, vec);
      if (MODE == 1) {
        f32x4 x2 = ld4(X2 + (long long)r * ldx2 + c0, valid, vec);
        x += x2;
        st4(Y + (long long)r * ldy + c0, x, valid, vec);
      }
      if (MODE == 2) {
        f32x4 dy = ld4(X2 + (long long)r * ldx2 + c0, valid, vec);
        s0 += dy;
        s1 += dy * ((x - mu) * rs);
      } else {
        s0 += x;
        s1 += x * x;
      }
    }
  }
  red[rg][cq * 4 + 0][0] = s0.x; red[rg][cq * 4 + 0][1] = s1.x;
  red[rg][cq * 4 + 1][0] = s0.y; red[rg][cq * 4 + 1][1] = s1.y;
  red[rg][cq * 4 + 2][0] = s0.z; red[rg][cq * 4 + 2][1] = s1.z;
  red[rg][cq * 4 + 3][0] = s0.w; red[rg][cq * 4 + 3][1] = s1.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 1, w = threadIdx.x & 1;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][c][w];
    const int col = blockIdx.x * 64 + c;
    if (col < H) part[((long long)blockIdx.y * H + col) * 2 + w] = s;
  }
}

// finalize: sums over chunks in fp64.
//   kind 0: out0 = mean, out1 = biased var; optional running stats update (momentum, unbiased var)
//   kind 1: out0 = s0, out1 = s1 (raw sums)
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ part, int chunks, int H, double count, int kind,
                                                           float* __restrict__ out0, float* __restrict__ out1,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  // block = 32 columns x 8 chunk groups: the chunk loop (up to 2048 partials) is split 8 ways and combined in a fixed order
  __shared__ double red[8][32][2];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < H) {
    // four independent accumulator pairs: the loads of four partials are in flight together instead of one dependent load + add per trip
    double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
    int k = grp;
    for (; k + 24 < chunks; k += 32) {
      const float2 p0 = *reinterpret_cast<const float2*>(part + ((long long)k * H + c) * 2);
      const float2 p1 = *reinterpret_cast<const float2*>(part + ((long long)(k + 8) * H + c) * 2);
      const float2 p2 = *reinterpret_cast<const float2*>(part + ((long long)(k + 16) * H + c) * 2);
      const float2 p3 = *reinterpret_cast<const float2*>(part + ((long long)(k + 24) * H + c) * 2);
      a += (double)p0.x; b += (double)p0.y; a1 += (double)p1.x; b1 += (double)p1.y;
      a2 += (double)p2.x; b2 += (double)p2.y; a3 += (double)p3.x; b3 += (double)p3.y;
    }
    for (; k < chunks; k += 8) {
      a += (double)part[((long long)k * H + c) * 2 + 0];
      b += (double)part[((long long)k * H + c) * 2 + 1];
    }
    a = (a + a1) + (a2 + a3);
    b = (b + b1) + (b2 + b3);
  }
  red[grp][cl][0] = a;
  red[grp][cl][1] = b;
  __syncthreads();
  if (grp != 0 || c >= H) return;
  a = 0.0;
  b = 0.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    a += red[g][cl][0];
    b += red[g][cl][1];
  }
  if (kind == 0) {
    const double m = a / count;
    double v = b / count - m * m;
    if (v < 0.0) v = 0.0;
    out0[c] = (float)m;
    out1[c] = (float)v;
    if (run_mean) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
      run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
    }
  } else {
    out0[c] = (float)a;
    if (out1) out1[c] = (float)b;
  }
}

// Y = (X - mean) * rstd * gamma + beta        (M,H) elementwise.
// Thread = one fixed column quad (grid.x tiles the columns, grid.y the rows): the four per-column parameters are loaded and folded
// into (scale, shift) ONCE, then the thread streams rows with one 16-byte load and one store each — the earlier form re-loaded
// four parameter vectors per element quad and ran at a third of the HBM rate.  OUT_BF16: write the bf16 GEMM operand directly
// (row pitch ldy % 8 == 0, pad columns H..ldy zero).
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void bn1d_apply_kernel(const float* __restrict__ X, int ldx, void* __restrict__ Yv, int ldy, int M, int H,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int vec) {
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int wcols = OUT_BF16 ? ldy : H;                       // columns written (bf16: incl. the zero pad)
  if (c0 >= wcols) return;
  const int valid = H - c0;                                   // <= 0 in the pad columns
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (valid > 0) {
    const f32x4 mu = ld4(mean + c0, valid, false), vv = ld4(var + c0, valid, false);
    const f32x4 g = ld4(gamma + c0, valid, false), b = ld4(beta + c0, valid, false);
    sc.x = rsqrtf(vv.x + eps) * g.x; sc.y = rsqrtf(vv.y + eps) * g.y; sc.z = rsqrtf(vv.z + eps) * g.z; sc.w = rsqrtf(vv.w + eps) * g.w;
    sh = b - mu * sc;
    if (valid < 4) {                                          // lanes past H produce exact zeros
      if (valid < 2) { sc.y = 0.f; sh.y = 0.f; }
      if (valid < 3) { sc.z = 0.f; sh.z = 0.f; }
      sc.w = 0.f; sh.w = 0.f;
    }
  }
  const int rend = min(M, (int)(blockIdx.y + 1) * BN_ROWS_PER_BLOCK);
#pragma unroll 8
  for (int r = blockIdx.y * BN_ROWS_PER_BLOCK; r < rend; ++r) {
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
    if (valid > 0) {
      const f32x4 x = ld4(X + (long long)r * ldx + c0, valid, vec);
      y = x * sc + sh;                                        // (x - mu) * rstd * g + b with the per-column constants folded
    }
    if (OUT_BF16) *reinterpret_cast<nbf16x4*>(reinterpret_cast<__bf16*>(Yv) + (long long)r * ldy + c0) = nbf16x4{(__bf16)y.x, (__bf16)y.y, (__bf16)y.z, (__bf16)y.w};
    else st4(reinterpret_cast<float*>(Yv) + (long long)r * ldy + c0, y, valid, vec);
  }
}

// dX = gamma*rstd * (dY - s0/M - xhat * s1/M) ; dgamma = s1 ; dbeta = s0 (written by block 0)
__global__ __launch_bounds__(256) void bn1d_bwd_apply_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ X,
                                                             int ldx, float* __restrict__ dX, int lddx, int M, int H,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             const float* __restrict__ gamma, const float* __restrict__ s0,
                                                             const float* __restrict__ s1, float eps, float inv_count,
                                                             int vec) {
  // thread = one fixed column quad; per-column constants folded once:  dX = k1 * dY - k2 - k3 * (x - mu)
  //   k1 = gamma * rstd ; k2 = k1 * s0 / M ; k3 = k1 * rstd * s1 / M
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int valid = H - c0;
  if (valid <= 0) return;
  const f32x4 mu = ld4(mean + c0, valid, false), vv = ld4(var + c0, valid, false), g = ld4(gamma + c0, valid, false);
  const f32x4 a = ld4(s0 + c0, valid, false), b = ld4(s1 + c0, valid, false);
  f32x4 rs, k1, k2, k3;
  rs.x = rsqrtf(vv.x + eps); rs.y = rsqrtf(vv.y + eps); rs.z = rsqrtf(vv.z + eps); rs.w = rsqrtf(vv.w + eps);
  k1 = g * rs;
  k2 = k1 * a * inv_count;
  k3 = k1 * rs * b * inv_count;
  const int rend = min(M, (int)(blockIdx.y + 1) * BN_ROWS_PER_BLOCK);
#pragma unroll 8
  for (int r = blockIdx.y * BN_ROWS_PER_BLOCK; r < rend; ++r) {
    const f32x4 dy = ld4(dY + (long long)r * lddy + c0, valid, vec);
    const f32x4 x = ld4(X + (long long)r * ldx + c0, valid, vec);
    const f32x4 o = k1 * dy - k2 - k3 * (x - mu);
    st4(dX + (long long)r * lddx + c0, o, valid, vec);
  }
}

// ------------------------------------------------------------------------------------------
// Channel reductions over (B, C, D, T) fp32 (T contiguous), masked by out_len[b].
//   MODE 0: s0 = sum y, s1 = sum y^2    (y already masked to 0 beyond len: plain sums)
//   MODE 1: dz = dA * [0 < z < 20] * [t < len],  z = (y-mean)*rstd*gamma+beta
//           s0 = sum dz, s1 = sum dz * xhat
// grid = (C, chunks) ; a chunk = a range of (b, d) rows ; block 256 threads stride over T with float4
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ Yraw, const float* __restrict__ dA, int Bn,
                                                          int C, int D, int T, int rows_per_chunk,
                                                          const int* __restrict__ lens, const float* __restrict__ mean,
                                                          const float* __restrict__ var, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ part) {
  __shared__ float red[4][2];
  const int c = blockIdx.x;
  const int nrows = Bn * D;
  const int rbeg = blockIdx.y * rows_per_chunk, rend = min(nrows, rbeg + rows_per_chunk);
  float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
  if (MODE == 1) { mu = mean[c]; rs = rsqrtf(var[c] + eps); ga = gamma[c]; be = beta[c]; }
  float s0 = 0.f, s1 = 0.f;
  // one element of the stream: e = index inside the contiguous run (a multiple of T precedes it, so t = e mod T)
  auto take = [&](float y, float da, int t, int len) {
    if (MODE == 0) {
      s0 += y;
      s1 += y * y;
    } else {
      const float xh = (y - mu) * rs;
      const float z = xh * ga + be;
      const float dz = (t < len && z > 0.f && z < 20.f) ? da : 0.f;
      s0 += dz;
      s1 += dz * xh;
    }
  };
  // The rows d0 .. d0 + nd of one (b, c) plane are ONE contiguous run of nd * T floats: it is streamed linearly with aligned 16-byte
  // loads (a scalar head up to the first aligned address, a scalar tail) instead of row by row - T = 501 made every row start
  // misaligned, i.e. four scalar loads per thread and half the block idle (126 quads per row for 256 threads): 3.4 -> ~5 TB/s.
  for (int row = rbeg; row < rend;) {
    const int b = row / D, d0 = row - b * D;
    const int nd = min(D - d0, rend - row);
    const long long off = (((long long)b * C + c) * D + d0) * T;
    const float* py = Yraw + off;
    const float* pd = MODE == 1 ? dA + off : py;
    const int n = nd * T;
    const int len = (MODE == 1) ? min(lens[b], T) : T;
    const bool same = ((reinterpret_cast<uintptr_t>(py) ^ reinterpret_cast<uintptr_t>(pd)) & 15) == 0;
    const int head = same ? min(n, (int)(((16 - (reinterpret_cast<uintptr_t>(py) & 15)) & 15) >> 2)) : n;   // (unequal alignment: all scalar)
    const int nq = (n - head) >> 2;
    for (int e = threadIdx.x; e < head; e += 256) take(py[e], pd[e], e % T, len);
    for (int i = threadIdx.x; i < nq; i += 256) {
      const int e0 = head + 4 * i;
      const f32x4 y = *reinterpret_cast<const f32x4*>(py + e0);
      f32x4 da = y;
      if (MODE == 1) da = *reinterpret_cast<const f32x4*>(pd + e0);
      int t = e0 % T;
      take(y.x, da.x, t, len); t = t + 1 == T ? 0 : t + 1;
      take(y.y, da.y, t, len); t = t + 1 == T ? 0 : t + 1;
      take(y.z, da.z, t, len); t = t + 1 == T ? 0 : t + 1;
      take(y.w, da.w, t, len);
    }
    for (int e = head + 4 * nq + threadIdx.x; e < n; e += 256) take(py[e], pd[e], e % T, len);
    row += nd;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = s0; red[wave][1] = s1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    part[((long long)blockIdx.y * C + c) * 2 + threadIdx.x] = s;
  }
}

// a = mask(clamp((y-mean)*rstd*gamma+beta, 0, 20))  on (B,C,D,T)
__global__ __launch_bounds__(256) void bn2d_act_fwd_kernel(const float* __restrict__ Yraw, float* __restrict__ A, int Bn, int C, int D,
                                                           int T, const int* __restrict__ lens, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps) {
  const int tq = (T + 3) / 4;
  const long long total = (long long)Bn * C * D * tq;
  const bool vec = (T % 4) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % tq;
    const long long row = i / tq;                 // (b*C + c)*D + d
    const int c = (row / D) % C, b = row / ((long long)C * D);
    const int t0 = q * 4, valid = T - t0;
    const int len = min(lens[b], T);
    const float mu = mean[c], sc = rsqrtf(var[c] + eps) * gamma[c], be = beta[c];
    f32x4 y = ld4(Yraw + row * T + t0, valid, vec);
    f32x4 a;
    a.x = (t0 + 0 < len) ? fminf(fmaxf((y.x - mu) * sc + be, 0.f), 20.f) : 0.f;
    a.y = (t0 + 1 < len) ? fminf(fmaxf((y.y - mu) * sc + be, 0.f), 20.f) : 0.f;
    a.z = (t0 + 2 < len) ? fminf(fmaxf((y.z - mu) * sc + be, 0.f), 20.f) : 0.f;
    a.w = (t0 + 3 < len) ? fminf(fmaxf((y.w - mu) * sc + be, 0.f), 20.f) : 0.f;
    st4(A + row * T + t0, a, valid, vec);
  }
}

// dy = mask(gamma*rstd*(dz - s0/N - xhat*s1/N)), dz recomputed from dA, y
__global__ __launch_bounds__(256) void bn2d_act_bwd_apply_kernel(const float* __restrict__ Yraw, const float* __restrict__ dA,
                                                                 float* __restrict__ dY, int Bn, int C, int D, int T,
                                                                 const int* __restrict__ lens, const float* __restrict__ mean,
                                                                 const float* __restrict__ var, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ s0,
                                                                 const float* __restrict__ s1, float eps, float inv_count) {
  const int tq = (T + 3) / 4;
  const long long total = (long long)Bn * C * D * tq;
  const bool vec = (T % 4) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = i % tq;
    const long long row = i / tq;
    const int c = (row / D) % C, b = row / ((long long)C * D);
    const int t0 = q * 4, valid = T - t0;
    const int len = min(lens[b], T);
    const float mu = mean[c], rs = rsqrtf(var[c] + eps), ga = gamma[c], be = beta[c];
    const float m0 = s0[c] * inv_count, m1 = s1[c] * inv_count;
    f32x4 y = ld4(Yraw + row * T + t0, valid, vec);
    f32x4 da = ld4(dA + row * T + t0, valid, vec);
    float yy[4] = {y.x, y.y, y.z, y.w}, dd[4] = {da.x, da.y, da.z, da.w}, oo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (yy[j] - mu) * rs;
      const float z = xh * ga + be;
      const bool in = t0 + j < len;
      const float dz = (in && z > 0.f && z < 20.f) ? dd[j] : 0.f;
      oo[j] = in ? ga * rs * (dz - m0 - xh * m1) : 0.f;
    }
    f32x4 o = {oo[0], oo[1], oo[2], oo[3]};
    st4(dY + row * T + t0, o, valid, vec);
  }
}

// ------------------------------------------------------------------------------------------
// bf16-mode fusions of the BatchNorm2d passes with the layout casts that used to follow them.  One workgroup = the 32 channels x 64
// frames tile of one (b, d) row: wave w streams channels w, w + 4, ... with 64 consecutive frames per wave-load (T = 501-style row
// lengths leave the rows only 4-byte aligned, so the float4 path of the kernels above never triggered at the bench shapes — the plain
// dword stream below is the coalesced form), and emits up to three copies of the result from that ONE read:
//   f32  (B,32,D,T)   fp32      — what the fp32 consumers take (conv1's weight gradient casts dY1 on the fly)
//   pad  (B,32,D,Tp)  bf16      — zero-padded rows (8 leading zeros, zero tail): operand of conv2_wgrad_bf16 (was ds2_padcast_bf16)
//   nhwc (B,D,T,32)   bf16      — channels-last through LDS: operand of conv2 forward / dgrad (was ds2_nhwc_bf16_f32)
// The backward form also leaves the per-channel sums of dY (= the conv bias gradient) as one ordered partial per workgroup
// (deterministic two-stage reduction), instead of a further full pass over dY.
// ------------------------------------------------------------------------------------------
typedef __bf16 nbf16;
template <bool BWD>
__global__ __launch_bounds__(256) void bn2d_tile_kernel(const float* __restrict__ Yraw, const float* __restrict__ dA, int Bn, int D, int T, int Tp,
                                                        const int* __restrict__ lens, const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ s0, const float* __restrict__ s1, float eps, float inv_count,
                                                        float* __restrict__ out_f32, nbf16* __restrict__ out_pad, nbf16* __restrict__ out_nhwc,
                                                        float* __restrict__ chan_part) {
  constexpr int CH = 32;
  __shared__ float tile[CH][65];
  const int t0 = blockIdx.x * 64, d = blockIdx.y, b = blockIdx.z;
  const int len = min(lens[b], T);
  // thread = (channel tid >> 4 (+ 16), 4 consecutive frames): 16-byte lane accesses.  The rows are only 4-byte aligned (T = 501-style
  // lengths); the hardware takes dword-aligned dwordx4 accesses and the vector types below say so.  (Rounds 2-4 streamed these rows with
  // 4-byte lane accesses, eight per thread and tensor: the forward pass ran at 2.9 TB/s, 172 us; with the wide form 108 us.)
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const int l16 = threadIdx.x & 15, tl4 = l16 * 4, tq0 = t0 + tl4;
  const bool full4 = tq0 + 3 < T;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (threadIdx.x >> 4) + 16 * i;
    const long long row = ((long long)b * CH + c) * D + d;
    const float mu = mean[c], rs = rsqrtf(var[c] + eps), ga = gamma[c], be = beta[c];
    float y[4] = {0.f, 0.f, 0.f, 0.f}, da[4] = {0.f, 0.f, 0.f, 0.f};
    if (full4) {
      const f32x4u q = *reinterpret_cast<const f32x4u*>(Yraw + row * T + tq0);
      y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w;
      if (BWD) {
        const f32x4u g = *reinterpret_cast<const f32x4u*>(dA + row * T + tq0);
        da[0] = g.x; da[1] = g.y; da[2] = g.z; da[3] = g.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (tq0 + j < T) {
          y[j] = Yraw[row * T + tq0 + j];
          if (BWD) da[j] = dA[row * T + tq0 + j];
        }
    }
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = tq0 + j < len;
      if (!BWD) {
        r[j] = live ? fminf(fmaxf((y[j] - mu) * (rs * ga) + be, 0.f), 20.f) : 0.f;
      } else {
        const float xh = (y[j] - mu) * rs;
        const float z = xh * ga + be;
        const float dz = (live && z > 0.f && z < 20.f) ? da[j] : 0.f;
        r[j] = live ? ga * rs * (dz - s0[c] * inv_count - xh * (s1[c] * inv_count)) : 0.f;
      }
      tile[c][tl4 + j] = r[j];
    }
    if (out_f32) {
      if (full4) *reinterpret_cast<f32x4u*>(out_f32 + row * T + tq0) = f32x4u{r[0], r[1], r[2], r[3]};
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (tq0 + j < T) out_f32[row * T + tq0 + j] = r[j];
      }
    }
    if (out_pad) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (tq0 + j < T) out_pad[row * Tp + 8 + tq0 + j] = (nbf16)r[j];
      // the row's zero frame: 8 leading zeros (first tile), tail (last tile)
      if (blockIdx.x == 0 && l16 < 8) out_pad[row * Tp + l16] = (nbf16)0.f;
      if (blockIdx.x == gridDim.x - 1) for (int q = l16; T + 8 + q < Tp; q += 16) out_pad[row * Tp + T + 8 + q] = (nbf16)0.f;
    }
    if (BWD && chan_part) {
      // the tile's sum of dY for this channel: 4 values per lane in order, then the 16 lanes of the channel by a fixed butterfly
      float sum = ((r[0] + r[1]) + r[2]) + r[3];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) sum += __shfl_xor(sum, m);
      const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      if (l16 == 0) chan_part[blk * CH + c] = sum;
    }
  }
  if (out_nhwc) {
    __syncthreads();
    // one 16-byte store per thread: 8 channels of one frame; a wave writes 16 consecutive frames = 1 KiB contiguous (2-byte lane stores —
    // eight 128-byte wave-stores per thread — held the forward pass at 2.9 TB/s)
    typedef nbf16 nbf16x8 __attribute__((ext_vector_type(8)));
    const int c8 = (threadIdx.x & 3) * 8, tl = threadIdx.x >> 2, tt = t0 + tl;
    if (tt < T) {
      nbf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (nbf16)tile[c8 + j][tl];
      *reinterpret_cast<nbf16x8*>(out_nhwc + (((long long)b * D + d) * T + tt) * CH + c8) = o;
    }
  }
}

// bf16 mode, second conv stage: BatchNorm2d + Hardtanh + mask fused with the (B, 32*D, T) -> (T, B, 32*D) collapse (deepspeech.py:135-137)
// and the cast to the first recurrent layer's bf16 GEMM operand — one pass over y2 instead of three (BN apply, transpose, cast).
// 32 features x 32 frames LDS tiles; feature f = c*D + d, so the BatchNorm parameters are per tile ROW.
__global__ __launch_bounds__(256) void bn2d_act_collapse_kernel(const float* __restrict__ Yraw, int Bn, int D, int T, const int* __restrict__ lens,
                                                                const float* __restrict__ mean, const float* __restrict__ var,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                float* __restrict__ out_f32, nbf16* __restrict__ out_bf, int ldb) {
  // 64 features x 32 frames per block; 16-byte lane accesses: reads = (feature tid >> 3 (+ 32), 4 consecutive frames: dword-aligned dwordx4,
  // see bn2d_tile_kernel), bf16 writes = (frame tid >> 3, 8 consecutive features) = 128 contiguous bytes per frame
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef nbf16 nbf16x8 __attribute__((ext_vector_type(8)));
  __shared__ float tile[64][33];
  const int F = 32 * D;
  const int b = blockIdx.z;
  const int f0 = blockIdx.y * 64, t0 = blockIdx.x * 32;
  const int len = min(lens[b], T);
  {
    const int q4 = (threadIdx.x & 7) * 4, t = t0 + q4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int fl = (threadIdx.x >> 3) + 32 * i, f = f0 + fl;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (f < F && t < len) {
        const int c = f / D;
        const float sc = rsqrtf(var[c] + eps) * gamma[c], mu = mean[c], be = beta[c];
        const float* p = Yraw + ((long long)b * F + f) * T + t;
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (t + 3 < T) { const f32x4u q = *reinterpret_cast<const f32x4u*>(p); y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w; }
        else for (int j = 0; j < 4; ++j) if (t + j < T) y[j] = p[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = (t + j < len) ? fminf(fmaxf((y[j] - mu) * sc + be, 0.f), 20.f) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[fl][q4 + j] = a[j];
    }
  }
  __syncthreads();
  const int tl = threadIdx.x >> 3, t = t0 + tl, f8 = (threadIdx.x & 7) * 8, f = f0 + f8;
  if (t < T) {
    const long long row = (long long)t * Bn + b;
    if (out_f32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (f + j < F) out_f32[row * F + f + j] = tile[f8 + j][tl];
    }
    if (out_bf && f < ldb) {                                 // (F and ldb are multiples of 8: an 8-run is either inside F, or all padding)
      nbf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f + j < F) ? (nbf16)tile[f8 + j][tl] : (nbf16)0.f;
      *reinterpret_cast<nbf16x8*>(out_bf + row * ldb + f) = o;
    }
  }
}

// out[c] = sum over n ordered partials part[k][c] (fp64 combine): the tile kernels' per-workgroup channel sums (C = 32).
// Two ordered stages with coalesced reads: stage 0, block j sums rows [j*rpb, (j+1)*rpb) of the (n, 32) partial matrix into tmp[j][c]
// (lanes = channels: every wave-load is one contiguous 128-byte row piece); stage 1, one block sums the tmp rows.
__global__ __launch_bounds__(256) void chan_part_finalize_kernel(const float* __restrict__ part, long long n, long long rpb, double* __restrict__ tmp,
                                                                 int nblk, float* __restrict__ out, int stage) {
  __shared__ double red[8][32];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  double s = 0.0;
  if (stage == 0) {
    const long long k0 = (long long)blockIdx.x * rpb, k1 = k0 + rpb < n ? k0 + rpb : n;
    long long k = k0 + r;
    for (; k + 24 < k1; k += 32) {                          // four loads in flight, added in order
      const float v0 = part[k * 32 + c], v1 = part[(k + 8) * 32 + c], v2 = part[(k + 16) * 32 + c], v3 = part[(k + 24) * 32 + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; k < k1; k += 8) s += (double)part[k * 32 + c];
  } else {
    for (int k = r; k < nblk; k += 8) s += tmp[(long long)k * 32 + c];
  }
  red[r][c] = s;
  __syncthreads();
  if (r == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    if (stage == 0) tmp[(long long)blockIdx.x * 32 + c] = t;
    else out[c] = (float)t;
  }
}

// (B, F, T) -> (T, B, F)  [dir 0]   or   (T, B, F) -> (B, F, T)  [dir 1] ; 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_bft_kernel(const float* __restrict__ src, float* __restrict__ dst, int Bn, int F, int T,
                                                            int dir) {
  // 64 x 64 tiles, 16-byte lane accesses on both sides (rows of length T are only 4-byte aligned: dword-aligned dwordx4, see bn2d_tile_kernel);
  // thread = (row tid >> 4 (+ 16 i), 4 consecutive elements).  (32 x 32 tiles with 4-byte lanes before: 103 us for the 336 MB of c3's backward.)
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int f0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int q4 = (threadIdx.x & 15) * 4, r16 = threadIdx.x >> 4;
  // the (B,F,T) side: row = feature, run along t;  the (T,B,F) side: row = frame, run along f
  auto bft = [&](int f, int t) { return ((long long)b * F + f) * T + t; };
  auto tbf = [&](int t, int f) { return ((long long)t * Bn + b) * F + f; };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r16 + 16 * i;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (dir == 0) {                                          // read (B,F,T): feature f0 + r, frames t0 + q4 ..
      const int f = f0 + r, t = t0 + q4;
      if (f < F) {
        if (t + 3 < T) { const f32x4u q = *reinterpret_cast<const f32x4u*>(src + bft(f, t)); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (int j = 0; j < 4; ++j) if (t + j < T) v[j] = src[bft(f, t + j)];
      }
    } else {                                                 // read (T,B,F): frame t0 + r, features f0 + q4 ..
      const int t = t0 + r, f = f0 + q4;
      if (t < T) {
        if (f + 3 < F) { const f32x4u q = *reinterpret_cast<const f32x4u*>(src + tbf(t, f)); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (int j = 0; j < 4; ++j) if (f + j < F) v[j] = src[tbf(t, f + j)];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][q4 + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r16 + 16 * i;
    const f32x4u o = {tile[q4][r], tile[q4 + 1][r], tile[q4 + 2][r], tile[q4 + 3][r]};
    if (dir == 0) {                                          // write (T,B,F): frame t0 + r, features f0 + q4 ..
      const int t = t0 + r, f = f0 + q4;
      if (t < T) {
        if (f + 3 < F) *reinterpret_cast<f32x4u*>(dst + tbf(t, f)) = o;
        else for (int j = 0; j < 4; ++j) if (f + j < F) dst[tbf(t, f + j)] = o[j];
      }
    } else {                                                 // write (B,F,T): feature f0 + r, frames t0 + q4 ..
      const int f = f0 + r, t = t0 + q4;
      if (f < F) {
        if (t + 3 < T) *reinterpret_cast<f32x4u*>(dst + bft(f, t)) = o;
        else for (int j = 0; j < 4; ++j) if (t + j < T) dst[bft(f, t + j)] = o[j];
      }
    }
  }
}

// generic 2-D transpose: src (R, Cc) ld -> dst (Cc, R) ld ; batched via blockIdx.z
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, int lds_, long long ss, float* __restrict__ dst,
                                                          int ldd, long long sd, int R, int Cc) {
  __shared__ float tile[32][33];
  src += (long long)blockIdx.z * ss;
  dst += (long long)blockIdx.z * sd;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < Cc) ? src[(long long)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < Cc) dst[(long long)c * ldd + r] = tile[tx][ty + 8 * i];
  }
}


// ------------------------------------------------------------------------------------------
// BatchNorm1d folded into the projection (round 6, DS2_BN_FOLD): the direction sum + statistics pass writes the CENTRED sum as the bf16 GEMM
// operand and nothing else.
//   y = Xa + Xb (blocks.py:92);  m0 = column mean of y from the per-tile sums of h the forward recurrence emitted (hsum: no pass over y);
//   yc = y - m0 -> bf16 (M, ldy);  partials of sum yc and sum yc^2 (fp32 values, before the rounding)
// finalize: delta = sum yc / M (round-off of m0: ~1e-7), mean = m0 + delta, var = sum yc^2 / M - delta^2 — statistics of y itself, exact to
// fp32 round-off, from ONE pass (the centring is what lets a single pass be as accurate as the two-pass form).
// grid = (ceil(H/64), chunks), block = 16 column-quads x 16 row groups, as col_reduce_kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_colstats_kernel(const float* __restrict__ Xa, int lda, const float* __restrict__ Xb, int ldb,
                                                              const float* __restrict__ hsum, int ntiles, __bf16* __restrict__ Yc, int ldy,
                                                              int M, int H, int rows_per_chunk, float inv_m, float* __restrict__ m0_out,
                                                              float* __restrict__ part, int vec) {
  __shared__ float red[16][64][2];
  const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cq * 4;
  const int valid = H - c0;
  const int rbeg = blockIdx.y * rows_per_chunk;
  const int rend = min(M, rbeg + rows_per_chunk);
  f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, m0 = {0, 0, 0, 0};
  if (valid > 0) {
    // forward tiles first, then reverse tiles, in tile order: every block computes the same bits
    for (int d = 0; d < 2; ++d)
      for (int t = 0; t < ntiles; ++t) m0 += ld4(hsum + ((long long)d * ntiles + t) * H + c0, valid, false);
    m0 *= inv_m;
    if (blockIdx.y == 0 && rg == 0) st4(m0_out + c0, m0, valid, false);
    for (int r = rbeg + rg; r < rend; r += 16) {
      const f32x4 x = ld4(Xa + (long long)r * lda + c0, valid, vec) + ld4(Xb + (long long)r * ldb + c0, valid, vec);
      const f32x4 xc = x - m0;
      s0 += xc;
      s1 += xc * xc;
      if (valid >= 4) {
        *reinterpret_cast<nbf16x4*>(Yc + (long long)r * ldy + c0) = nbf16x4{(__bf16)xc.x, (__bf16)xc.y, (__bf16)xc.z, (__bf16)xc.w};
      } else {
        const float v[4] = {xc.x, xc.y, xc.z, xc.w};
        for (int e = 0; e < 4; ++e) Yc[(long long)r * ldy + c0 + e] = (__bf16)(e < valid ? v[e] : 0.f);
      }
    }
  } else if (c0 < ldy) {                                   // pad columns H .. ldy of the GEMM operand: zeros
    for (int r = rbeg + rg; r < rend; r += 16)
      for (int e = 0; e < 4 && c0 + e < ldy; ++e) Yc[(long long)r * ldy + c0 + e] = (__bf16)0.f;
  }
  red[rg][cq * 4 + 0][0] = s0.x; red[rg][cq * 4 + 0][1] = s1.x;
  red[rg][cq * 4 + 1][0] = s0.y; red[rg][cq * 4 + 1][1] = s1.y;
  red[rg][cq * 4 + 2][0] = s0.z; red[rg][cq * 4 + 2][1] = s1.z;
  red[rg][cq * 4 + 3][0] = s0.w; red[rg][cq * 4 + 3][1] = s1.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 1, w = threadIdx.x & 1;
    float sacc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sacc += red[g][c][w];
    const int col = blockIdx.x * 64 + c;
    if (col < H) part[((long long)blockIdx.y * H + col) * 2 + w] = sacc;
  }
}

// partials [chunks][H][2] of (sum yc, sum yc^2) + m0 -> delta, mean, biased var (+ running statistics: momentum, unbiased var); fp64 combine
__global__ __launch_bounds__(256) void center_finalize_kernel(const float* __restrict__ part, int chunks, int H, double count,
                                                              const float* __restrict__ m0, float* __restrict__ mean, float* __restrict__ var,
                                                              float* __restrict__ delta, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, float momentum) {
  __shared__ double red[8][32][2];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < H)
    for (int k = grp; k < chunks; k += 8) {
      a += (double)part[((long long)k * H + c) * 2 + 0];
      b += (double)part[((long long)k * H + c) * 2 + 1];
    }
  red[grp][cl][0] = a;
  red[grp][cl][1] = b;
  __syncthreads();
  if (grp != 0 || c >= H) return;
  a = 0.0; b = 0.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) { a += red[g][cl][0]; b += red[g][cl][1]; }
  const double d = a / count;
  double v = b / count - d * d;
  if (v < 0.0) v = 0.0;
  const double m = (double)m0[c] + d;
  delta[c] = (float)d;
  mean[c] = (float)m;
  var[c] = (float)v;
  if (run_mean) {
    const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
    run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
    run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
  }
}

// BatchNorm1d backward sums with the BatchNorm INPUT given as bf16 (the centred operand above; `mean` is then its delta):
//   s0 = sum dY, s1 = sum dY * (x - mean) * rstd.   Same grid / reduction order as col_reduce_kernel<2>.
__global__ __launch_bounds__(256) void col_reduce_bwd_xbf_kernel(const __bf16* __restrict__ X, int ldx, const float* __restrict__ dY, int lddy, int M, int H,
                                                                 int rows_per_chunk, const float* __restrict__ mean, const float* __restrict__ var,
                                                                 float eps, float* __restrict__ part, int vec) {
  __shared__ float red[16][64][2];
  const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cq * 4;
  const int valid = H - c0;
  const int rbeg = blockIdx.y * rows_per_chunk;
  const int rend = min(M, rbeg + rows_per_chunk);
  f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
  if (valid > 0) {
    const f32x4 mu = ld4(mean + c0, valid, false), vv = ld4(var + c0, valid, false);
    f32x4 rs;
    rs.x = rsqrtf(vv.x + eps); rs.y = rsqrtf(vv.y + eps); rs.z = rsqrtf(vv.z + eps); rs.w = rsqrtf(vv.w + eps);
    for (int r = rbeg + rg; r < rend; r += 16) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid >= 4) {
        const nbf16x4 q = *reinterpret_cast<const nbf16x4*>(X + (long long)r * ldx + c0);
        x = f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
      } else {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < valid; ++e) v[e] = (float)X[(long long)r * ldx + c0 + e];
        x = f32x4{v[0], v[1], v[2], v[3]};
      }
      const f32x4 dy = ld4(dY + (long long)r * lddy + c0, valid, vec);
      s0 += dy;
      s1 += dy * ((x - mu) * rs);
    }
  }
  red[rg][cq * 4 + 0][0] = s0.x; red[rg][cq * 4 + 0][1] = s1.x;
  red[rg][cq * 4 + 1][0] = s0.y; red[rg][cq * 4 + 1][1] = s1.y;
  red[rg][cq * 4 + 2][0] = s0.z; red[rg][cq * 4 + 2][1] = s1.z;
  red[rg][cq * 4 + 3][0] = s0.w; red[rg][cq * 4 + 3][1] = s1.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 1, w = threadIdx.x & 1;
    float sacc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sacc += red[g][c][w];
    const int col = blockIdx.x * 64 + c;
    if (col < H) part[((long long)blockIdx.y * H + col) * 2 + w] = sacc;
  }
}

// dX = k1 dY - k2 - k3 (x - mu) with the BatchNorm input x as bf16 (H % 4 == 0): the materialised form for recurrences that cannot fuse it
__global__ __launch_bounds__(256) void bn1d_bwd_apply_xbf_kernel(const float* __restrict__ dY, int lddy, const __bf16* __restrict__ X, int ldx,
                                                                 float* __restrict__ dX, int lddx, int M, int H, const float* __restrict__ mean,
                                                                 const float* __restrict__ var, const float* __restrict__ gamma,
                                                                 const float* __restrict__ s0, const float* __restrict__ s1, float eps,
                                                                 float inv_count, int vec) {
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c0 >= H) return;
  const f32x4 mu = ld4(mean + c0, 4, false), vv = ld4(var + c0, 4, false), g = ld4(gamma + c0, 4, false);
  const f32x4 a = ld4(s0 + c0, 4, false), b = ld4(s1 + c0, 4, false);
  f32x4 rs, k1, k2, k3;
  rs.x = rsqrtf(vv.x + eps); rs.y = rsqrtf(vv.y + eps); rs.z = rsqrtf(vv.z + eps); rs.w = rsqrtf(vv.w + eps);
  k1 = g * rs;
  k2 = k1 * a * inv_count;
  k3 = k1 * rs * b * inv_count;
  const int rend = min(M, (int)(blockIdx.y + 1) * BN_ROWS_PER_BLOCK);
#pragma unroll 8
  for (int r = blockIdx.y * BN_ROWS_PER_BLOCK; r < rend; ++r) {
    const f32x4 dy = ld4(dY + (long long)r * lddy + c0, 4, vec);
    const nbf16x4 q = *reinterpret_cast<const nbf16x4*>(X + (long long)r * ldx + c0);
    const f32x4 x = {(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
    st4(dX + (long long)r * lddx + c0, k1 * dy - k2 - k3 * (x - mu), 4, vec);
  }
}

// C[r][c] = C[r][c] * scale[c] + rowv[r] * shift[c]     (R, N) fp32, N % 4 == 0: the weight gradient of a projection whose BatchNorm was
// folded into it — dW_ih = (dGx^T yc) diag(s) + db_ih (x) (beta - delta s)
__global__ __launch_bounds__(256) void scale_rank1_kernel(float* __restrict__ C, int ldc, int R, int N, const float* __restrict__ scale,
                                                          const float* __restrict__ rowv, const float* __restrict__ shift) {
  const int cq = N / 4;
  const long long total = (long long)R * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cq), c0 = (int)(i % cq) * 4;
    f32x4* p = reinterpret_cast<f32x4*>(C + (long long)r * ldc + c0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0), sh = *reinterpret_cast<const f32x4*>(shift + c0);
    *p = *p * sc + sh * rowv[r];
  }
}

int pick_chunks(int M, int colblocks, int min_rows) {
  int chunks = 2048 / (colblocks > 0 ? colblocks : 1);
  if (chunks < 1) chunks = 1;
  const int maxc = ceil_div(M, min_rows);
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  return chunks;
}

}  // namespace

// workspace: chunks * H * 2 floats, chunks <= 2048
extern "C" size_t ds2_colreduce_workspace_bytes(int M, int H) { return (size_t)2048 * 2 * sizeof(float) * (size_t)(H > 64 ? H : 64); }

static int col_reduce_launch(int mode, const float* X, int ldx, const float* X2, int ldx2, float* Y, int ldy, int M, int H,
                             const float* mean, const float* var, float eps, int kind, float* out0, float* out1,
                             float* run_mean, float* run_var, float momentum, void* ws, size_t ws_bytes, hipStream_t s) {
  const int colblocks = ceil_div(H, 64);
  const int chunks = pick_chunks(M, colblocks, 64);
  const int rpc = ceil_div(M, chunks);
  const int nch = ceil_div(M, rpc);
  DS2_REQUIRE(ws && ws_bytes >= (size_t)nch * H * 2 * sizeof(float), "col_reduce: workspace too small");
  int vec = (ldx % 4 == 0) && ((uintptr_t)X % 16 == 0);
  if (X2) vec = vec && (ldx2 % 4 == 0) && ((uintptr_t)X2 % 16 == 0);
  if (Y) vec = vec && (ldy % 4 == 0) && ((uintptr_t)Y % 16 == 0);
  dim3 grid(colblocks, nch), block(256);
  float* part = (float*)ws;
  if (mode == 0) hipLaunchKernelGGL((col_reduce_kernel<0>), grid, block, 0, s, X, ldx, X2, ldx2, Y, ldy, M, H, rpc, mean, var, eps, part, vec);
  else if (mode == 1) hipLaunchKernelGGL((col_reduce_kernel<1>), grid, block, 0, s, X, ldx, X2, ldx2, Y, ldy, M, H, rpc, mean, var, eps, part, vec);
  else hipLaunchKernelGGL((col_reduce_kernel<2>), grid, block, 0, s, X, ldx, X2, ldx2, Y, ldy, M, H, rpc, mean, var, eps, part, vec);
  DS2_LAUNCH_CHECK("col_reduce_kernel");
  hipLaunchKernelGGL(col_finalize_kernel, dim3(ceil_div(H, 32)), dim3(256), 0, s, (const float*)part, nch, H, (double)M, kind,
                     out0, out1, run_mean, run_var, momentum);
  DS2_LAUNCH_CHECK("col_finalize_kernel");
  return 0;
}

// internal (common.h): raw column sums from [chunks][H][2] fp32 partials (fp64 combine) — used by the fused cast + colsum pass
int ds2i_col_finalize_sums(const float* part, int chunks, int H, float* out0, float* out1, hipStream_t s) {
  hipLaunchKernelGGL(col_finalize_kernel, dim3(ceil_div(H, 32)), dim3(256), 0, s, part, chunks, H, 1.0, 1, out0, out1, (float*)nullptr,
                     (float*)nullptr, 0.f);
  DS2_LAUNCH_CHECK("col_finalize_kernel");
  return 0;
}

// mean/biased var per column of X (M,H); optional running-stat update (momentum, unbiased var).
extern "C" int ds2_colstats_f32(const float* X, int ldx, int M, int H, float* mean, float* var, float* run_mean, float* run_var,
                                float momentum, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(X && mean && var && M > 0 && H > 0, "ds2_colstats_f32: bad args");
  return col_reduce_launch(0, X, ldx, nullptr, 0, nullptr, 0, M, H, nullptr, nullptr, 0.f, 0, mean, var, run_mean, run_var,
                           momentum, ws, ws_bytes, (hipStream_t)stream);
}

// Y = Xa + Xb and column stats of Y in one pass (direction sum, blocks.py:92, fused with the next BN's statistics)
extern "C" int ds2_add_colstats_f32(const float* Xa, int lda, const float* Xb, int ldb, float* Y, int ldy, int M, int H,
                                    float* mean, float* var, float* run_mean, float* run_var, float momentum, void* ws,
                                    size_t ws_bytes, void* stream) {
  DS2_REQUIRE(Xa && Xb && Y && mean && var && M > 0 && H > 0, "ds2_add_colstats_f32: bad args");
  return col_reduce_launch(1, Xa, lda, Xb, ldb, Y, ldy, M, H, nullptr, nullptr, 0.f, 0, mean, var, run_mean, run_var, momentum,
                           ws, ws_bytes, (hipStream_t)stream);
}

// column sums: out0[c] = sum_r X[r][c]; out1[c] = sum_r X[r][c]^2
extern "C" int ds2_colsum_f32(const float* X, int ldx, int M, int H, float* sum, float* sumsq, void* ws, size_t ws_bytes,
                              void* stream) {
  DS2_REQUIRE(X && sum && sumsq && M > 0 && H > 0, "ds2_colsum_f32: bad args");
  return col_reduce_launch(0, X, ldx, nullptr, 0, nullptr, 0, M, H, nullptr, nullptr, 0.f, 1, sum, sumsq, nullptr, nullptr, 0.f,
                           ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int ds2_bn1d_apply_f32(const float* X, int ldx, float* Y, int ldy, int M, int H, const float* mean, const float* var,
                                  const float* gamma, const float* beta, float eps, void* stream) {
  DS2_REQUIRE(X && Y && mean && var && gamma && beta, "ds2_bn1d_apply_f32: null pointer");
  const int vec = (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0);
  dim3 grid(ceil_div(ceil_div(H, 4), 256), ceil_div(M, BN_ROWS_PER_BLOCK));
  hipLaunchKernelGGL((bn1d_apply_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, X, ldx, (void*)Y, ldy, M, H, mean, var, gamma, beta,
                     eps, vec);
  DS2_LAUNCH_CHECK("bn1d_apply_kernel");
  return 0;
}

// Y (M, ldy) bf16 = BN(X) with zero pad columns; ldy % 8 == 0, ldy >= H
extern "C" int ds2_bn1d_apply_bf16(const float* X, int ldx, void* Y, int ldy, int M, int H, const float* mean, const float* var,
                                   const float* gamma, const float* beta, float eps, void* stream) {
  DS2_REQUIRE(X && Y && mean && var && gamma && beta, "ds2_bn1d_apply_bf16: null pointer");
  DS2_REQUIRE(ldy >= H && (ldy % 8) == 0, "ds2_bn1d_apply_bf16: bad output pitch %d", ldy);
  const int vec = (ldx % 4 == 0) && ((uintptr_t)X % 16 == 0);
  dim3 grid(ceil_div(ldy / 4, 256), ceil_div(M, BN_ROWS_PER_BLOCK));
  hipLaunchKernelGGL((bn1d_apply_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, X, ldx, Y, ldy, M, H, mean, var, gamma, beta, eps, vec);
  DS2_LAUNCH_CHECK("bn1d_apply_kernel<bf16>");
  return 0;
}

int ds2i_bn1d_bwd_apply(const float* dY, int lddy, const float* X, int ldx, float* dX, int lddx, int M, int H, const float* mean, const float* var,
                        const float* gamma, const float* s0, const float* s1, float eps, hipStream_t stream) {
  const int vec = (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                  ((uintptr_t)dY % 16 == 0) && ((uintptr_t)dX % 16 == 0);
  dim3 grid(ceil_div(ceil_div(H, 4), 256), ceil_div(M, BN_ROWS_PER_BLOCK));
  hipLaunchKernelGGL(bn1d_bwd_apply_kernel, grid, dim3(256), 0, stream, dY, lddy, X, ldx, dX, lddx, M, H, mean, var, gamma, s0, s1, eps,
                     1.0f / (float)M, vec);
  DS2_LAUNCH_CHECK("bn1d_bwd_apply_kernel");
  return 0;
}

// BatchNorm1d backward (training mode): dX, dgamma, dbeta from dY and the forward input X.  dX == NULL: only the column sums (dgamma,
// dbeta) — the caller hands dY, X and the sums to ds2_rnn_bwd_bn, whose K-split recurrence kernel applies the elementwise half on the fly.
extern "C" int ds2_bn1d_bwd_f32(const float* dY, int lddy, const float* X, int ldx, float* dX, int lddx, int M, int H,
                                const float* mean, const float* var, const float* gamma, float eps, float* dgamma, float* dbeta,
                                void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(dY && X && mean && var && gamma && dgamma && dbeta, "ds2_bn1d_bwd_f32: null pointer");
  int rc = col_reduce_launch(2, X, ldx, dY, lddy, nullptr, 0, M, H, mean, var, eps, 1, dbeta, dgamma, nullptr, nullptr, 0.f, ws,
                             ws_bytes, (hipStream_t)stream);
  if (rc || !dX) return rc;
  return ds2i_bn1d_bwd_apply(dY, lddy, X, ldx, dX, lddx, M, H, mean, var, gamma, dbeta, dgamma, eps, (hipStream_t)stream);
}


int ds2i_bn1d_bwd_apply_xbf(const float* dY, int lddy, const void* X, int ldx, float* dX, int lddx, int M, int H, const float* mean, const float* var,
                            const float* gamma, const float* s0, const float* s1, float eps, hipStream_t stream) {
  DS2_REQUIRE((H % 4) == 0 && (ldx % 4) == 0, "bn1d_bwd_apply (bf16 input): H and its pitch must be multiples of 4");
  const int vec = (lddy % 4 == 0) && (lddx % 4 == 0) && ((uintptr_t)dY % 16 == 0) && ((uintptr_t)dX % 16 == 0);
  dim3 grid(ceil_div(ceil_div(H, 4), 256), ceil_div(M, BN_ROWS_PER_BLOCK));
  hipLaunchKernelGGL(bn1d_bwd_apply_xbf_kernel, grid, dim3(256), 0, stream, dY, lddy, (const __bf16*)X, ldx, dX, lddx, M, H, mean, var, gamma, s0, s1, eps,
                     1.0f / (float)M, vec);
  DS2_LAUNCH_CHECK("bn1d_bwd_apply_xbf_kernel");
  return 0;
}

// ---- BatchNorm1d folded into the following projection (round 6): see center_colstats_kernel -------------------------------------------------
// Y = Xa + Xb is never written: Yc = bf16(Y - m0) (M, ldyc: pad columns zero) with m0 the column means from hsum (2, ntiles, H) — the
// per-direction, per-16-row-tile sums of h over time that ds2_rnn_fwd_x emits; mean / var = the batch statistics of Y (what nn.BatchNorm1d
// computes, blocks.py:75, 85-86), delta = mean - m0 (the mean of the centred operand: the "mean" that BatchNorm formulas written on Yc use);
// running statistics updated like ds2_add_colstats_f32.  workspace: ds2_colreduce_workspace_bytes(M, H) + H floats.
extern "C" int ds2_center_colstats(const float* Xa, int lda, const float* Xb, int ldb, const float* hsum, int ntiles, void* Yc_bf16, int ldyc, int M,
                                   int H, float* mean, float* var, float* delta, float* run_mean, float* run_var, float momentum, void* ws,
                                   size_t ws_bytes, void* stream) {
  DS2_REQUIRE(Xa && Xb && hsum && Yc_bf16 && mean && var && delta && M > 0 && H > 0 && ntiles > 0, "ds2_center_colstats: bad args");
  DS2_REQUIRE(ldyc >= H && (ldyc % 8) == 0 && ((uintptr_t)Yc_bf16 % 16) == 0, "ds2_center_colstats: bad operand pitch %d", ldyc);
  const int colblocks = ceil_div(ldyc, 64);
  const int chunks = pick_chunks(M, colblocks, 64);
  const int rpc = ceil_div(M, chunks);
  const int nch = ceil_div(M, rpc);
  DS2_REQUIRE(ws && ws_bytes >= ((size_t)nch * H * 2 + H) * sizeof(float), "ds2_center_colstats: workspace too small");
  const int vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)Xa % 16 == 0) && ((uintptr_t)Xb % 16 == 0);
  float* part = (float*)ws;
  float* m0 = part + (size_t)nch * H * 2;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(center_colstats_kernel, dim3(colblocks, nch), dim3(256), 0, s, Xa, lda, Xb, ldb, hsum, ntiles, (__bf16*)Yc_bf16, ldyc, M, H, rpc,
                     1.0f / (float)M, m0, part, vec);
  DS2_LAUNCH_CHECK("center_colstats_kernel");
  hipLaunchKernelGGL(center_finalize_kernel, dim3(ceil_div(H, 32)), dim3(256), 0, s, (const float*)part, nch, H, (double)M, (const float*)m0, mean, var,
                     delta, run_mean, run_var, momentum);
  DS2_LAUNCH_CHECK("center_finalize_kernel");
  return 0;
}

// ds2_bn1d_bwd_f32 with the BatchNorm input X given as bf16 (pitch ldx): the column sums dbeta = sum dY, dgamma = sum dY xhat and, when dX is
// given (H % 4 == 0), the materialised gradient.  `mean` is the mean OF X (for the centred operand of ds2_center_colstats: its delta).
extern "C" int ds2_bn1d_bwd_xbf16(const float* dY, int lddy, const void* X_bf16, int ldx, float* dX, int lddx, int M, int H, const float* mean,
                                  const float* var, const float* gamma, float eps, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(dY && X_bf16 && mean && var && gamma && dgamma && dbeta && M > 0 && H > 0 && (ldx % 4) == 0, "ds2_bn1d_bwd_xbf16: bad args");
  const int colblocks = ceil_div(H, 64);
  const int chunks = pick_chunks(M, colblocks, 64);
  const int rpc = ceil_div(M, chunks);
  const int nch = ceil_div(M, rpc);
  DS2_REQUIRE(ws && ws_bytes >= (size_t)nch * H * 2 * sizeof(float), "ds2_bn1d_bwd_xbf16: workspace too small");
  const int vec = (lddy % 4 == 0) && ((uintptr_t)dY % 16 == 0);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(col_reduce_bwd_xbf_kernel, dim3(colblocks, nch), dim3(256), 0, s, (const __bf16*)X_bf16, ldx, dY, lddy, M, H, rpc, mean, var, eps,
                     (float*)ws, vec);
  DS2_LAUNCH_CHECK("col_reduce_bwd_xbf_kernel");
  hipLaunchKernelGGL(col_finalize_kernel, dim3(ceil_div(H, 32)), dim3(256), 0, s, (const float*)ws, nch, H, (double)M, 1, dbeta, dgamma, (float*)nullptr,
                     (float*)nullptr, 0.f);
  DS2_LAUNCH_CHECK("col_finalize_kernel");
  if (!dX) return 0;
  return ds2i_bn1d_bwd_apply_xbf(dY, lddy, X_bf16, ldx, dX, lddx, M, H, mean, var, gamma, dbeta, dgamma, eps, s);
}

// C[r][c] = C[r][c] * scale[c] + rowv[r] * shift[c], (R, N) fp32 in place, N % 4 == 0, 16-byte aligned rows
extern "C" int ds2_scale_rank1_f32(float* C, int ldc, int R, int N, const float* scale, const float* rowv, const float* shift, void* stream) {
  DS2_REQUIRE(C && scale && rowv && shift && R > 0 && N > 0 && (N % 4) == 0 && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 &&
              ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0, "ds2_scale_rank1_f32: bad args");
  const long long total = (long long)R * (N / 4);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(scale_rank1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, C, ldc, R, N, scale, rowv, shift);
  DS2_LAUNCH_CHECK("scale_rank1_kernel");
  return 0;
}

extern "C" size_t ds2_chanreduce_workspace_bytes(int C) { return (size_t)2048 * 2 * sizeof(float) * (size_t)(C > 1 ? C : 1); }

static int chan_reduce_launch(int mode, const float* Y, const float* dA, int Bn, int C, int D, int T, const int* lens,
                              const float* mean, const float* var, const float* gamma, const float* beta, float eps, int kind,
                              float* out0, float* out1, float* run_mean, float* run_var, float momentum, void* ws,
                              size_t ws_bytes, hipStream_t s) {
  const int nrows = Bn * D;
  int chunks = 2048 / C;
  if (chunks < 1) chunks = 1;
  if (chunks > nrows) chunks = nrows;
  const int rpc = ceil_div(nrows, chunks);
  const int nch = ceil_div(nrows, rpc);
  DS2_REQUIRE(ws && ws_bytes >= (size_t)nch * C * 2 * sizeof(float), "chan_reduce: workspace too small");
  float* part = (float*)ws;
  dim3 grid(C, nch), block(256);
  if (mode == 0) hipLaunchKernelGGL((chan_reduce_kernel<0>), grid, block, 0, s, Y, dA, Bn, C, D, T, rpc, lens, mean, var, gamma, beta, eps, part);
  else hipLaunchKernelGGL((chan_reduce_kernel<1>), grid, block, 0, s, Y, dA, Bn, C, D, T, rpc, lens, mean, var, gamma, beta, eps, part);
  DS2_LAUNCH_CHECK("chan_reduce_kernel");
  hipLaunchKernelGGL(col_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, s, (const float*)part, nch, C,
                     (double)Bn * D * T, kind, out0, out1, run_mean, run_var, momentum);
  DS2_LAUNCH_CHECK("col_finalize_kernel");
  return 0;
}

// Two ordered stages for MANY partial rows of 32 channels x (sum, sum of squares) (a conv epilogue writes one row per block: thousands):
// stage 0, block j sums rows [j*rpb, (j+1)*rpb) into tmp[j][64] in fp64 (lanes = the 64 consecutive floats of a row: coalesced);
// stage 1, one block sums the tmp rows and finishes mean / biased variance / running statistics.
__global__ __launch_bounds__(256) void chanstats_partials_kernel(const float* __restrict__ part, int n, int rpb, double* __restrict__ tmp, int nblk,
                                                                 double count, float* __restrict__ mean, float* __restrict__ var,
                                                                 float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, int stage) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  double s = 0.0;
  if (stage == 0) {
    const int k0 = blockIdx.x * rpb, k1 = min(n, k0 + rpb);
    int k = k0 + r;
    for (; k + 12 < k1; k += 16) {                          // four loads in flight, added in order
      const float v0 = part[(long long)k * 64 + c], v1 = part[(long long)(k + 4) * 64 + c], v2 = part[(long long)(k + 8) * 64 + c],
                  v3 = part[(long long)(k + 12) * 64 + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; k < k1; k += 4) s += (double)part[(long long)k * 64 + c];
  } else {
    for (int k = r; k < nblk; k += 4) s += tmp[(long long)k * 64 + c];
  }
  red[r][c] = s;
  __syncthreads();
  if (r != 0) return;
  const double t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
  if (stage == 0) {
    tmp[(long long)blockIdx.x * 64 + c] = t;
    return;
  }
  red[0][c] = t;                                            // column c = channel (c >> 1), slot c & 1
  __syncthreads();
  if (c < 32) {
    const double m = red[0][2 * c] / count;
    double v = red[0][2 * c + 1] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
    if (run_mean) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
      run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
    }
  }
}

// mean / biased variance (+ running-stat update) of C channels from [nblk][C][2] (sum, sum of squares) partials written by a conv
// forward epilogue (ds2_conv1_fwd_bf16_stats / ds2_conv2_fwd_bf16_stats); count = B*D*T elements per channel
extern "C" size_t ds2_chanstats_from_partials_workspace_bytes(void) { return (size_t)128 * 64 * sizeof(double); }

extern "C" int ds2_chanstats_from_partials(const float* part, int nblk, int C, double count, float* mean, float* var, float* run_mean,
                                           float* run_var, float momentum, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(part && mean && var && nblk > 0 && C > 0 && count > 0, "ds2_chanstats_from_partials: bad args");
  if (C == 32 && nblk >= 512 && ws && ws_bytes >= ds2_chanstats_from_partials_workspace_bytes()) {
    constexpr int SB = 128;
    const int rpb = ceil_div(nblk, SB), nb = ceil_div(nblk, rpb);
    hipLaunchKernelGGL(chanstats_partials_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, part, nblk, rpb, (double*)ws, nb, count, mean, var,
                       run_mean, run_var, momentum, 0);
    hipLaunchKernelGGL(chanstats_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblk, rpb, (double*)ws, nb, count, mean, var,
                       run_mean, run_var, momentum, 1);
    DS2_LAUNCH_CHECK("chanstats_partials_kernel");
    return 0;
  }
  hipLaunchKernelGGL(col_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, (hipStream_t)stream, part, nblk, C, count, 0, mean, var, run_mean,
                     run_var, momentum);
  DS2_LAUNCH_CHECK("col_finalize_kernel");
  return 0;
}

// per-channel batch statistics of the (already time-masked) conv output Y (B,C,D,T); count = B*D*T
// (padding included).
extern "C" int ds2_bn2d_stats_f32(const float* Y, int B, int C, int D, int T, float* mean, float* var, float* run_mean,
                                  float* run_var, float momentum, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(Y && mean && var, "ds2_bn2d_stats_f32: null pointer");
  return chan_reduce_launch(0, Y, nullptr, B, C, D, T, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, mean, var, run_mean,
                            run_var, momentum, ws, ws_bytes, (hipStream_t)stream);
}

// A = mask(hardtanh_{0,20}(BN(Y)))  (deepspeech.py:62-63 / 65-66 with blocks.py:48-55 masks)
extern "C" int ds2_bn2d_act_fwd_f32(const float* Y, float* A, int B, int C, int D, int T, const int* lens_dev, const float* mean,
                                    const float* var, const float* gamma, const float* beta, float eps, void* stream) {
  DS2_REQUIRE(Y && A && lens_dev && mean && var && gamma && beta, "ds2_bn2d_act_fwd_f32: null pointer");
  const long long total = (long long)B * C * D * ((T + 3) / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn2d_act_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, A, B, C, D, T, lens_dev, mean, var,
                     gamma, beta, eps);
  DS2_LAUNCH_CHECK("bn2d_act_fwd_kernel");
  return 0;
}

// backward of mask∘hardtanh∘mask∘BN∘(mask): dY (masked), dgamma, dbeta from dA and the raw conv output Y
extern "C" int ds2_bn2d_act_bwd_f32(const float* Y, const float* dA, float* dY, int B, int C, int D, int T, const int* lens_dev,
                                    const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                    float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(Y && dA && dY && lens_dev && mean && var && gamma && beta && dgamma && dbeta, "ds2_bn2d_act_bwd_f32: null pointer");
  int rc = chan_reduce_launch(1, Y, dA, B, C, D, T, lens_dev, mean, var, gamma, beta, eps, 1, dbeta, dgamma, nullptr, nullptr,
                              0.f, ws, ws_bytes, (hipStream_t)stream);
  if (rc) return rc;
  const long long total = (long long)B * C * D * ((T + 3) / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn2d_act_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, dA, dY, B, C, D, T, lens_dev,
                     mean, var, gamma, beta, (const float*)dbeta, (const float*)dgamma, eps, 1.0f / ((float)B * D * T));
  DS2_LAUNCH_CHECK("bn2d_act_bwd_apply_kernel");
  return 0;
}

// bf16-mode BatchNorm2d + Hardtanh + mask with the layout casts fused (bn2d_tile_kernel): any of a_f32 (B,32,D,T) fp32,
// a_pad (B,32,D,Tp) bf16 [Tp = ds2_conv_padded_pitch(T)], a_nhwc (B,D,T,32) bf16 may be NULL.  C = 32 channels.
extern "C" int ds2_bn2d_act_fwd_fused(const float* Y, int B, int D, int T, const int* lens_dev, const float* mean, const float* var,
                                      const float* gamma, const float* beta, float eps, float* a_f32, void* a_pad, void* a_nhwc, void* stream) {
  DS2_REQUIRE(Y && lens_dev && mean && var && gamma && beta && (a_f32 || a_pad || a_nhwc), "ds2_bn2d_act_fwd_fused: null pointer");
  const int Tp = (T + 16 + 7) / 8 * 8;                     // = ds2_conv_padded_pitch(T) (conv_bf16.hip)
  hipLaunchKernelGGL((bn2d_tile_kernel<false>), dim3(ceil_div(T, 64), D, B), dim3(256), 0, (hipStream_t)stream, Y, (const float*)nullptr, B, D, T, Tp,
                     lens_dev, mean, var, gamma, beta, (const float*)nullptr, (const float*)nullptr, eps, 0.f, a_f32, (nbf16*)a_pad, (nbf16*)a_nhwc,
                     (float*)nullptr);
  DS2_LAUNCH_CHECK("bn2d_tile_kernel<fwd>");
  return 0;
}

// x (T*B, 32*D) = collapse(mask(hardtanh(BN(Y)))) as fp32 (pitch 32*D; may be NULL) and / or bf16 (pitch ld_bf >= 32*D, ld_bf % 8 == 0, the pad
// columns behind 32*D are written as zeros: the engine rounds the pitch up to the GEMMs' 64-deep k-tile, 1312 -> 1344; may be NULL)
extern "C" int ds2_bn2d_act_collapse(const float* Y, int B, int D, int T, const int* lens_dev, const float* mean, const float* var,
                                     const float* gamma, const float* beta, float eps, float* x_f32, void* x_bf16, int ld_bf, void* stream) {
  DS2_REQUIRE(Y && lens_dev && mean && var && gamma && beta && (x_f32 || x_bf16), "ds2_bn2d_act_collapse: null pointer");
  DS2_REQUIRE(!x_bf16 || (ld_bf >= 32 * D && (ld_bf % 8) == 0), "ds2_bn2d_act_collapse: bad bf16 pitch %d", ld_bf);
  hipLaunchKernelGGL(bn2d_act_collapse_kernel, dim3(ceil_div(T, 32), ceil_div(ld_bf > 32 * D ? ld_bf : 32 * D, 64), B), dim3(256), 0, (hipStream_t)stream, Y, B, D, T, lens_dev, mean, var,
                     gamma, beta, eps, x_f32, (nbf16*)x_bf16, ld_bf);
  DS2_LAUNCH_CHECK("bn2d_act_collapse_kernel");
  return 0;
}

constexpr int CPF_BLOCKS = 128;                            // stage-0 blocks of chan_part_finalize_kernel
extern "C" size_t ds2_bn2d_act_bwd_fused_workspace_bytes(int B, int D, int T) {
  return ds2_chanreduce_workspace_bytes(32) + align_up((size_t)B * D * ceil_div(T, 64) * 32 * sizeof(float), 16) + (size_t)CPF_BLOCKS * 32 * sizeof(double);
}

// Backward of the same block with the layout casts AND the conv bias gradient fused: dgamma, dbeta (32), dbias (32, = per-channel
// sums of dY) and dY in any of the three forms (see ds2_bn2d_act_fwd_fused).
extern "C" int ds2_bn2d_act_bwd_fused(const float* Y, const float* dA, int B, int D, int T, const int* lens_dev, const float* mean,
                                      const float* var, const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta,
                                      float* dbias, float* dy_f32, void* dy_pad, void* dy_nhwc, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(Y && dA && lens_dev && mean && var && gamma && beta && dgamma && dbeta && dbias, "ds2_bn2d_act_bwd_fused: null pointer");
  DS2_REQUIRE(ws && ws_bytes >= ds2_bn2d_act_bwd_fused_workspace_bytes(B, D, T), "ds2_bn2d_act_bwd_fused: workspace too small");
  const size_t red_bytes = ds2_chanreduce_workspace_bytes(32);
  int rc = chan_reduce_launch(1, Y, dA, B, 32, D, T, lens_dev, mean, var, gamma, beta, eps, 1, dbeta, dgamma, nullptr, nullptr, 0.f, ws,
                              red_bytes, (hipStream_t)stream);
  if (rc) return rc;
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + red_bytes);
  const int Tp = (T + 16 + 7) / 8 * 8;
  const dim3 grid(ceil_div(T, 64), D, B);
  hipLaunchKernelGGL((bn2d_tile_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, Y, dA, B, D, T, Tp, lens_dev, mean, var, gamma, beta,
                     (const float*)dbeta, (const float*)dgamma, eps, 1.0f / ((float)B * D * T), dy_f32, (nbf16*)dy_pad, (nbf16*)dy_nhwc, part);
  DS2_LAUNCH_CHECK("bn2d_tile_kernel<bwd>");
  const long long nparts = (long long)grid.x * grid.y * grid.z;
  double* tmp = reinterpret_cast<double*>(reinterpret_cast<char*>(part) + align_up((size_t)nparts * 32 * sizeof(float), 16));
  const long long rpb = (nparts + CPF_BLOCKS - 1) / CPF_BLOCKS;
  const int nblk = (int)((nparts + rpb - 1) / rpb);
  hipLaunchKernelGGL(chan_part_finalize_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)part, nparts, rpb, tmp, nblk, dbias, 0);
  hipLaunchKernelGGL(chan_part_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part, nparts, rpb, tmp, nblk, dbias, 1);
  DS2_LAUNCH_CHECK("chan_part_finalize_kernel");
  return 0;
}

// dir 0: (B,F,T) -> (T,B,F)  (deepspeech.py:135-137 collapse) ; dir 1: inverse
extern "C" int ds2_transpose_bft_f32(const float* src, float* dst, int B, int F, int T, int dir, void* stream) {
  DS2_REQUIRE(src && dst && B > 0 && F > 0 && T > 0, "ds2_transpose_bft_f32: bad args");
  dim3 grid(ceil_div(T, 64), ceil_div(F, 64), B);
  hipLaunchKernelGGL(transpose_bft_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, F, T, dir);
  DS2_LAUNCH_CHECK("transpose_bft_kernel");
  return 0;
}

extern "C" int ds2_transpose2d_f32(const float* src, int ld_src, long long stride_src, float* dst, int ld_dst, long long stride_dst,
                                   int R, int Cc, int batch, void* stream) {
  DS2_REQUIRE(src && dst && R > 0 && Cc > 0 && batch > 0, "ds2_transpose2d_f32: bad args");
  dim3 grid(ceil_div(Cc, 32), ceil_div(R, 32), batch);
  hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, stride_src, dst, ld_dst, stride_dst, R, Cc);
  DS2_LAUNCH_CHECK("transpose2d_kernel");
  return 0;
}
