// Bidirectional GRU / LSTM recurrence (fp32), padded + masked formulation of
// pack_padded_sequence -> aten::gru/lstm -> pad_packed_sequence (blocks.py:87-89) and its backward.
//
// The input projections X W_ih^T + b_ih for all t and both directions are one big MFMA GEMM
// (gemm.hip); this file is the strictly sequential part.  One launch per time step and BOTH
// directions per launch (dir 0 walks t = s, dir 1 walks t = T-1-s): a kernel boundary is the
// cheapest all-to-all seam on MI355X (≈1.5 us, vs 4-7 us for an in-kernel grid barrier), and the
// W_hh slices a block re-reads every step stay resident in its XCD's L2.
//
// Step kernel = [h_{t-1} (BT x H) @ W_hh^T slice] on the f32 matrix cores (v_mfma_f32_16x16x4_f32,
// operands loaded straight global/L2 -> VGPR as K-contiguous 16-byte fragments, K split over the
// block's 4 waves, partial tiles reduced through LDS) fused with the gate non-linearities, the
// per-sample length mask and the state write-back.  A block owns 16 hidden units x all gates x
// BT batch rows of one direction.
//
// Saved for backward (in place of the x-projections): activated gates; aux = W_hn h + b_hn (GRU)
// or the cell state c (LSTM); h per direction.  Rows t >= len[b] hold zeros everywhere, which is
// what makes the reverse direction start at each sample's own last frame.
#include "common.h"

namespace {

struct RnnArgs {
  float* gx;          // (T,B,2,G*H)  fwd: in x-proj / out gates ; bwd: in gates / out d(pre-activations wrt x-proj)
  float* aux;         // (T,B,2,H)    GRU fwd: out hn ; GRU bwd: in hn / out d(hn) ; LSTM: cell state (read-only in bwd)
  float* hbuf;        // (T,B,2,H)    h per direction (fwd: out, bwd: in)
  const float* w;     // fwd: W_hh (2, G*H, H) ; bwd: W_hh^T (2, H, G*H)
  const float* bhh;   // (2, G*H) (fwd only)
  const float* dy;    // (T,B,H) grad wrt y = h_fwd + h_bwd (bwd only), row pitch lddy
  float* dcar;        // (2 parity, 2 dir, B, H) bwd carry: GRU dh*z ; LSTM dc*f
  const int* lens;    // (B) valid output frames per sample
  int T, B, H, lddy;
};

__device__ __forceinline__ f32x4 ldfrag(const float* __restrict__ p, int valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
  }
  return v;
}

// D[mb][nb] (16x16 tiles) += A(16*MB rows x K) * B(16*NB rows x K)^T, both K-contiguous.
// aptr(mb, k) / bptr(nb, k) return this lane's row pointer at column k, or nullptr for a zero row.
// K is consumed in chunks of 16 (4 k-quads x float4), chunk c handled by wave (c & 3).
template <int MB, int NB, class AP, class BP>
__device__ __forceinline__ void mfma_rows_kcont(f32x4 (&acc)[MB][NB], int K, int wave, int kq, AP aptr, BP bptr) {
  const int nch = (K + 15) / 16;
  f32x4 a0[MB], b0[NB], a1[MB], b1[NB];
  auto load = [&](f32x4(&a)[MB], f32x4(&b)[NB], int c) {
    const int k = c * 16 + 4 * kq;
    const int valid = (c < nch) ? (K - k) : 0;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const float* p = aptr(i, k);
      a[i] = ldfrag(p, p ? valid : 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* p = bptr(j, k);
      b[j] = ldfrag(p, p ? valid : 0);
    }
  };
  auto mul = [&](const f32x4(&a)[MB], const f32x4(&b)[NB]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };
  int c = wave;
  load(a0, b0, c);
  load(a1, b1, c + 4);
  for (; c < nch; c += 8) {
    f32x4 ta[MB], tb[NB], ua[MB], ub[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) { ta[i] = a0[i]; ua[i] = a1[i]; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { tb[j] = b0[j]; ub[j] = b1[j]; }
    load(a0, b0, c + 8);
    load(a1, b1, c + 12);
    mul(ta, tb);
    mul(ua, ub);   // chunk c+4 (zeros if beyond K)
  }
}

// ------------------------------------------------------------------------------------------
// forward step
// ------------------------------------------------------------------------------------------
template <int G, int MB>
__global__ __launch_bounds__(256) void rnn_fwd_step_kernel(RnnArgs a, int s, int nbt) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4][MB * G][64];
  const int dir = blockIdx.y;
  const int slice = blockIdx.x / nbt, bt = blockIdx.x % nbt;
  const int j0 = slice * 16, b0 = bt * (16 * MB);
  const int T = a.T, B = a.B, H = a.H;
  const int t = dir == 0 ? s : T - 1 - s;
  const int tp = dir == 0 ? t - 1 : t + 1;
  const bool has_prev = s > 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;

  f32x4 acc[MB][G];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (has_prev) {
    const float* hprev = a.hbuf + ((long long)tp * B * 2 + dir) * H;   // + b*2H
    const float* w = a.w + (long long)dir * G * H * H;
    auto aptr = [&](int mb, int k) -> const float* {
      const int b = b0 + mb * 16 + r16;
      return (b < B) ? hprev + (long long)b * 2 * H + k : nullptr;
    };
    auto bptr = [&](int g, int k) -> const float* {
      const int j = j0 + r16;
      return (j < H) ? w + ((long long)g * H + j) * H + k : nullptr;
    };
    mfma_rows_kcont<MB, G>(acc, H, wave, kq, aptr, bptr);
  }
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) red[wave][i * G + g][lane] = acc[i][g];
  __syncthreads();

  // epilogue: MB*256 (b, j) pairs, consecutive threads -> consecutive j
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int q = threadIdx.x;
    const int jl = q & 15, brow = q >> 4;
    const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
    const int b = b0 + i * 16 + brow, j = j0 + jl;
    if (b >= B || j >= H) continue;
    float gh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int tile = i * G + g;
      gh[g] = red[0][tile][src_lane][reg] + red[1][tile][src_lane][reg] + red[2][tile][src_lane][reg] +
              red[3][tile][src_lane][reg] + a.bhh[(dir * G + g) * H + j];
    }
    const bool valid = t < a.lens[b];
    const long long row = ((long long)t * B + b) * 2 + dir;
    float* gx = a.gx + row * G * H + j;
    float* ho = a.hbuf + row * H + j;
    float* ax = a.aux + row * H + j;
    if (!valid) {
#pragma unroll
      for (int g = 0; g < G; ++g) gx[g * H] = 0.f;
      *ho = 0.f;
      *ax = 0.f;
      continue;
    }
    const long long prow = ((long long)tp * B + b) * 2 + dir;
    if constexpr (G == 3) {
      const float hp = has_prev ? a.hbuf[prow * H + j] : 0.f;
      const float r = sigmoidf_(gx[0] + gh[0]);
      const float z = sigmoidf_(gx[H] + gh[1]);
      const float n = tanhf(gx[2 * H] + r * gh[2]);
      gx[0] = r; gx[H] = z; gx[2 * H] = n;
      *ax = gh[2];
      *ho = (1.f - z) * n + z * hp;
    } else {
      const float cp = has_prev ? a.aux[prow * H + j] : 0.f;
      const float ig = sigmoidf_(gx[0] + gh[0]);
      const float fg = sigmoidf_(gx[H] + gh[1]);
      const float gg = tanhf(gx[2 * H] + gh[2]);
      const float og = sigmoidf_(gx[3 * H] + gh[G - 1]);
      const float c = fg * cp + ig * gg;
      gx[0] = ig; gx[H] = fg; gx[2 * H] = gg; gx[3 * H] = og;
      *ax = c;
      *ho = og * tanhf(c);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward step
// ------------------------------------------------------------------------------------------
template <int G, int MB>
__global__ __launch_bounds__(256) void rnn_bwd_step_kernel(RnnArgs a, int s, int nbt) {
  __shared__ __attribute__((aligned(16))) f32x4 red[4][MB][64];
  const int dir = blockIdx.y;
  const int slice = blockIdx.x / nbt, bt = blockIdx.x % nbt;
  const int j0 = slice * 16, b0 = bt * (16 * MB);
  const int T = a.T, B = a.B, H = a.H;
  const int t = dir == 0 ? T - 1 - s : s;          // reverse of the forward order
  const int tq = dir == 0 ? t + 1 : t - 1;         // step processed just before (its d-gates feed our carry)
  const int tpf = dir == 0 ? t - 1 : t + 1;        // previous step in FORWARD order (h_{prev}, c_{prev})
  const bool has_q = s > 0;
  const bool has_pf = dir == 0 ? (t > 0) : (t < T - 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;

  f32x4 acc[MB][1];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (has_q) {
    const float* dg = a.gx + ((long long)tq * B * 2 + dir) * G * H;     // + b*2*G*H
    const float* dn = a.aux + ((long long)tq * B * 2 + dir) * H;        // + b*2*H   (GRU d(hn))
    const float* wt = a.w + (long long)dir * H * G * H;
    auto aptr = [&](int mb, int k) -> const float* {
      const int b = b0 + mb * 16 + r16;
      if (b >= B) return nullptr;
      if (G == 3 && k >= 2 * H) return dn + (long long)b * 2 * H + (k - 2 * H);
      return dg + (long long)b * 2 * G * H + k;
    };
    auto bptr = [&](int, int k) -> const float* {
      const int j = j0 + r16;
      return (j < H) ? wt + (long long)j * G * H + k : nullptr;
    };
    mfma_rows_kcont<MB, 1>(acc, G * H, wave, kq, aptr, bptr);
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) red[wave][i][lane] = acc[i][0];
  __syncthreads();

  const float* dcar_in = a.dcar + ((long long)(((s + 1) & 1) * 2 + dir)) * B * H;
  float* dcar_out = a.dcar + ((long long)((s & 1) * 2 + dir)) * B * H;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int q = threadIdx.x;
    const int jl = q & 15, brow = q >> 4;
    const int src_lane = (brow >> 2) * 16 + jl, reg = brow & 3;
    const int b = b0 + i * 16 + brow, j = j0 + jl;
    if (b >= B || j >= H) continue;
    const float carry = red[0][i][src_lane][reg] + red[1][i][src_lane][reg] + red[2][i][src_lane][reg] + red[3][i][src_lane][reg];
    const bool valid = t < a.lens[b];
    const long long row = ((long long)t * B + b) * 2 + dir;
    float* gx = a.gx + row * G * H + j;
    float* ax = a.aux + row * H + j;
    float* dco = dcar_out + (long long)b * H + j;
    if (!valid) {
#pragma unroll
      for (int g = 0; g < G; ++g) gx[g * H] = 0.f;
      if (G == 3) *ax = 0.f;
      *dco = 0.f;
      continue;
    }
    const float dci = has_q ? dcar_in[(long long)b * H + j] : 0.f;
    const float dyv = a.dy[((long long)t * B + b) * a.lddy + j];
    const long long prow = ((long long)tpf * B + b) * 2 + dir;
    if constexpr (G == 3) {
      const float dh = dyv + carry + dci;
      const float hp = has_pf ? a.hbuf[prow * H + j] : 0.f;
      const float r = gx[0], z = gx[H], n = gx[2 * H], hn = *ax;
      const float dn = dh * (1.f - z);
      const float dz = dh * (hp - n);
      const float dpn = dn * (1.f - n * n);
      const float dr = dpn * hn;
      gx[0] = dr * r * (1.f - r);
      gx[H] = dz * z * (1.f - z);
      gx[2 * H] = dpn;
      *ax = dpn * r;
      *dco = dh * z;
    } else {
      const float dh = dyv + carry;
      const float ig = gx[0], fg = gx[H], gg = gx[2 * H], og = gx[3 * H];
      const float c = *ax;
      const float cp = has_pf ? a.aux[prow * H + j] : 0.f;
      const float tc = tanhf(c);
      const float dc = dci + dh * og * (1.f - tc * tc);
      gx[0] = dc * gg * ig * (1.f - ig);
      gx[H] = dc * cp * fg * (1.f - fg);
      gx[2 * H] = dc * ig * (1.f - gg * gg);
      gx[3 * H] = dh * tc * og * (1.f - og);
      *dco = dc * fg;
    }
  }
}

template <int G>
int launch_steps(bool bwd, RnnArgs a, hipStream_t st) {
  const int nsl = ceil_div(a.H, 16);
  // 32-row batch tiles halve the W_hh re-reads; use them when that still fills the chip
  int mb = (a.B > 16 && (long long)nsl * ceil_div(a.B, 32) * 2 >= 200) ? 2 : 1;
  const int nbt = ceil_div(a.B, 16 * mb);
  dim3 grid(nsl * nbt, 2), block(256);
  for (int s = 0; s < a.T; ++s) {
    if (!bwd) {
      if (mb == 2) hipLaunchKernelGGL((rnn_fwd_step_kernel<G, 2>), grid, block, 0, st, a, s, nbt);
      else hipLaunchKernelGGL((rnn_fwd_step_kernel<G, 1>), grid, block, 0, st, a, s, nbt);
    } else {
      if (mb == 2) hipLaunchKernelGGL((rnn_bwd_step_kernel<G, 2>), grid, block, 0, st, a, s, nbt);
      else hipLaunchKernelGGL((rnn_bwd_step_kernel<G, 1>), grid, block, 0, st, a, s, nbt);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ds2_set_error("rnn step launch failed: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

// gates: 3 = GRU (r,z,n), 4 = LSTM (i,f,g,o).
//   gx   (T,B,2,G*H)  in: X W_ih^T + b_ih for [fwd | reverse] ; out: activated gates (saved for backward)
//   whh  (2,G*H,H), bhh (2,G*H)     [weight_hh_l0, weight_hh_l0_reverse], [bias_hh_l0, bias_hh_l0_reverse]
//   hbuf (T,B,2,H) out: h per direction (0 beyond each sample's length)
//   aux  (T,B,2,H) out: GRU W_hn h + b_hn ; LSTM cell state
extern "C" int ds2_rnn_fwd_f32(int gates, float* gx, const float* whh, const float* bhh, float* hbuf, float* aux,
                               const int* lens_dev, int T, int B, int H, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_fwd_f32: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE(gx && whh && bhh && hbuf && aux && lens_dev, "ds2_rnn_fwd_f32: null pointer");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_fwd_f32: need H %% 4 == 0 (H=%d)", H);
  RnnArgs a{};
  a.gx = gx; a.aux = aux; a.hbuf = hbuf; a.w = whh; a.bhh = bhh; a.lens = lens_dev; a.T = T; a.B = B; a.H = H;
  return gates == 3 ? launch_steps<3>(false, a, (hipStream_t)stream) : launch_steps<4>(false, a, (hipStream_t)stream);
}

extern "C" size_t ds2_rnn_bwd_workspace_bytes(int B, int H) { return (size_t)4 * B * H * sizeof(float); }

//   dy    (T,B,H) pitch lddy: grad wrt y = h_fwd + h_bwd
//   gx    in: gates from fwd ; out: grad wrt the x-projections (T,B,2,G*H)  (= dGx, feeds dW_ih, db_ih, dX)
//   aux   GRU: in hn, out d(hn) [so that dGh = (dGx_r, dGx_z, aux)] ; LSTM: cell state (unchanged; dGh = dGx)
//   whhT  (2,H,G*H): per-direction transpose of W_hh
extern "C" int ds2_rnn_bwd_f32(int gates, const float* dy, int lddy, float* gx, float* aux, const float* hbuf, const float* whhT,
                               const int* lens_dev, int T, int B, int H, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(gates == 3 || gates == 4, "ds2_rnn_bwd_f32: gates must be 3 (GRU) or 4 (LSTM)");
  DS2_REQUIRE(dy && gx && aux && hbuf && whhT && lens_dev, "ds2_rnn_bwd_f32: null pointer");
  DS2_REQUIRE(T > 0 && B > 0 && H > 0 && (H % 4) == 0, "ds2_rnn_bwd_f32: need H %% 4 == 0 (H=%d)", H);
  DS2_REQUIRE(ws && ws_bytes >= ds2_rnn_bwd_workspace_bytes(B, H), "ds2_rnn_bwd_f32: workspace too small");
  RnnArgs a{};
  a.gx = gx; a.aux = aux; a.hbuf = const_cast<float*>(hbuf); a.w = whhT; a.dy = dy; a.lddy = lddy; a.dcar = (float*)ws;
  a.lens = lens_dev; a.T = T; a.B = B; a.H = H;
  return gates == 3 ? launch_steps<3>(true, a, (hipStream_t)stream) : launch_steps<4>(true, a, (hipStream_t)stream);
}
