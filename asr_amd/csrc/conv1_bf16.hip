// conv1 forward and weight gradient with bf16 MFMA operands (precision = "bf16"; conv.hip's fp32 kernels stay the parity path).
// Conv2d(1, 32, (41, 11), stride (2, 2), padding (20, 5)), deepspeech.py:61:
//     y1[b,co,o,t] = bias[co] + sum_{kd<41, kt<11} W[co,0,kd,kt] * x[b, 2o + kd - 20, 2t + kt - 5]
//
// One input channel gives the MFMA no channel axis to contract over, so the 11 time taps play that role (padded to 16, the five extra
// taps have zero weights).  Operand images, built from one read of the spectrogram:
//     XB  [b][f][P]       bf16 ROWS: XB[..][7 + s] = x[b][f][s], zeros in front and behind (P = 2 pad32(T) + 72)      -> forward operand
//     X16T[b][f][16][Tp]  (time-contiguous: X16T[..][c][t] = x[b][f][2t + c - 5], c < 11, else 0)                      -> wgrad operand
// Forward (round 5): the 16 taps of output step t are the 16 CONSECUTIVE samples 2t - 5 .. 2t + 10 of a row, so the MFMA fragment of lane t
// is a 16-byte run of the raw row at byte offset 4t (+ 16 for the upper half-wave) of a window that starts at sample 2 t0 - 5 = XB index
// 2 t0 + 2: 4-byte aligned — two ds_read2_b32 per fragment.  Rounds 2-4 materialised the tap-expanded image X16[b][f][t][16] (165 MB, eight
// times the rows) and re-read it 3.4-fold (55 input rows per 8 output rows): 900 MB of traffic bound the forward kernel at 256 us whatever
// its stores or DMA issue cost (scripts/probe_conv1.hip).
//   forward: GEMM M = co (32), N = t, K = (kd, c) = 41 x 16.  block = (b, 8 output rows, half of the time axis), persistent over
//            32-step tiles: all 41 weight rows are staged once, the 55 input rows a tile touches are DMA'd (global_load_lds, one
//            1 KiB instruction per row) into a double buffer while the previous tile is multiplied; 8 waves, one output row each (two
//            per SIMD so that one wave's DMA issue, fragment reads and stores hide under the other's MFMAs), two accumulator chains.
//   wgrad:   GEMM M = co, N = (kd, c) = 656, K = t.  block = (b, 64 time steps) loops over the 81 output rows with a 44-slot LDS
//            ring of input rows (two new rows per step, written while the step computes) and a double-buffered dY tile cast from
//            fp32 on the fly; 21 accumulator tiles of 32 columns (= 2 kernel rows x 16 taps) over 4 waves; ordered reduction
//            of the per-block partials afterwards (deterministic).
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_c1 __attribute__((ext_vector_type(4), aligned(4)));     // a dword-aligned dwordx4 access

constexpr int KD = 41, KTAPS = 11, PD = 20, PT = 5, CO = 32, NC = 16;

__device__ __attribute__((aligned(16))) float g_zero_c1[4] = {0.f, 0.f, 0.f, 0.f};

// -DDS2_C1_TRACE (scripts/probe_conv1.hip only): wave 0 of every block sums the shader clocks it spends in the phases of its main loop
#ifdef DS2_C1_TRACE
__device__ unsigned long long g_c1_trace[8192 * 8];
#define C1T_DECL unsigned long long c1t_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c1t_last_ = __builtin_amdgcn_s_memtime()
#define C1T(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); c1t_[k] += n_ - c1t_last_; c1t_last_ = n_; } while (0)
#define C1T_DUMP(blk) do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) g_c1_trace[(long long)(blk) * 8 + k_] = c1t_[k_]; } while (0)
#else
#define C1T_DECL
#define C1T(k)
#define C1T_DUMP(blk)
#endif

// ---- operand images --------------------------------------------------------------------------------------------------
// thread = (b, f, 8 consecutive output steps): 25 input samples -> 8 X16 pixels (256 contiguous bytes) and one 16-byte run in
// each of the 16 X16T rows
__global__ __launch_bounds__(256) void conv1_gather_kernel(const float* __restrict__ x, __bf16* __restrict__ X16, __bf16* __restrict__ X16T,
                                                           long long rows /* B*F */, int Tin, int T, int Tp) {
  const int octs = Tp / 8;
  const long long total = rows * octs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / octs;
    const int t0 = (int)(i % octs) * 8;
    const float* xr = x + row * Tin;
    // 25 consecutive samples from 2 t0 - 5 on: six dword-aligned dwordx4 loads + one (the run starts at an odd sample: 4-byte aligned
    // only; round 2-4: 25 scalar loads per thread), element-wise at the row's ends
    float v[25];
    const int s0 = 2 * t0 - PT;
    if (s0 >= 0 && s0 + 25 <= Tin) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x4_c1 w = *reinterpret_cast<const f32x4_c1*>(xr + s0 + 4 * q);
        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
      }
      v[24] = xr[s0 + 24];
    } else {
#pragma unroll
      for (int j = 0; j < 25; ++j) {
        const int ti = s0 + j;
        v[j] = (ti >= 0 && ti < Tin) ? xr[ti] : 0.f;
      }
    }
    // pixel (t0 + e), tap c  <-  v[2e + c]
    if (X16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (t0 + e < T) {
          bf16x8 lo, hi;
#pragma unroll
          for (int c = 0; c < 8; ++c) lo[c] = (__bf16)v[2 * e + c];
#pragma unroll
          for (int c = 0; c < 8; ++c) hi[c] = (c + 8 < KTAPS) ? (__bf16)v[2 * e + c + 8] : (__bf16)0.f;
          __bf16* p = X16 + (row * T + t0 + e) * NC;
          *reinterpret_cast<bf16x8*>(p) = lo;
          *reinterpret_cast<bf16x8*>(p + 8) = hi;
        }
      }
    }
    if (X16T) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (c < KTAPS && t0 + e < T) ? (__bf16)v[2 * e + c] : (__bf16)0.f;
        *reinterpret_cast<bf16x8*>(X16T + (row * NC + c) * Tp + t0) = o;
      }
    }
  }
}

// XB[b][f][7 + s] = bf16(x[b][f][s]), zeros elsewhere; thread = 8 consecutive elements of a row (16-byte stores; the source run starts one
// float before an 8-element boundary, so it is read by scalar loads)
constexpr int XB_LEFT = 7;
inline __host__ __device__ int xb_pitch(int T) { return 2 * ((T + 31) / 32 * 32) + 72; }
__global__ __launch_bounds__(256) void conv1_rows_kernel(const float* __restrict__ x, __bf16* __restrict__ XB, long long rows, int Tin, int P) {
  const int oct = P / 8;
  const long long total = rows * oct;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / oct;
    const int k0 = (int)(i % oct) * 8;
    const float* xr = x + row * Tin;
    bf16x8 o;
    const int sb = k0 - XB_LEFT;
    if (sb >= 0 && sb + 8 <= Tin) {                         // two dword-aligned dwordx4 loads
      const f32x4_c1 w0 = *reinterpret_cast<const f32x4_c1*>(xr + sb), w1 = *reinterpret_cast<const f32x4_c1*>(xr + sb + 4);
      o = bf16x8{(__bf16)w0.x, (__bf16)w0.y, (__bf16)w0.z, (__bf16)w0.w, (__bf16)w1.x, (__bf16)w1.y, (__bf16)w1.z, (__bf16)w1.w};
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int sidx = sb + j;
        o[j] = (sidx >= 0 && sidx < Tin) ? (__bf16)xr[sidx] : (__bf16)0.f;
      }
    }
    *reinterpret_cast<bf16x8*>(XB + row * P + k0) = o;
  }
}

// 32-byte records (one pixel / one output channel = 16 taps) read by 16-lane ds_read_b128 groups at record stride would hit
// every bank row twice; swapping the two 16-byte halves of records 8..15 (mod 16) makes the 16 lanes of a group land on 16
// distinct bank quads.  Applied to the packed weights here and, through the DMA source addresses, to the staged input rows.
__device__ __forceinline__ int swz_half(int rec, int half) { return half ^ ((rec >> 3) & 1); }

// W1 (32,1,41,11) fp32 -> Wp[kd][co][2 halves (swizzled)][8] bf16 (taps 11..15 zero)
__global__ void conv1_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KD * CO * NC) return;
  const int c = i % NC, co = (i / NC) % CO, kd = i / (NC * CO);
  const float v = (c < KTAPS) ? w[(co * KD + kd) * KTAPS + c] : 0.f;
  wp[(kd * CO + co) * NC + swz_half(co, c >> 3) * 8 + (c & 7)] = (__bf16)v;
}

// ---- forward -----------------------------------------------------------------------------------------------------------
constexpr int F_OG = 8, F_TT = 32, F_SPLIT = 2;
constexpr int F_ROWS = 2 * (F_OG - 1) + KD;        // 55 input rows per block
constexpr int F_ROWB = 256;                        // bytes per staged row: 128 samples from 2 t0 - 5 on (78 are read); four rows per global_load_lds wave-instruction
constexpr int F_BUF = (F_ROWS + 3) / 4 * 1024;         // 14336: the row groups of one window
constexpr int F_WB = KD * CO * NC * 2;             // 41984 bytes of weights
constexpr int F_LDS = 2 * F_BUF + F_WB;            // 70656: two row windows (double buffer) + all weights -> two blocks per CU

struct C1Args {
  const __bf16* XB; const __bf16* wp; const float* bias; const int* lens; float* y;
  int B, F, T, D1, P;
};

typedef __attribute__((address_space(3))) void c1_lds_void;
typedef const __attribute__((address_space(1))) void c1_gbl_void;

// block = (half of the time axis, 8 output rows, b), persistent over its time tiles: the weights are staged once, the 55-row
// input window of tile i+2 is DMA'd (one 1 KiB global_load_lds per row) while tiles i+1 / i are multiplied / stored.
// 8 waves, one output row each (two per SIMD, so that one wave's DMA issue, fragment reads and stores hide under the other's MFMAs).
__global__ __launch_bounds__(512) void conv1_bf16_fwd_kernel(C1Args a, float* __restrict__ stat_part) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* wl = lds + 2 * F_BUF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int o0 = blockIdx.y * F_OG, b = blockIdx.z;
  const int len = a.lens ? min(a.lens[b], a.T) : a.T;
  const int ntile = (a.T + F_TT - 1) / F_TT, per = (ntile + F_SPLIT - 1) / F_SPLIT;
  const int tile_beg = blockIdx.x * per, tile_end = min(ntile, tile_beg + per);
  const int f0 = 2 * o0 - PD;
  const int o = o0 + wave;

  // DMA of one tile's window: one wave-instruction lands FOUR rows (lane = row (lane >> 4), 16-byte chunk (lane & 15) of the 256 staged
  // bytes); wave w moves the row groups w, w + 8.  Rows outside the spectrogram read the zero page.
  auto stage = [&](int buf, int t0) {
    for (int gq = wave; gq < (F_ROWS + 3) / 4; gq += 8) {
      const int r = gq * 4 + (lane >> 4), f = f0 + r;
      const bool ok = r < F_ROWS && f >= 0 && f < a.F;
      const void* src = ok ? (const void*)(a.XB + ((long long)b * a.F + f) * a.P + 2 * t0 + 2 + (lane & 15) * 8) : (const void*)g_zero_c1;
      __builtin_amdgcn_global_load_lds((c1_gbl_void*)src, (c1_lds_void*)(lds + buf * F_BUF + gq * 1024), 16, 0, 0);
    }
  };
  const bool any = tile_beg < tile_end && tile_beg * F_TT < len;        // block-uniform
  if (any) {
    for (int c = tid; c < F_WB / 16; c += 512) *reinterpret_cast<u32x4*>(wl + c * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.wp) + c * 16);
    stage(0, tile_beg * F_TT);
    if (tile_beg + 1 < tile_end && (tile_beg + 1) * F_TT < len) stage(1, (tile_beg + 1) * F_TT);
  }
  __syncthreads();                                                        // vmcnt(0) + barrier: weights and the first two windows are in LDS
  const int rd = swz_half(l31, half) * 16;                                // this lane's 16-byte slot inside its 32-byte record
  // The product is formed TRANSPOSED (input fragment as the MFMA's first operand): a lane then holds ONE output channel (lane & 31) at 4 x 4
  // consecutive frames, so a tile leaves as four 16-byte stores per lane instead of sixteen 4-byte ones (the rows of y1 are only 4-byte
  // aligned: dword-aligned dwordx4, see norm.hip) and bias / BatchNorm statistics are one register each.
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const float bv = a.bias ? a.bias[l31] : 0.f;             // loaded ONCE before any store is in flight (one in-order memory counter)
  float s1 = 0.f, s2 = 0.f;                                // BatchNorm statistics of what this lane stores (stat_part != NULL)
  C1T_DECL;
  C1T(0);                                                    // prologue
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int t0 = tile * F_TT, buf = (tile - tile_beg) & 1;
    const bool live = t0 < len;                                           // block-uniform; later tiles of a short utterance are all zero
    // two independent accumulator chains (even / odd kernel rows): back-to-back MFMAs never wait on their own result
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (live && o < a.D1) {
      // output row o0 + wave reads local input rows 2*wave + kd, kd = 0..40
      // this lane's 16 taps of output step t0 + l31 = 32 consecutive bytes of the raw row from byte 4 l31 on; the upper half-wave takes taps 8..15
      typedef bf16x8 bf16x8u __attribute__((aligned(4)));
      const char* bp = lds + buf * F_BUF + (2 * wave) * F_ROWB + l31 * 4 + half * 16;
      const char* ap = wl + l31 * (NC * 2) + rd;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ap + kd * (CO * NC * 2));
        const bf16x8 xb = *reinterpret_cast<const bf16x8u*>(bp + kd * F_ROWB);
        if (kd & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, wf, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, wf, acc0, 0, 0, 0);
      }
    }
    // The wait must be EXPLICIT: the compiler does not put a vmcnt wait in front of this loop's barrier (it emits a bare s_barrier:
    // the LDS-DMA issued one iteration ago is not something its fence lowering tracks across the back edge), and without it
    // the next tile's ds_reads could overtake a still-landing window row — measured as run-to-run differences in ~1 % of forward
    // passes (scripts/det_check_fwd.py).  vmcnt(0) also covers the previous tile's stores, issued a whole tile ago (they have retired by
    // then: 90 clocks per block in this wait, scripts/probe_conv1.hip).
    C1T(1);                                                  // MFMAs issued (incl. their fragment reads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C1T(2);                                                  // DMA of the other window + this wave's earlier stores retired
    __syncthreads();            // everyone is done reading this window; the other window (DMA'd a whole tile ago) has landed
    C1T(3);                                                  // barrier
    if (tile + 2 < tile_end && (tile + 2) * F_TT < len) stage(buf, (tile + 2) * F_TT);     // refill it two tiles ahead
    C1T(4);                                                  // DMA issue
    // the stores of this tile drain under the next tile's MFMAs.  (Issuing them INSIDE the next tile's MFMA loop — one behind every second
    // MFMA, counted wait — was built and is slower: 250 against 222 us.)
    if (o < a.D1) {
      float* yrow = a.y + (((long long)b * CO + l31) * a.D1 + o) * a.T;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int t = t0 + 8 * g4 + 4 * half;              // frames t .. t + 3 = accumulator registers 4 g4 .. 4 g4 + 3
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (t + j < len) ? acc0[4 * g4 + j] + acc1[4 * g4 + j] + bv : 0.f;
          if (t + j < a.T) { s1 += v[j]; s2 += v[j] * v[j]; }
        }
        if (t + 3 < a.T) *reinterpret_cast<f32x4u*>(yrow + t) = f32x4u{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (t + j < a.T) yrow[t + j] = v[j];
        }
      }
    }
    C1T(5);                                                  // stores issued
  }
  C1T_DUMP(((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  if (stat_part) {
    // per-channel sums of the block: over the 32 lanes of a half-wave, then over the 8 waves (output rows) in order; the window
    // buffers are dead here (every wave has passed the last tile's barrier)
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);            // [8 waves][32 co][2]
    s1 += __shfl_xor(s1, 32);                              // the two half-waves hold the other frames of the same channel
    s2 += __shfl_xor(s2, 32);
    if (half == 0) { red[(wave * 32 + l31) * 2 + 0] = s1; red[(wave * 32 + l31) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 64) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += red[w * 64 + tid];
      const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      stat_part[blk * 64 + tid] = sum;                     // [blk][co][2]
    }
  }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
constexpr int W_TT = 64;                       // time steps (K) per block
constexpr int W_PF = 4;                        // output rows in flight between their HBM request and their LDS publish
constexpr int W_PITCH = W_TT * 2 + 16;         // 144 bytes per (row, tap) / per co: conflict-free ds_read_b128
constexpr int W_SLOT = NC * W_PITCH;           // 2304 bytes per input row
constexpr int W_NR = 44;                       // ring slots (window 41 + 2 being written + 1)
constexpr int W_DY = CO * W_PITCH;             // 4608 bytes per dY tile
constexpr int W_LDS = (W_NR + 1) * W_SLOT + 2 * W_DY;   // ring + zero slot + 2 dY buffers = 112896
constexpr int W_TILES = (KD + 1) / 2;          // 21 column tiles of 32 = 2 kernel rows x 16 taps
constexpr int W_TPW = (W_TILES + 3) / 4;       // 6 tiles per wave at most

struct C1WArgs {
  const __bf16* X16T; const float* dy; const int* lens; float* part;
  int B, F, T, Tp, D1;
};

__global__ __launch_bounds__(256) void conv1_bf16_wgrad_kernel(C1WArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ring = lds;
  char* zslot = lds + W_NR * W_SLOT;
  char* dyl = zslot + W_SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int t0 = blockIdx.x * W_TT, b = blockIdx.y;
  const int len = a.lens ? min(a.lens[b], a.T) : a.T;
  float* out = a.part + ((long long)b * gridDim.x + blockIdx.x) * (CO * KD * KTAPS);

  f32x16 acc[W_TPW];
#pragma unroll
  for (int i = 0; i < W_TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (t0 < len) {                                  // dY is zero beyond the utterance: nothing to add from this tile otherwise
    // one input row = 16 taps x 64 steps = 128 chunks of 16 bytes; thread tid < 128 owns chunk (c = tid >> 3, q = tid & 7)
    auto row_src = [&](int f, int c, int q) -> const void* {
      const bool ok = f >= 0 && f < a.F;
      return ok ? (const void*)(a.X16T + (((long long)b * a.F + f) * NC + c) * a.Tp + t0 + q * 8) : (const void*)g_zero_c1;
    };
    // four scalar loads, no control flow (rows of dY are 4-byte aligned only; out-of-range steps read the zero page): a static
    // number of loads per step lets the compiler wait for the OLDER register set with a counted vmcnt instead of vmcnt(0)
    auto dy_load = [&](int o, int k, f32x4& v) {    // chunk k (0..511): co = k >> 4, 4 steps at (k & 15) * 4
      const int co = k >> 4, tq = t0 + (k & 15) * 4;
      const bool row_ok = o < a.D1;
      const float* p = a.dy + (((long long)b * CO + co) * a.D1 + (row_ok ? o : 0)) * a.T + tq;
      const float* z = g_zero_c1;
      v.x = *((row_ok && tq < a.T) ? p : z);
      v.y = *((row_ok && tq + 1 < a.T) ? p + 1 : z);
      v.z = *((row_ok && tq + 2 < a.T) ? p + 2 : z);
      v.w = *((row_ok && tq + 3 < a.T) ? p + 3 : z);
    };
    auto dy_store = [&](int buf, int k, const f32x4& v) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const int co = k >> 4, tq = (k & 15) * 4;
      *reinterpret_cast<bf16x4*>(dyl + buf * W_DY + co * W_PITCH + tq * 2) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    };
    // ---- prologue: zero slot, window of o = 0 (rows f = -20 .. 20 -> slots 0 .. 40), dY tile of o = 0
    for (int c = tid; c < W_SLOT / 16; c += 256) *reinterpret_cast<u32x4*>(zslot + c * 16) = u32x4{0u, 0u, 0u, 0u};
    for (int c = tid; c < KD * 128; c += 256) {
      const int r = c >> 7, k = c & 127;
      *reinterpret_cast<u32x4*>(ring + r * W_SLOT + (k >> 3) * W_PITCH + (k & 7) * 16) = *reinterpret_cast<const u32x4*>(row_src(r - PD, k >> 3, k & 7));
    }
    {
      f32x4 v0, v1;
      dy_load(0, tid, v0);
      dy_load(0, tid + 256, v1);
      dy_store(0, tid, v0);
      dy_store(0, tid + 256, v1);
    }
    // Data of step o that is not in LDS yet = R(o): the two new rows f = 2o + 19, 2o + 20 (256 chunks, one per thread) and the
    // dY tile of row o.  R(o + 4) is loaded into registers during step o and published at the end of step o + 3 (see the loop).
    struct Regs { u32x4 nr; f32x4 d0, d1; };
    const int nrow = tid >> 7, nk = tid & 127;
    // R(o) by 9 loads whose addresses are a UNIFORM base (a scalar function of o) + a 32-bit lane offset fixed for the whole block: the
    // per-step index arithmetic (64-bit multiplies, pointer selects: 1150 clocks per step with the matrix pipe idle, scripts/probe_conv1.hip)
    // is gone.  Steps of a dY row beyond T are CLAMPED to the row's last step instead of read from a zero page (their X16T partners are
    // zero and dY is finite), rows beyond D1 to the last row (never published); input rows beyond F read the zero page (wave-uniform).
    unsigned dyoff[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = tid + 256 * h, co = k >> 4, tq = t0 + (k & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) dyoff[h][j] = (unsigned)(((long long)co * a.D1 * a.T + min(tq + j, a.T - 1)) * 4);
    }
    const unsigned rowoff = (unsigned)((((long long)nrow * NC + (nk >> 3)) * a.Tp + t0 + (nk & 7) * 8) * 2);
    const char* dy_b = reinterpret_cast<const char*>(a.dy + (long long)b * CO * a.D1 * a.T);
    const char* x_b = reinterpret_cast<const char*>(a.X16T + (long long)b * a.F * NC * a.Tp);
    const int nrow_u = __builtin_amdgcn_readfirstlane(nrow);
    // a time tile that lies completely inside T (all but the last one): a thread's four steps of a dY row are one dword-aligned dwordx4
    const bool full_tile = t0 + W_TT <= a.T;                                         // block-uniform
    auto fetch = [&](int o, Regs& r, auto full_c) {
      const int f = 2 * o - 1 + PD + nrow_u;                                       // wave-uniform (>= 0)
      const bool ok = f < a.F;
      const char* xs = ok ? x_b + (long long)f * NC * a.Tp * 2 - (long long)nrow_u * NC * a.Tp * 2 : reinterpret_cast<const char*>(g_zero_c1);
      r.nr = *reinterpret_cast<const u32x4*>(xs + (ok ? rowoff : 0u));
      const char* ds_ = dy_b + (long long)min(o, a.D1 - 1) * a.T * 4;
      if constexpr (decltype(full_c)::value) {
        const f32x4_c1 q0 = *reinterpret_cast<const f32x4_c1*>(ds_ + dyoff[0][0]), q1 = *reinterpret_cast<const f32x4_c1*>(ds_ + dyoff[1][0]);
        r.d0 = f32x4{q0.x, q0.y, q0.z, q0.w};
        r.d1 = f32x4{q1.x, q1.y, q1.z, q1.w};
      } else {
        r.d0 = f32x4{*reinterpret_cast<const float*>(ds_ + dyoff[0][0]), *reinterpret_cast<const float*>(ds_ + dyoff[0][1]),
                     *reinterpret_cast<const float*>(ds_ + dyoff[0][2]), *reinterpret_cast<const float*>(ds_ + dyoff[0][3])};
        r.d1 = f32x4{*reinterpret_cast<const float*>(ds_ + dyoff[1][0]), *reinterpret_cast<const float*>(ds_ + dyoff[1][1]),
                     *reinterpret_cast<const float*>(ds_ + dyoff[1][2]), *reinterpret_cast<const float*>(ds_ + dyoff[1][3])};
      }
    };
    auto publish = [&](int o, const Regs& r) {           // R(o) -> ring slots outside the window of step o - 1, dY buffer o & 1
      if (o < a.D1) {
        const int slot = (2 * o - 1 + nrow + 2 * PD) % W_NR;
        *reinterpret_cast<u32x4*>(ring + slot * W_SLOT + (nk >> 3) * W_PITCH + (nk & 7) * 16) = r.nr;
        dy_store(o & 1, tid, r.d0);
        dy_store(o & 1, tid + 256, r.d1);
      }
    };
    auto compute = [&](int o) {                          // MFMAs of step o: K = 64 steps, tiles j = wave, wave + 4, ...
      const char* ap = dyl + (o & 1) * W_DY + l31 * W_PITCH + half * 16;
      const int cl = l31 & 15, kdl = l31 >> 4;
      const int s0 = (2 * o + 2 * wave + kdl) % W_NR;     // ring slot of this lane's kernel row in tile i = 0; tile i adds 8 rows
      const char* bp[W_TPW];
#pragma unroll
      for (int i = 0; i < W_TPW; ++i) {
        const int kd = 2 * (wave + 4 * i) + kdl;          // input row f = 2o + kd - 20 -> slot (2o + kd) % 44
        int slot = s0 + 8 * i;
        slot -= (slot >= W_NR) ? W_NR : 0;
        slot -= (slot >= W_NR) ? W_NR : 0;
        bp[i] = (kd < KD ? ring + slot * W_SLOT : zslot) + cl * W_PITCH + half * 16;
      }
      // k-step outer: the dY fragment is read once per k-step for all of the wave's tiles, and consecutive MFMAs go to different
      // accumulators (no dependent-issue stalls).  The fragments of k-step ks + 1 are read BEFORE the MFMAs of k-step ks issue (two register
      // sets): one wave per SIMD, so an LDS round trip in front of every k-step's MFMAs is fully exposed otherwise.
      bf16x8 af[2], bf[2][W_TPW];
      auto frags = [&](int ks, int set) {
        af[set] = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
#pragma unroll
        for (int i = 0; i < W_TPW; ++i) bf[set][i] = *reinterpret_cast<const bf16x8*>(bp[i] + ks * 32);   // (tile slots past the 21st read the zero slot)
      };
      frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < W_TT / 16; ++ks) {
        if (ks + 1 < W_TT / 16) frags(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < W_TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1], bf[ks & 1][i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // W_PF register sets: R(o + W_PF) is requested during step o and published at the end of step o + W_PF - 1.  With two sets (round 4) a
    // step could not be shorter than half an HBM round trip — one block per CU (113 KB of LDS), nothing else resident to hide it: 1.8 us per
    // step against 0.35 us of MFMAs (four sets: 1.57, profiles/r05_conv1_wgrad.txt).
    auto run = [&](auto full_c) {                        // (two copies of the loop: the number of loads per step is static in each)
      Regs rs[W_PF];
#pragma unroll
      for (int k = 1; k < W_PF; ++k) fetch(k, rs[k], full_c);
      __syncthreads();
      C1T_DECL;
      C1T(0);
      for (int o = 0; o < a.D1; o += W_PF) {
#pragma unroll
        for (int k = 0; k < W_PF; ++k) {
          if (o + k < a.D1) {                            // (block-uniform)
            fetch(o + k + W_PF, rs[k], full_c);
            C1T(1);
            compute(o + k);
            C1T(2);
            publish(o + k + 1, rs[(k + 1) % W_PF]);
            C1T(3);
            __syncthreads();
            C1T(4);
          }
        }
      }
      C1T_DUMP(4096 + (long long)blockIdx.y * gridDim.x + blockIdx.x);
    };
    if (full_tile) run(std::true_type{});
    else run(std::false_type{});
  }
#pragma unroll
  for (int i = 0; i < W_TPW; ++i) {
    const int j = wave + 4 * i;
    if (j < W_TILES) {
      const int kd = 2 * j + (l31 >> 4), c = l31 & 15;
      if (kd < KD && c < KTAPS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
          out[(co * KD + kd) * KTAPS + c] = acc[i][r];
        }
      }
    }
  }
}

// dW[idx] = sum over blocks of part[k][idx]; block = 32 outputs x 8 groups of partials, combined in a fixed order
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int nblk) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + col;
  float s = 0.f;
  if (idx < CO * KD * KTAPS) {
    int k = grp;
    for (; k + 24 < nblk; k += 32) {                        // four loads in flight, added in order
      const float v0 = part[(long long)k * (CO * KD * KTAPS) + idx], v1 = part[(long long)(k + 8) * (CO * KD * KTAPS) + idx];
      const float v2 = part[(long long)(k + 16) * (CO * KD * KTAPS) + idx], v3 = part[(long long)(k + 24) * (CO * KD * KTAPS) + idx];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nblk; k += 8) s += part[(long long)k * (CO * KD * KTAPS) + idx];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && idx < CO * KD * KTAPS) {
    s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][col];
    dW[idx] = s;
  }
}

inline int pad64(int T) { return (T + 63) / 64 * 64; }

}  // namespace

// sizes (bytes): which = 0 packed weights, 1 XB (forward operand: bf16 rows, pitch ds2_conv1_bf16_row_pitch), 2 X16T (weight-gradient operand image)
extern "C" int ds2_conv1_bf16_row_pitch(int T) { return xb_pitch(T); }
extern "C" size_t ds2_conv1_bf16_bytes(int which, int B, int F, int T) {
  if (which == 0) return (size_t)KD * CO * NC * 2;
  if (which == 1) return (size_t)B * F * xb_pitch(T) * 2;
  return (size_t)B * F * NC * pad64(T) * 2;
}

extern "C" int ds2_conv1_pack_bf16(const float* w1, void* wp, void* stream) {
  DS2_REQUIRE(w1 && wp, "ds2_conv1_pack_bf16: null pointer");
  hipLaunchKernelGGL(conv1_pack_kernel, dim3(ceil_div(KD * CO * NC, 256)), dim3(256), 0, (hipStream_t)stream, w1, (__bf16*)wp);
  DS2_LAUNCH_CHECK("conv1_pack_kernel");
  return 0;
}

// x (B,1,F,Tin) fp32 -> XB (B,F,P) and/or X16T (either may be NULL); T = output steps = (Tin + 2*5 - 11)/2 + 1, P = ds2_conv1_bf16_row_pitch(T)
extern "C" int ds2_conv1_gather_bf16(const float* x, void* XB, void* X16T, int B, int F, int Tin, void* stream) {
  DS2_REQUIRE(x && (XB || X16T) && B > 0 && F > 0 && Tin > 0, "ds2_conv1_gather_bf16: bad args");
  const int T = (Tin + 2 * PT - KTAPS) / 2 + 1, Tp = pad64(T);
  if (XB) {
    const int P = xb_pitch(T);
    DS2_REQUIRE(XB_LEFT + Tin + 8 <= P, "ds2_conv1_gather_bf16: row pitch %d too small for Tin=%d", P, Tin);
    const long long total = (long long)B * F * (P / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv1_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)XB, (long long)B * F, Tin, P);
    DS2_LAUNCH_CHECK("conv1_rows_kernel");
  }
  if (X16T) {
    const long long total = (long long)B * F * (Tp / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv1_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)nullptr, (__bf16*)X16T, (long long)B * F, Tin,
                       T, Tp);
    DS2_LAUNCH_CHECK("conv1_gather_kernel");
  }
  return 0;
}

// y1 (B,32,D1,T) fp32 = conv1(x) + bias, zero for t >= lens[b] (MaskConv).  XB from ds2_conv1_gather_bf16, wp from ds2_conv1_pack_bf16.
extern "C" int ds2_conv1_fwd_bf16_stat_blocks(int B, int F, int Tin) {
  (void)Tin;
  return F_SPLIT * ceil_div((F + 2 * PD - KD) / 2 + 1, F_OG) * B;
}

// stat_part: NULL, or ds2_conv1_fwd_bf16_stat_blocks() x 32 x 2 floats: per-block (sum, sum of squares) of every output channel over what
// the block stored (masked frames count as zeros: what BatchNorm2d sees) - finished by ds2_chanstats_from_partials
extern "C" int ds2_conv1_fwd_bf16_stats(const void* XB, const void* wp, const float* bias, const int* lens_dev, float* y1, int B, int F, int Tin,
                                        float* stat_part, void* stream) {
  DS2_REQUIRE(XB && wp && y1, "ds2_conv1_fwd_bf16: null pointer");
  C1Args a{};
  a.XB = (const __bf16*)XB; a.wp = (const __bf16*)wp; a.bias = bias; a.lens = lens_dev; a.y = y1;
  a.B = B; a.F = F; a.T = (Tin + 2 * PT - KTAPS) / 2 + 1; a.D1 = (F + 2 * PD - KD) / 2 + 1; a.P = xb_pitch(a.T);
  static bool attr_set = false;
  if (!attr_set) {
    DS2_HIP(hipFuncSetAttribute((const void*)conv1_bf16_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(conv1_bf16_fwd_kernel, dim3(F_SPLIT, ceil_div(a.D1, F_OG), B), dim3(512), F_LDS, (hipStream_t)stream, a, stat_part);
  DS2_LAUNCH_CHECK("conv1_bf16_fwd_kernel");
  return 0;
}

extern "C" int ds2_conv1_fwd_bf16(const void* XB, const void* wp, const float* bias, const int* lens_dev, float* y1, int B, int F, int Tin,
                                  void* stream) {
  return ds2_conv1_fwd_bf16_stats(XB, wp, bias, lens_dev, y1, B, F, Tin, nullptr, stream);
}

extern "C" size_t ds2_conv1_wgrad_bf16_workspace_bytes(int B, int Tin) {
  const int T = (Tin + 2 * PT - KTAPS) / 2 + 1;
  return (size_t)B * ceil_div(T, W_TT) * CO * KD * KTAPS * sizeof(float);
}

// dW1 (32,1,41,11) fp32 from X16T (ds2_conv1_gather_bf16) and dY1 (B,32,D1,T) fp32 (already zero beyond each length)
extern "C" int ds2_conv1_wgrad_bf16(const void* X16T, const float* dy1, const int* lens_dev, float* dW1, int B, int F, int Tin, void* ws,
                                    size_t ws_bytes, void* stream) {
  DS2_REQUIRE(X16T && dy1 && dW1 && ws, "ds2_conv1_wgrad_bf16: null pointer");
  DS2_REQUIRE(ws_bytes >= ds2_conv1_wgrad_bf16_workspace_bytes(B, Tin), "ds2_conv1_wgrad_bf16: workspace too small");
  C1WArgs a{};
  a.X16T = (const __bf16*)X16T; a.dy = dy1; a.lens = lens_dev; a.part = (float*)ws;
  a.B = B; a.F = F; a.T = (Tin + 2 * PT - KTAPS) / 2 + 1; a.Tp = pad64(a.T); a.D1 = (F + 2 * PD - KD) / 2 + 1;
  static bool attr_set = false;
  if (!attr_set) {
    DS2_HIP(hipFuncSetAttribute((const void*)conv1_bf16_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ntt = ceil_div(a.T, W_TT);
  hipLaunchKernelGGL(conv1_bf16_wgrad_kernel, dim3(ntt, B), dim3(256), W_LDS, s, a);
  DS2_LAUNCH_CHECK("conv1_bf16_wgrad_kernel");
  hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(ceil_div(CO * KD * KTAPS, 32)), dim3(256), 0, s, (const float*)ws, dW1, ntt * B);
  DS2_LAUNCH_CHECK("conv1_wgrad_reduce_kernel");
  return 0;
}
