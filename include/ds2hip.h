/* libds2hip — C ABI of the MI355X-native (gfx950) DeepSpeech2 train-step kernels.
 *
 * Drop-in boundary for the hot path of zakuro-ai/asr (`asr_deepspeech`).  The reference has no FFI
 * of its own (it is pure Python over torch ops); each entry point below replaces the torch/ATen op
 * sequence at the cited reference call site (paths relative to the reference tree).  The Python
 * host layer `asr_amd/` binds these with ctypes (see INTEGRATION.md for the stub a maintainer of
 * the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; the library never
 *     allocates, frees or retains memory: caller owns inputs, outputs, saved tensors, workspaces.
 *   - `stream` is a hipStream_t (as void*); all work is enqueued asynchronously on it.
 *   - return 0 on success, negative on error; `ds2_last_error()` returns a thread-local message.
 *   - threading: the library keeps NO mutable global state besides that thread-local message.  Everything the recurrence entry points
 *     remember between calls (which kernel family the last call took, the cooldown after a starved persistent launch, the enable
 *     switches, the debug selectors, where the starvation record and the poison word live) is in a caller-owned `ds2_rnn_ctx`
 *     (below); one context per thread / stream that launches recurrences, calls on different contexts are independent.  Environment
 *     switches (DS2_*) are tuning / A-B overrides read once and never written.
 *   - all tensors fp32, row-major, contiguous unless a pitch (`ld*`) is given.
 *   - lengths (`lens_dev`) are int32 per-sample valid OUTPUT frame counts (after the conv stack,
 *     modules/deepspeech.py:275-288), sorted descending as the reference requires (blocks.py:87).
 */
#ifndef DS2HIP_H
#define DS2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ds2_version(void);
const char* ds2_last_error(void);
int ds2_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);
int ds2_ablation_build(void);

/* ---- caller-owned state of the recurrence entry points (modules/blocks.py:84-93 is stateless in the reference: aten::gru / aten::lstm) ----
 * The caller allocates the struct (any memory it owns), zeroes it and calls ds2_rnn_ctx_init with
 *   status_dev   device memory, 8 ints, zero-initialised: the starvation record of persistent launches made through this context
 *   poison_host / poison_dev   optional (both NULL, or both set): host and device address of ONE pinned, mapped int that
 *                ds2_rnn_poison_if_starved raises — lets a caller that never synchronises notice a starved launch with a host read.
 * The library never allocates any of it and holds no reference beyond the call it is passed to.  ctx == NULL is accepted by the launch
 * entry points: the recurrence then runs on the one-launch-per-step kernels (no persistent launch: nothing to remember, nothing to record). */
typedef struct ds2_rnn_ctx {
  int size;                 /* sizeof(ds2_rnn_ctx) of the build that initialised it (ABI check) */
  int persist_fwd, persist_bwd;   /* which recurrences may run as ONE persistent launch (ds2_rnn_persistent_enable) */
  int cooldown;             /* recurrence calls left on the one-launch-per-step kernels after a starved launch (0 armed, < 0 never re-arm) */
  int rearm_calls;          /* length of that cooldown (default 64 calls; 0 = never re-arm); DS2_RNN_REARM_CALLS at init */
  int starved_total;        /* launches through this context that starved (reporting) */
  int last_path;            /* out: bits describing what the last forward / backward call launched (ds2_rnn_last_path) */
  int last_bwd_kind;        /* out: 0 step kernels, 1 all-gather persistent, 2 K-split persistent */
  int debug_flags;          /* kernel-family selectors (ds2_debug_flags) */
  int ws_prearmed;          /* in, one-shot (cleared by the next recurrence call through this context): the caller has filled that call's whole
                             * workspace with 0xff bytes (ds2_memset_async) EARLIER in the stream — a persistent launch then skips its own
                             * fill of the exchange buffers, which otherwise sits between the projection GEMM and the launch that needs every CU */
  int reserved[6];
  int* status_dev;
  int* poison_host;
  int* poison_dev;
} ds2_rnn_ctx;
int ds2_rnn_ctx_init(ds2_rnn_ctx* ctx, int* status_dev, int* poison_host, int* poison_dev);
/* hipMemsetAsync through the C ABI (workspace pre-arming, see ws_prearmed). */
int ds2_memset_async(void* dst, int value, size_t bytes, void* stream);
/* Kernel-family selectors of the recurrence (every selection computes the full result; used by the parity tests and A/B scripts):
 * 8 / 16 alternative tile shapes of the wide step kernels, 64 one launch per time step instead of the persistent kernels, 128 the
 * all-gather persistent backward kernel instead of the K-split one.  Returns the previous value; 0 = production.  Bits 1 / 2 (skip the
 * recurrent product / the gate epilogue, scripts/ablate_rnn.py) are honoured only by a library built with -DDS2_ABLATE
 * (ds2_ablation_build() == 1); the shipped library masks them off. */
int ds2_debug_flags(ds2_rnn_ctx* ctx, int flags);

/* ---- dense GEMM on the f32 matrix cores -------------------------------------------------------
 * C[M,N] (+)= op(A) op(B) (+ bias[N]);  transA: A stored (K,M);  transB: B stored (N,K).
 * Replaces aten::addmm/mm inside aten::gru / aten::lstm (input projections, modules/blocks.py:76-78,88),
 * nn.Linear (modules/deepspeech.py:105) and their autograd backward (dX, dW).
 * batch > 1: strided batches; splitk > 1: deterministic split-K through `workspace`. */
size_t ds2_gemm_f32_workspace_bytes(int M, int N, int batch, int splitk);
int ds2_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, long long strideA, const float* B, int ldb,
                 long long strideB, float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch, int splitk,
                 void* workspace, size_t workspace_bytes, void* stream);

/* bf16-operand / fp32-accumulate GEMM (v_mfma_f32_32x32x16_bf16), "NT" form only: C[M,N] (+)= A[M,K] B[N,K]^T (+bias).
 * Used for the RNN input projections and their dX / dW when the model runs with precision="bf16" (BASELINE configs[2],[4]);
 * the cast passes produce the K-contiguous bf16 operands (dst pitch % 8 == 0, pad columns zero-filled). */
size_t ds2_gemm_bf16_workspace_bytes(int M, int N, int batch, int splitk);
int ds2_gemm_bf16_nt(int M, int N, int K, const void* A, int lda, long long strideA, const void* B, int ldb, long long strideB, float* C,
                     int ldc, long long strideC, const float* bias, int accumulate, int batch, int splitk, void* workspace,
                     size_t workspace_bytes, void* stream);
/* C[M,N] **bf16** = A[M,K] B[N,K]^T + bias (fp32 accumulation and bias add, ONE rounding at the store; ldc in bf16 elements, N, ldc % 8 == 0):
 * the x-projections of a recurrent layer in the bf16 training mode (aten::addmm inside aten::gru / lstm, blocks.py:76-78, 88), read once by
 * ds2_rnn_fwd_x.  Returns 1 — nothing launched, call ds2_gemm_bf16_nt — where the four-wave kernel does not apply (K % 64 != 0, fewer
 * 256 x 256 tiles than CUs, alignment); 0 = launched; < 0 = error. */
int ds2_gemm_bf16_nt_obf16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, void* stream);
/* n contiguous bf16 -> fp32 (n % 8 == 0, 16-byte aligned bases). */
int ds2_cast_f32_from_bf16(const void* src, float* dst, long long n, void* stream);
/* "TN" form: C[M,N] (+)= A[K,M]^T B[K,N], both operands bf16 row-major with the reduction index on the rows (pitches lda / ldb).
 * The weight-gradient product of the recurrent layers (dW = dGx^T [Xn | h], K = T*B) without a transposed copy of any operand.
 * M, N, lda, ldb, strides multiples of 8; batch > 1: independent products at the given element strides (may be negative). */
int ds2_gemm_bf16_tn(int M, int N, int K, const void* A, int lda, long long strideA, const void* B, int ldb, long long strideB, float* C,
                     int ldc, long long strideC, int accumulate, int batch, int splitk, void* workspace, size_t workspace_bytes,
                     void* stream);
/* Grouped TN products in ONE launch of a kernel sized to run BESIDE a persistent backward recurrence (4 waves of <= 128 registers, at most
 * one workgroup per CU): all weight-gradient products of one recurrent layer — dW_hh of both directions, a GRU's n-gate rows, dW_ih
 * (asr_deepspeech/modules/blocks.py:76-78,88 under autograd) — as one flat list of 128 x 128 tiles walked by one workgroup per CU; every
 * tile is a full reduction over K_p (no split-K, no workspace).  Same operand rules as ds2_gemm_bf16_tn; up to 8 problems. */
typedef struct ds2_tn_problem {
  const void* A; const void* B; float* C;
  int M, N, K, lda, ldb, ldc;
} ds2_tn_problem;
int ds2_gemm_bf16_tn_group(int nprob, const ds2_tn_problem* problems, int max_workgroups, void* stream);
/* The same problem list through the 256 x 256 TN kernel with ONE split-K factor for all of them: one GEMM launch whose work items are
 * (problem, K slice, tile) — a layer's 192 weight-gradient tiles x 4 slices are 768 equal items = three full rounds of the chip, where
 * three separate launches each had their own ramp and tail and up to 8 slabs per tile — and one reduce launch.  workspace: the slabs.
 * Consecutive problems that name the SAME C are TERMS of one product and are summed (the fp32 mode's hi.hi + hi.lo + lo.hi on views of
 * split operands, ds2_split_bf16): up to 16 entries per launch. */
size_t ds2_gemm_bf16_tn_splitk_group_workspace_bytes(int nprob, const ds2_tn_problem* problems, int splitk);
int ds2_gemm_bf16_tn_splitk_group(int nprob, const ds2_tn_problem* problems, int splitk, void* workspace, size_t workspace_bytes, void* stream);
/* ds2_gemm_bf16_tn_splitk_group with an epilogue on product ep_index, applied by the reduce launch: C = (A^T B) diag(scale) + rowv (x) shift
 * (scale / shift: N floats, 16-byte aligned; rowv: M floats) — dW_ih of a projection whose BatchNorm1d was folded into it (ds2_wih_fold_bf16).
 * Returns 1, nothing launched, when that product would have a single slab: use the plain entry + ds2_scale_rank1_f32. */
int ds2_gemm_bf16_tn_splitk_group_ep(int nprob, const ds2_tn_problem* probs, int splitk, int ep_index, const float* ep_scale, const float* ep_rowv,
                                     const float* ep_shift, void* workspace, size_t workspace_bytes, void* stream);
int ds2_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, void* stream);
/* fp32 mode (precision="fp32", BASELINE configs[1],[3]), large GEMMs: every fp32 operand is SPLIT into two bf16 terms, x = hi + lo with
 * hi = bf16(x), lo = bf16(x - hi) (x is represented to 2^-18 relative), and a product is taken as a_hi b_hi + a_hi b_lo + a_lo b_hi on the bf16
 * matrix cores with fp32 accumulation (the dropped lo.lo term is 2^-18 of the product; the result is within ~1e-5 of the fp32 product,
 * two orders inside north_star's 1e-3) — torch.nn.functional.linear / autograd's mm (blocks.py:76-78,88) at ~5x the fp32-MFMA roof.
 * dst (R, ld_dst) bf16 = blocks of pad8(C) columns each: order 0 [hi | hi | lo] (A operand), 1 [hi | lo | hi] (B operand: one NT product
 * over a reduction index 3 pad8(C) long IS the three-term product), 2 [hi | lo] (TN products take row-pitched views of the blocks). */
int ds2_split_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, int order, void* stream);
int ds2_cast_transpose_bf16(const float* src, int ld_src, void* dst, int ld_dst, int R, int Cc, void* stream);
/* both copies from ONE read of src: dst_r (R, ld_r) = bf16(src) (NULL: skipped), dst_t (C, ld_t) = bf16(src)^T, pads zero
 * (ld_r % 8 == 0, C <= ld_r <= C rounded up to a multiple of 64; ld_t % 8 == 0, ld_t >= R); colsum (C) optional: column sums of src from the same read
 * (the bias gradient db_ih = sum_rows dGx), then ws >= ds2_cast_bf16_both_workspace_bytes(R, C) */
size_t ds2_cast_bf16_both_workspace_bytes(int R, int Cc);
int ds2_cast_bf16_both(const float* src, int ld_src, void* dst_r, int ld_r, void* dst_t, int ld_t, int R, int Cc, float* colsum, void* ws,
                       size_t ws_bytes, void* stream);
/* bf16 (R, C) row-major (pitch ld_src % 8 == 0) -> bf16 (C, ld_t) transposed, pads zero; colsum (C) fp32 optional: column sums from
 * the same read (ws >= ds2_cast_bf16_both_workspace_bytes(R, C)). */
int ds2_transpose_bf16(const void* src, int ld_src, void* dst_t, int ld_t, int R, int Cc, float* colsum, void* ws, size_t ws_bytes,
                       void* stream);

/* ---- BatchNorm1d over (T*B, H) rows, padding rows included -----------------------------------
 * modules/blocks.py:75,85-86 (SequenceWise(BatchNorm1d)) and modules/deepspeech.py:104 (fc block).
 * stats: biased variance for normalisation; running stats updated with momentum and the unbiased
 * variance when run_mean/run_var are non-null (torch.nn.BatchNorm1d training semantics). */
size_t ds2_colreduce_workspace_bytes(int M, int H);
int ds2_colstats_f32(const float* X, int ldx, int M, int H, float* mean, float* var, float* run_mean, float* run_var, float momentum,
                     void* ws, size_t ws_bytes, void* stream);
/* Y = Xa + Xb (sum of the two RNN directions, modules/blocks.py:92) fused with the statistics of Y */
int ds2_add_colstats_f32(const float* Xa, int lda, const float* Xb, int ldb, float* Y, int ldy, int M, int H, float* mean, float* var,
                         float* run_mean, float* run_var, float momentum, void* ws, size_t ws_bytes, void* stream);
int ds2_colsum_f32(const float* X, int ldx, int M, int H, float* sum, float* sumsq, void* ws, size_t ws_bytes, void* stream);
int ds2_bn1d_apply_f32(const float* X, int ldx, float* Y, int ldy, int M, int H, const float* mean, const float* var,
                       const float* gamma, const float* beta, float eps, void* stream);
/* same, Y written as bf16 (M, ldy), ldy % 8 == 0, pad columns zero: feeds the bf16 input-projection GEMM without a cast pass */
int ds2_bn1d_apply_bf16(const float* X, int ldx, void* Y, int ldy, int M, int H, const float* mean, const float* var, const float* gamma,
                        const float* beta, float eps, void* stream);
/* (dX may be NULL: only the column sums dgamma / dbeta are produced — see ds2_rnn_bwd_bn) */
int ds2_bn1d_bwd_f32(const float* dY, int lddy, const float* X, int ldx, float* dX, int lddx, int M, int H, const float* mean,
                     const float* var, const float* gamma, float eps, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                     void* stream);

/* ---- BatchNorm1d folded into the input projection of the recurrent layer behind it (bf16 training mode, round 6) --------------------------
 * blocks.py:85-86 (SequenceWise(BatchNorm1d)) + :92 (direction sum of the layer in front) + aten::addmm inside aten::gru / lstm (:88).
 * y = Xa + Xb is never written.  ds2_center_colstats writes the CENTRED sum yc = y - m0 as the bf16 GEMM operand (M, ldyc; pad columns
 * zero), m0 = the column means of y taken from hsum (2, ntiles, H: ds2_rnn_fwd_x), and returns the batch statistics of y (mean, biased var:
 * what nn.BatchNorm1d computes; running statistics updated with momentum and the unbiased variance) plus delta = mean - m0, the mean of yc.
 * ds2_wih_fold_bf16 then folds the normalisation into the projection:  BN(y) W^T + b = yc (W diag(s))^T + (b + W c),
 * s = gamma rsqrt(var + eps), c = beta - delta s: W2 = bf16(W diag(s)) (R rows, pitch ldw2), bias2 = b + W c in fp32, colscale = s, colshift = c.
 * Backward: the weight gradient of the folded projection is dW = (dGx^T yc) diag(s) + db (x) c (ds2_scale_rank1_f32 on the TN product);
 * the BatchNorm backward formulas hold on (yc, delta, var) unchanged — ds2_bn1d_bwd_xbf16 / ds2_rnn_bwd_bn_xbf16 take the bf16 operand.
 * workspace of ds2_center_colstats: ds2_colreduce_workspace_bytes(M, H) + H * sizeof(float). */
int ds2_center_colstats(const float* Xa, int lda, const float* Xb, int ldb, const float* hsum, int ntiles, void* Yc_bf16, int ldyc, int M, int H,
                        float* mean, float* var, float* delta, float* run_mean, float* run_var, float momentum, void* ws, size_t ws_bytes,
                        void* stream);
int ds2_wih_fold_bf16(const float* W, int ldw, const float* bias, int R, int I, const float* var, const float* gamma, const float* beta,
                      const float* delta, float eps, void* W2_bf16, int ldw2, float* bias2, float* colscale, float* colshift, void* stream);
/* C[r][c] = C[r][c] * scale[c] + rowv[r] * shift[c], (R, N) fp32 in place (N, ldc % 4 == 0, 16-byte aligned) */
int ds2_scale_rank1_f32(float* C, int ldc, int R, int N, const float* scale, const float* rowv, const float* shift, void* stream);
/* ds2_bn1d_bwd_f32 with the BatchNorm input as bf16 (pitch ldx % 4 == 0; dX, when given, needs H % 4 == 0); `mean` = the mean of X itself */
int ds2_bn1d_bwd_xbf16(const float* dY, int lddy, const void* X_bf16, int ldx, float* dX, int lddx, int M, int H, const float* mean, const float* var,
                       const float* gamma, float eps, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm2d + Hardtanh(0,20) + MaskConv time mask on (B,C,D,T) ----------------------------
 * modules/deepspeech.py:62-63,65-66 under modules/blocks.py:48-55 (mask after EVERY sub-module). */
size_t ds2_chanreduce_workspace_bytes(int C);
int ds2_bn2d_stats_f32(const float* Y, int B, int C, int D, int T, float* mean, float* var, float* run_mean, float* run_var,
                       float momentum, void* ws, size_t ws_bytes, void* stream);
int ds2_bn2d_act_fwd_f32(const float* Y, float* A, int B, int C, int D, int T, const int* lens_dev, const float* mean,
                         const float* var, const float* gamma, const float* beta, float eps, void* stream);
int ds2_bn2d_act_bwd_f32(const float* Y, const float* dA, float* dY, int B, int C, int D, int T, const int* lens_dev, const float* mean,
                         const float* var, const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, void* ws,
                         size_t ws_bytes, void* stream);

/* bf16 mode: the same two blocks with the layout casts that follow them fused into the pass (deepspeech.py:62-66 + the operand
 * preparation of conv2 forward / dgrad / wgrad): each result may be written as fp32 (B,32,D,T), as zero-padded bf16 rows
 * (B,32,D,Tp), Tp = ds2_conv_padded_pitch(T), and as channels-last bf16 (B,D,T,32) — NULL skips a form.  The backward form also
 * returns dbias (32) = per-channel sums of dY, the gradient of the bias of the convolution in front (deepspeech.py:61,64). */
int ds2_bn2d_act_fwd_fused(const float* Y, int B, int D, int T, const int* lens_dev, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, float* a_f32, void* a_pad, void* a_nhwc, void* stream);
/* second conv stage, bf16 mode: the same block fused with the (B,32*D,T) -> (T*B, 32*D) collapse (deepspeech.py:135-137) and the cast to
 * the first recurrent layer's bf16 GEMM operand; x_f32 (pitch 32*D) and x_bf16 (pitch ld_bf >= 32*D, a multiple of 8; the columns behind
 * 32*D are written as zeros) may each be NULL */
int ds2_bn2d_act_collapse(const float* Y, int B, int D, int T, const int* lens_dev, const float* mean, const float* var,
                          const float* gamma, const float* beta, float eps, float* x_f32, void* x_bf16, int ld_bf, void* stream);
size_t ds2_bn2d_act_bwd_fused_workspace_bytes(int B, int D, int T);
int ds2_bn2d_act_bwd_fused(const float* Y, const float* dA, int B, int D, int T, const int* lens_dev, const float* mean,
                           const float* var, const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta,
                           float* dbias, float* dy_f32, void* dy_pad, void* dy_nhwc, void* ws, size_t ws_bytes, void* stream);
/* dir 0: (B,F,T) -> (T,B,F) = view+transpose+contiguous of modules/deepspeech.py:135-137; dir 1: inverse */
int ds2_transpose_bft_f32(const float* src, float* dst, int B, int F, int T, int dir, void* stream);
int ds2_transpose2d_f32(const float* src, int ld_src, long long stride_src, float* dst, int ld_dst, long long stride_dst, int R, int Cc,
                        int batch, void* stream);

/* ---- conv front-end (MFMA implicit GEMM) -------------------------------------------------------
 * conv1 = nn.Conv2d(1,32,(41,11),(2,2),(20,5)) modules/deepspeech.py:61; conv2 = nn.Conv2d(32,32,(21,11),(2,1),(10,5)) :64;
 * forward epilogues add the bias and apply the MaskConv mask (modules/blocks.py:50-55). */
void ds2_conv_dims(int F, int Tin, int* D1, int* D2, int* T);
size_t ds2_conv_packed_floats(int which /*0: conv1 fwd, 1: conv2 fwd, 2: conv2 dgrad*/);
int ds2_conv_pack_f32(const float* w1, const float* w2, float* wpk1, float* wpk2, float* wpk2d, void* stream);
int ds2_conv1_fwd_f32(const float* x, const float* wpk1, const float* bias, const int* lens_dev, float* y1, int B, int F, int Tin,
                      void* stream);
int ds2_conv2_fwd_f32(const float* a1, const float* wpk2, const float* bias, const int* lens_dev, float* y2, int B, int D1, int T,
                      void* stream);
int ds2_conv2_dgrad_f32(const float* dy2, const float* wpk2d, float* da1, int B, int D1, int T, void* stream);
size_t ds2_conv_wgrad_workspace_bytes(int which /*0: conv1, 1: conv2*/, int B, int F);
int ds2_conv1_wgrad_f32(const float* x, const float* dy1, const int* lens_dev, float* dW1, int B, int F, int Tin, int accumulate, void* ws,
                        size_t ws_bytes, void* stream);
int ds2_conv2_wgrad_f32(const float* a1, const float* dy2, const int* lens_dev, float* dW2, int B, int D1, int T, int accumulate, void* ws,
                        size_t ws_bytes, void* stream);

/* conv2 forward / data-gradient with bf16 MFMA operands (precision="bf16"): activations channels-last (B,D,T,32) bf16
 * (ds2_nhwc_bf16_f32 converts from the (B,32,D,T) fp32 layout), weights re-packed by ds2_conv2_pack_bf16; fp32 output. */
size_t ds2_conv2_bf16_packed_bytes(int which /*0: forward, 1: dgrad even rows, 2: dgrad odd rows*/);
int ds2_conv2_pack_bf16(const float* w2, void* wf, void* wd0, void* wd1, void* stream);
int ds2_nhwc_bf16_f32(const float* src, void* dst, int B, int D, int T, void* stream);
int ds2_conv2_fwd_bf16(const void* a1_nhwc, const void* wf, const float* bias, const int* lens_dev, float* y2, int B, int D1, int T,
                       void* stream);
int ds2_conv2_fwd_bf16_stat_blocks(int B, int D1, int T);
int ds2_conv2_fwd_bf16_stats(const void* a1_nhwc, const void* wf, const float* bias, const int* lens_dev, float* y2, int B, int D1, int T,
                             float* stat_part, void* stream);
size_t ds2_chanstats_from_partials_workspace_bytes(void);
int ds2_chanstats_from_partials(const float* part, int nblk, int C, double count, float* mean, float* var, float* run_mean, float* run_var,
                                float momentum, void* ws, size_t ws_bytes, void* stream);
int ds2_conv2_dgrad_bf16(const void* dy2_nhwc, const void* wd0, const void* wd1, float* da1, int B, int D1, int T, void* stream);

/* conv2 weight gradient with bf16 MFMA operands: zero-padded bf16 copies (R, Tp) of the (B,32,D,T) tensors, Tp = ds2_conv_padded_pitch(T) */
int ds2_conv_padded_pitch(int T);
int ds2_padcast_bf16(const float* src, void* dst, long long R, int T, void* stream);
size_t ds2_conv2_wgrad_bf16_workspace_bytes(int B, int D1);
int ds2_conv2_wgrad_bf16(const void* a1p, const void* dy2p, const int* lens_dev, float* dW2, int B, int D1, int T, void* ws, size_t ws_bytes,
                         void* stream);
/* The same weight gradient from the CHANNELS-LAST bf16 operands the conv2 forward / data-gradient kernels already take (a1 (B,D1,T,32),
 * dy2 (B,D2,T,32)): no padded copies, operands go global -> LDS by DMA and the kernel-tap shift is a row offset of the time-major LDS image
 * (Conv2d weight gradient under loss.backward(), modules/deepspeech.py:64 / trainers/deepspeech_trainer.py:87).  Workspace: as above. */
int ds2_conv2_wgrad_nhwc_bf16(const void* a1_nhwc, const void* dy2_nhwc, const int* lens_dev, float* dW2, int B, int D1, int T, void* ws,
                              size_t ws_bytes, void* stream);

/* ---- bidirectional GRU / LSTM recurrence -------------------------------------------------------
 * pack_padded_sequence -> aten::gru / aten::lstm -> pad_packed_sequence, modules/blocks.py:87-89, h0 = 0,
 * gate order r,z,n (GRU) / i,f,g,o (LSTM); gates = 3 | 4.  See asr_amd/csrc/rnn.hip for buffer roles. */
size_t ds2_rnn_packed_bytes(int gates, int H, int which /*0: forward operand, 1: backward operand*/, int bf16);
/* re-pack W_hh = [weight_hh_l0 ; weight_hh_l0_reverse] (2,G*H,H) fp32 into MFMA-fragment order, fp32 or bf16 fragments
 * (once per optimizer step).  bf16 = 2 (here, in ds2_rnn_packed_bytes, ds2_rnn_fwd_workspace_bytes and ds2_rnn_fwd*): the fp32 mode with the
 * SPLIT persistent forward recurrence — h_t and W_hh each as two bf16 planes, x = bf16(x) + bf16(x - bf16(x)), product = hi.hi + lo.hi + hi.lo on
 * the bf16 matrix cores with fp32 accumulation (fp32-grade: ~1e-6 of the fp32 kernels, far inside north_star's 1e-3): the forward operand
 * is then [fp32 fragments | hi fragments | lo fragments (| the ten-unit-slice hi / lo operand of csrc/rnn_fwd_u10.h when H % 160 == 0, H <= 1280)]
 * and the fp32 kernels remain the fallback (shape does not fit, cooldown). */
int ds2_rnn_pack_whh(int gates, const float* whh, void* wp_fwd, void* wp_bwd, int H, int bf16, void* stream);
/* Status of the persistent recurrences (a layer's whole recurrence in ONE launch whose workgroups exchange h_t / dGh_t through memory;
 * it needs every workgroup resident at once).  out8 = {starved, slice | workgroup, tile | XCD, direction | kind, step, wave, pending
 * chunk mask, L2-local exchange}: starved != 0 (1 forward, 2 backward, 3 the launch never became resident) means a wave gave up
 * polling since the last call (DS2_RNN_SPIN_LIMIT polls, default 2^20 ~ 1 s) and the results of that launch are invalid — the caller
 * must treat the step as failed.  Synchronises the device and clears the record.  After a starved launch the next DS2_RNN_REARM_CALLS
 * (default 64) recurrence calls run on the one-launch-per-step kernels, then the persistent kernels are armed again.
 * DS2_RNN_PERSISTENT=0 selects the one-launch-per-step kernels from the start; DS2_RNN_XCD_LOCAL=0 keeps the placement-independent
 * (sc1) exchange instead of the exchange through the group's own L2 (asr_amd/csrc/rnn.hip: persist_role). */
int ds2_rnn_persistent_status(ds2_rnn_ctx* ctx, int* out8);
/* reporting: out2 = {launches through this context that starved, recurrence calls left before the persistent kernels are armed
 * again (0 = armed, -1 = never)} */
int ds2_rnn_persistent_counters(const ds2_rnn_ctx* ctx, int* out2);
/* Inference path (DeepSpeech.forward in eval mode, modules/deepspeech.py:130-149): instead of a device synchronisation per forward, a
 * kernel in stream order that overwrites buf[0..n) (the logits) with NaN if a persistent launch before it recorded starvation; the record is
 * neither read by the host nor cleared (the next ds2_rnn_persistent_status, at a natural sync point, raises). */
int ds2_rnn_poison_if_starved(ds2_rnn_ctx* ctx, float* buf, size_t n, void* stream);
/* 1 if a ds2_rnn_poison_if_starved kernel has fired since the last ds2_rnn_persistent_status: a read of a pinned host word the kernel
 * sets, no synchronisation and no device call.  Lets inference callers that only ever call forward (deepspeech.py:130-149 in eval mode)
 * notice a starved launch and settle it (status call: report, clear, cooldown onto the step kernels) before their next forward. */
int ds2_rnn_poison_seen(const ds2_rnn_ctx* ctx);
/* device-side validity of the train step enqueued so far, evaluated when the kernel RUNS (stream order): flag_dev[0] = -1 if a persistent
 * recurrence launch starved, else 1 if the loss is finite and >= 0 [check_loss, functional.py:45-61], else 0.  Under data parallelism the
 * MIN over ranks is taken; the gated optimizer applies the update only for 1. */
int ds2_rnn_step_gate(const ds2_rnn_ctx* ctx, const float* loss_dev, int* flag_dev, void* stream);
/* Which recurrences may run as one persistent launch (default both).  Switch the backward one off when other kernels (collectives on a
 * communication stream) run on the device during backward: a persistent launch needs all of its workgroups resident at once. */
int ds2_rnn_persistent_enable(ds2_rnn_ctx* ctx, int forward, int backward);
/* Footprint of one workgroup of the K-split persistent backward recurrence for this (gates, H), read from the loaded binary: out3 =
 * {registers per lane, static LDS bytes, threads}; returns 1 if the shape has such a kernel, 0 if not.  The host side decides with it
 * whether ds2_gemm_bf16_tn_group (4 waves x 128 registers, 84 KB of LDS) fits on a CU BESIDE the recurrence of the layer below
 * (asr_deepspeech/modules/blocks.py:87-89 backward; the weight gradients of blocks.py:76-78 are off its critical path). */
int ds2_rnn_bwd_ksplit_footprint(int gates, int H, int* out3);
/* reporting: bit 0 / bit 1 set if the last ds2_rnn_fwd / ds2_rnn_bwd call through this context ran as a persistent launch (bench.py labels its roofline with it) */
int ds2_rnn_last_path(const ds2_rnn_ctx* ctx);
size_t ds2_rnn_fwd_workspace_bytes(int B, int H, int bf16);
/* gates_bf16: NULL, or a (T,B,2,H,4) bf16 buffer that receives the saved-for-backward record of every hidden unit as ONE 8-byte
 * store — GRU [r, z, n, W_hn h + b_hn], LSTM [i, f, g, o] — instead of four fp32 stores into gx / aux (gx is then left untouched
 * and, for GRU, aux is not written).  Pass the same buffer to ds2_rnn_bwd. */
int ds2_rnn_fwd(ds2_rnn_ctx* ctx, int gates, float* gx, const void* wp_fwd, const float* bhh, float* hbuf, float* aux, const int* lens_dev, int T, int B,
                int H, int bf16, void* gates_bf16, void* ws, size_t ws_bytes, void* stream);
/* ds2_rnn_fwd plus h_bf16: NULL, or a (T,B,2,H) bf16 buffer that receives a bf16 copy of hbuf (the K-row-major operand of the TN-form
 * dW_hh GEMM).  Written by a PERSISTENT launch only: check ds2_rnn_last_path() & 1 after the call. */
int ds2_rnn_fwd_ex(ds2_rnn_ctx* ctx, int gates, float* gx, const void* wp_fwd, const float* bhh, float* hbuf, float* aux, const int* lens_dev, int T, int B,
                   int H, int bf16, void* gates_bf16, void* h_bf16, void* ws, size_t ws_bytes, void* stream);
/* ds2_rnn_fwd_ex for the bf16 TRAINING mode (bf16 operands, packed gate records required) with two more optional operands.
 * gx_bf16: the x-projections as the **bf16** tensor ds2_gemm_bf16_nt_obf16 wrote (same (T,B,2,G*H) layout; half the bytes written by the
 *   projection and read here: aten::gru / aten::lstm of blocks.py:87-89 take them from aten::addmm at full precision — the rounding is part of
 *   the stated bf16-mode tolerance); gx may then be NULL.  Persistent kernels only: the call returns 1 — nothing launched, nothing counted —
 *   when it cannot run as a persistent launch (cooldown, forward kernel switched off, no persistent kernel for the shape); widen with
 *   ds2_cast_f32_from_bf16 and call again with gx.
 * hsum: (2, ceil(B/16), H) fp32, per direction and 16-row batch tile the sums over time of h — the column sums of the layer's output
 *   y = h_fwd + h_bwd (blocks.py:92) without a pass over it; written by a PERSISTENT launch only (ds2_rnn_last_path() & 1); input of
 *   ds2_center_colstats.
 * Returns 0 = done, 1 = see gx_bf16, < 0 = error. */
int ds2_rnn_fwd_x(ds2_rnn_ctx* ctx, int gates, float* gx, const void* gx_bf16, const void* wp_fwd, const float* bhh, float* hbuf, float* aux,
                  const int* lens_dev, int T, int B, int H, void* gates_bf16, void* h_bf16, float* hsum, void* ws, size_t ws_bytes, void* stream);
size_t ds2_rnn_bwd_workspace_bytes(int gates, int B, int H, int bf16);
/* dgx_bf16: NULL, or a (T,B,2,G*H) bf16 buffer that receives the gradient wrt the x-projections instead of gx (which then keeps
 * the gates): the bf16-mode GEMMs consume it directly.  gates_bf16: NULL, or the packed records written by ds2_rnn_fwd — read
 * instead of gx (and, for GRU, instead of aux, which is then output only: d(W_hn h + b_hn)); gx may be NULL when both are given. */
int ds2_rnn_bwd(ds2_rnn_ctx* ctx, int gates, const float* dy, int lddy, float* gx, float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev,
                int T, int B, int H, int bf16, void* dgx_bf16, const void* gates_bf16, void* ws, size_t ws_bytes, void* stream);

/* ds2_rnn_last_path() bits after a backward call: 2 = ran as ONE persistent launch; 4 = that launch was the K-split kernel (bf16, H a multiple
 * of 256: asr_amd/csrc/rnn_bwd_ksplit.h — bf16 partial sums of dh are exchanged instead of dGh; results equal those of the other kernel
 * families within 4e-3 relative L2 and sit at the same distance from an fp64 recurrence, run-to-run bit-identical).  A K-split launch that
 * is given dhn_bf16 writes ONLY that bf16 copy of d(W_hn h + b_hn), not the fp32 one into aux.
 * ds2_rnn_bwd plus two optional outputs of a PERSISTENT launch (check ds2_rnn_last_path() & 2 after the call; untouched otherwise):
 * dhn_bf16 (GRU): (T,B,2,H) bf16 copy of d(W_hn h + b_hn); bias_part: (B,2,4,H) fp32 per-batch-row sums over time of
 * [d r, d z, d n, d(hn)] (GRU) / [d i, d f, d g, d o] (LSTM) - their column sums over B are the bias gradients, so no pass over dGx. */
int ds2_rnn_bwd_ex(ds2_rnn_ctx* ctx, int gates, const float* dy, int lddy, float* gx, float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev,
                   int T, int B, int H, int bf16, void* dgx_bf16, const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws,
                   size_t ws_bytes, void* stream);

/* ds2_rnn_bwd_ex for a layer whose output y feeds a BatchNorm1d (SequenceWise(BatchNorm1d) of the next layer, modules/blocks.py:75,85-86, or
 * of the fc block, modules/deepspeech.py:104): dyn = gradient wrt that BatchNorm's OUTPUT, bn_x = its input (= y, pitch ldx), bn_mean /
 * bn_var / bn_gamma its batch statistics and weight, bn_s0 / bn_s1 the column sums of dyn and dyn * xhat (dbeta / dgamma of
 * ds2_bn1d_bwd_f32 called with dX = NULL).  Where the K-split persistent kernel takes the call it applies the elementwise half of the
 * BatchNorm backward on the fly (ds2_rnn_last_path() & 16) and dy_scratch is not touched; otherwise dy is materialised into dy_scratch
 * (T*B, H) and the call proceeds as ds2_rnn_bwd_ex.  dy_scratch may be NULL: the call then returns 1 — nothing launched, nothing counted —
 * when the buffer is needed after all, and the caller repeats it with one (the fused launch never allocates or touches (T*B, H) fp32).
 * Replaces autograd's native_batch_norm_backward + the recurrence backward. */
int ds2_rnn_bwd_bn(ds2_rnn_ctx* ctx, int gates, const float* dyn, int lddyn, const float* bn_x, int ldx, const float* bn_mean, const float* bn_var,
                   const float* bn_gamma, const float* bn_s0, const float* bn_s1, float bn_eps, float* dy_scratch, float* gx, float* aux,
                   const float* hbuf, const void* wp_bwd, const int* lens_dev, int T, int B, int H, int bf16, void* dgx_bf16,
                   const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws, size_t ws_bytes, void* stream);

/* ds2_rnn_bwd_bn with the BatchNorm's input given as bf16 (the centred operand of ds2_center_colstats; bn_mean = its delta; even pitch, H % 4 == 0) */
int ds2_rnn_bwd_bn_xbf16(ds2_rnn_ctx* ctx, int gates, const float* dyn, int lddyn, const void* bn_x_bf16, int ldx, const float* bn_mean,
                         const float* bn_var, const float* bn_gamma, const float* bn_s0, const float* bn_s1, float bn_eps, float* dy_scratch, float* gx,
                         float* aux, const float* hbuf, const void* wp_bwd, const int* lens_dev, int T, int B, int H, int bf16, void* dgx_bf16,
                         const void* gates_bf16, void* dhn_bf16, float* bias_part, void* ws, size_t ws_bytes, void* stream);

/* bias gradients of one recurrent layer from ds2_rnn_bwd_ex's bias_part (B,2,4,H): db_ih (2,G*H) [bias_ih_l0 | bias_ih_l0_reverse] and
 * db_hh (2,G*H); GRU: db_ih = [d r, d z, d n], db_hh = [d r, d z, d(hn)]; LSTM: both = [d i, d f, d g, d o]. */
int ds2_rnn_bias_grads(int gates, const float* bias_part, int B, int H, float* dbih, float* dbhh, void* stream);

/* ---- log-softmax + CTC loss + gradient ---------------------------------------------------------
 * out.float().log_softmax(2) + torch.nn.CTCLoss(reduction="sum") and their backward,
 * trainers/deepspeech_trainer.py:108-112, trainers/__main__.py:53.  blank = 0, zero_infinity = False. */
size_t ds2_ctc_workspace_bytes(int T, int B, int max_target_len);
int ds2_ctc_loss_f32(const float* logits, int ld, int T, int B, int C, const int* targets_dev, const int* tgt_off_dev,
                     const int* in_lens_dev, const int* tgt_lens_dev, int max_target_len, float* nll_dev, float* grad, int ldg,
                     float grad_scale, void* ws, size_t ws_bytes, void* stream);

/* out[0] = sum_b nll[b] / B on the device, fixed summation order: `loss = criterion(...) / inputs.size(0)`,
 * trainers/deepspeech_trainer.py:110-112 */
int ds2_ctc_batch_mean_f32(const float* nll_dev, int B, float* out_dev, void* stream);

/* softmax over the last dim (eval-mode InferenceBatchSoftmax, modules/blocks.py:59-64) */
int ds2_softmax_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int C, void* stream);

/* greedy CTC decode: per-frame arg-max (ties -> lowest class, torch.max's first-maximum rule), collapse repeats,
 * drop blanks; replaces GreedyDecoder.decode / convert_to_strings / process_string,
 * decoders/greedy_decoder.py:10-68 (a per-frame host loop with one .item() sync per frame).
 * probs (B,T,C) fp32 with element strides ld_b, ld_t (C contiguous); sizes_dev (B) int32 or NULL (= T frames).
 * Outputs: ids/offs (B,T) int32 — the first out_len[b] entries of row b are the kept labels and their frames. */
size_t ds2_greedy_decode_workspace_bytes(int B, int T);
int ds2_greedy_decode_f32(const float* probs, long long ld_b, long long ld_t, int B, int T, int C, const int* sizes_dev, int blank,
                          int* ids, int* offs, int* out_len, void* ws, size_t ws_bytes, void* stream);

/* conv1 in bf16 mode (Conv2d(1,32,(41,11),s=(2,2),p=(20,5)), deepspeech.py:61, forward + weight gradient; conv1 has no data
 * gradient).  ds2_conv1_gather_bf16 builds the two bf16 operand images from the spectrogram batch: XB (B,F,P) = the rows themselves as
 * bf16, XB[..][7 + s] = x[..][s] with zeros in front and behind (P = ds2_conv1_bf16_row_pitch(T)) for the forward — the 16 taps of an output
 * step are 16 consecutive samples of a row — and X16T (B,F,16,pad64(T)) time-contiguous per tap for the weight gradient (either may be NULL).
 * ds2_conv1_bf16_bytes(which = 0 packed weights | 1 XB | 2 X16T, B, F, T) sizes the buffers. */
int ds2_conv1_bf16_row_pitch(int T);
size_t ds2_conv1_bf16_bytes(int which, int B, int F, int T);
int ds2_conv1_pack_bf16(const float* w1, void* wp, void* stream);
int ds2_conv1_gather_bf16(const float* x, void* XB, void* X16T, int B, int F, int Tin, void* stream);
int ds2_conv1_fwd_bf16(const void* XB, const void* wp, const float* bias, const int* lens_dev, float* y1, int B, int F, int Tin,
                       void* stream);
/* ... with the BatchNorm2d statistics of y1 taken in the epilogue: stat_part = ds2_conv1_fwd_bf16_stat_blocks() x 32 x 2 floats of per-block
 * (sum, sum of squares) per channel; ds2_chanstats_from_partials turns them into mean / biased var (+ running stats) - no pass over y1 */
int ds2_conv1_fwd_bf16_stat_blocks(int B, int F, int Tin);
int ds2_conv1_fwd_bf16_stats(const void* XB, const void* wp, const float* bias, const int* lens_dev, float* y1, int B, int F, int Tin,
                             float* stat_part, void* stream);
size_t ds2_conv1_wgrad_bf16_workspace_bytes(int B, int Tin);
int ds2_conv1_wgrad_bf16(const void* X16T, const float* dy1, const int* lens_dev, float* dW1, int B, int F, int Tin, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- spectrogram front-end -------------------------------------------------------------
 * SpectrogramParser.parse_audio's arithmetic, data/parsers/spectrogram_parser.py:45-60, for a whole batch:
 * librosa.stft (centred frames, win_length = n_fft) -> |.| -> log1p -> optional (x - mean) / std(unbiased) per utterance,
 * written in the zero-padded (B,1,n_bins,T) layout of _collate_fn (functional.py:18-30).
 * audio (B, ld_audio) fp32 device waveforms, n_samples_dev (B) int32; basis (n_fft, 2*n_bins) = window-folded DFT basis
 * [w cos | -w sin interleaved per bin]; pad_mode 0 = zeros (librosa >= 0.10 default) / 1 = reflect; hop % 4 == 0. */
int ds2_spectrogram_frames(int n_samples, int hop);
size_t ds2_spectrogram_workspace_bytes(int B, int T, int n_fft, int hop);
int ds2_spectrogram_f32(const float* audio, long long ld_audio, const int* n_samples_dev, int B, int T, int n_fft, int hop,
                        const float* basis, int pad_mode, int normalize, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- optimizer ----------------------------------------------------------------------------------
 * torch.optim.AdamW.step over one flat parameter buffer, trainers/__main__.py:41-47. */
int ds2_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, void* stream);
/* the same update behind a device-side gate: apply_flag (device int, may be NULL = always) is read when the kernel runs, 0 = no-op */
int ds2_adamw_gated_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, const int* apply_flag, void* stream);
int ds2_scale_f32(float* x, long long n, float s, void* stream);
/* fp32 mode, conv2 on the bf16 matrix cores (three-term split products; engine.F32_CONV): r = x - float(bf16(x)) — the bf16 mode's cast / pack
 * entries applied to r give the "lo" operand — and out = a + b + c for the partial results (out may alias a; c may be NULL: out = a + b). */
int ds2_bf16_residual_f32(const float* x, float* r, long long n, void* stream);
int ds2_sum3_f32(const float* a, const float* b, const float* c, float* out, long long n, void* stream);
/* x[0..n) += v (int64): every BatchNorm's num_batches_tracked (torch.nn.BatchNorm*d.forward in training mode) in one launch */
int ds2_add_i64(long long* x, int n, long long v, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DS2HIP_H */
