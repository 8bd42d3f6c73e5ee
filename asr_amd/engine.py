"""Forward / backward orchestration of the DeepSpeech2 hot path over the libds2hip kernels.

Mirrors, block for block, what autograd replays for the reference's
`DeepSpeech.forward` (modules/deepspeech.py:130-149) — MaskConv conv stack (modules/blocks.py:42-56),
L x BatchRNN (modules/blocks.py:84-93), fc block (modules/deepspeech.py:103-109) — but as an explicit
schedule of HIP kernels with hand-derived backward passes, no torch compute ops on the path.

`W` is a mapping {reference state_dict key -> fp32 GPU tensor} plus the zero-copy concatenated views
`rnns.{l}.wih_cat (2GH, I)`, `rnns.{l}.whh_cat (2, GH, H)`, `rnns.{l}.bih_cat (2GH,)`,
`rnns.{l}.bhh_cat (2, GH)` provided by `asr_amd.params.FlatParams` (forward and reverse direction
weights are stored adjacently, so both directions run as ONE GEMM / ONE recurrence launch series).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib, ops

Tensor = torch.Tensor
import os as _os
# Off-critical-path work of backward (DS2_OVERLAP):
#   "2" (default)  the operand-preparation passes of the weight-gradient GEMMs (transposing casts of dGx / h / d(hn) / Xn, with the bias
#                  column sums they carry) run on a side stream: they are HBM-bound and light (17 KB of LDS, ~40 registers), so they
#                  co-reside with whatever runs on the compute stream — the compute-bound GEMMs, or the latency-bound persistent
#                  recurrence of the next layer, which leaves most of the memory system idle.  The weight-gradient GEMMs themselves stay
#                  on the compute stream, one layer late (behind the next layer's recurrence launch), so that nothing heavy ever waits
#                  beside a persistent launch that needs every CU.  c3: 31.7 -> 30.x ms per step.
#   "1"            everything off the critical path (GEMMs included) on the side stream (round 1's form: +2.6 ms with the step kernels,
#                  -0.6 ms with the persistent ones)
#   "0"            one stream
def _tune(name: str, default: str) -> str:
    """Tuning / A-B switches of the schedule (most of them experiments that lost and are kept runnable for the record): honoured only together
    with DS2_EXPERIMENTAL=1, so that a stray variable cannot change what the product runs.  The precision selectors DS2_F32_GEMM / _RNN / _CONV
    ("f32" = exact fp32 arithmetic instead of split-bf16 products) and the data-parallel schedule DS2_DP_MODE are user choices, not gated."""
    return _os.environ.get(name, default) if _os.environ.get("DS2_EXPERIMENTAL") == "1" else default


OVERLAP_MODE = _tune("DS2_OVERLAP", "2")
OVERLAP_WGRAD = OVERLAP_MODE == "1"
# Weight gradients of the recurrent layers in bf16 mode (DS2_WGRAD_TN, default 1): dW_ih / dW_hh as TN-form GEMMs straight from the row-major
# bf16 buffers the recurrence kernels write (dGx, d(hn), h) and the forward pass kept (Xn), bias gradients from the per-row sums of the
# backward recurrence — no cast / transpose pass at all.  Needs both recurrences of the layer to have run as persistent launches (they are
# the ones that write the bf16 copies); a layer whose recurrences did not keeps the transposing-cast passes below.  0: always those passes.
WGRAD_TN = _tune("DS2_WGRAD_TN", "1") != "0"
# bf16 mode: BatchNorm2d batch statistics from the conv forward epilogues; needs the rows-per-block conv2 kernel (DS2_CONV2_ROWS != 1)
CONV_STATS = _tune("DS2_CONV_STATS", "1") != "0" and _tune("DS2_CONV2_ROWS", "") != "1"
# bf16 training: the elementwise half of every BatchNorm1d backward is applied inside the K-split backward recurrence of the layer below
# (ops.rnn_bwd_bn: one more 4-byte load per pair and step instead of a pass over (T*B, H)); 0: a separate bn1d_bwd_apply pass as before
FUSE_BN_BWD = _tune("DS2_FUSE_BN_BWD", "1") != "0"
# bf16 training, weight gradients of a recurrent layer whose recurrences ran as persistent launches (DS2_WGRAD_SIDE):
#   "sk" (default) ONE launch of the 256 x 256 TN kernel over all products of the layer with a common split-K factor + ONE reduce launch
#                  (ops.gemm_bf16_tn_splitk_group: 192 tiles x 4 K slices = 768 equal work items = three full rounds of the chip), compute stream
#   "0"            round 3's schedule: three TN launches, each with its own split-K factor, slabs and reduce pass, compute stream
#   "1"            ONE launch of the co-resident grouped TN kernel (ops.gemm_bf16_tn_group: 4 waves x 128 registers, one workgroup per CU, no
#                  split-K) on the side stream, released when the compute stream reaches the backward recurrence of the layer BELOW.  The
#                  kernel is sized for exactly the registers / LDS the K-split recurrence leaves on a CU, so the recurrence's residency
#                  holds by construction (csrc/gemm_tn_group.h; 0 starved steps, gradients bit-identical to "main").  Layer 0's products run
#                  beside the conv-stack backward.  MEASURED AND NOT KEPT AS THE DEFAULT (profiles/r04_wgrad_side_ab.txt): the recurrence's
#                  per-step gather queues behind the co-resident kernel's operand DMA in the CU's own memory path and the recurrence loses
#                  1.65-2.0 us per time step (0.83-1.0 ms per layer) — more than the 0.86 ms of work it hides: 27.3 vs 25.9 ms per step.
#   "main"         the same grouped kernel on the compute stream right behind the layer's critical-path work (the one-stream schedule the
#                  side-stream results are compared with bit for bit: tests/test_gpu_round4.py); +0.5 ms per step against "0"
WGRAD_SIDE = _tune("DS2_WGRAD_SIDE", "sk")
# "sk" only: run a layer's grouped launch on the side stream beside the NEXT layer's persistent backward recurrence when that recurrence leaves
# at least DS2_WGRAD_IDLE_MIN_CUS compute units without a workgroup (B = 32 shapes: c2, c4); see _backward_rnn_deferred
WGRAD_IDLE = _tune("DS2_WGRAD_IDLE", "1") != "0"
WGRAD_IDLE_MIN_CUS = int(_tune("DS2_WGRAD_IDLE_MIN_CUS", "64"))
# fp32 mode, the dense input-to-hidden products (forward projection, dXn, dW_ih, dW_hh) of layers large enough for the 256 x 256 kernels
# (DS2_F32_GEMM): "split" (default) = every fp32 operand split into two bf16 terms (hi = bf16(x), lo = bf16(x - hi)) and the product taken
# as hi.hi + hi.lo + lo.hi on the bf16 matrix cores with fp32 accumulation — ~1e-5 of the fp32 product (the 2^-18 lo.lo term is dropped),
# two orders inside north_star's 1e-3, at 3 bf16 MFMA products instead of one at a sixteenth of the rate (ops.split_bf16);
# "f32" = the fp32-input MFMA kernels (csrc/gemm.hip) for every shape, as rounds 1-3.  Recurrences, BatchNorm, CTC, conv stack: fp32 as before.
F32_GEMM = _os.environ.get("DS2_F32_GEMM", "split")


# fp32 mode, recurrences (DS2_F32_RNN): "split" (default) = the persistent kernels with the moving operand (h_t / dGh_t) and W_hh as two bf16
# planes each (hi + lo) and three bf16 MFMAs per product where the shape fits (forward: GRU up to H = 1024, LSTM up to H = 768; backward:
# GRU / LSTM up to H = 768: csrc/rnn.hip, SP) — fp32-grade results (~1e-6 of the
# fp32 kernels) at a fifth of the fp32-MFMA time; "f32" = the fp32-MFMA kernels.  The library falls back to them by itself where the split
# kernel does not fit, and during a cooldown.
F32_RNN = _os.environ.get("DS2_F32_RNN", "split")


# fp32 mode, conv2's BACKWARD (DS2_F32_CONV): "split" (default) = the bf16 mode's conv2 input-gradient and weight-gradient kernels run three
# times each on split operands — a = hi + lo with hi = bf16(a), lo = bf16(a - hi) (ops.bf16_residual + the bf16 mode's own cast / pack
# kernels), product = hi.hi + lo.hi + hi.lo, the three fp32 results summed by ops.sum3_ (<= 2e-5 of fp64) — instead of the fp32-input MFMA
# kernels (csrc/conv.hip), which run at a sixteenth of the bf16 rate; "f32" = those kernels, as rounds 1-3.  The conv FORWARD stays on the
# fp32 kernels on purpose: measured, a 1e-6 perturbation of y2 flips a ~1e-6 fraction of the Hardtanh(0, 20) branches behind it, and a
# flipped element switches its whole gradient on or off, so the conv-stack gradients move by ~sqrt(1e-6) = 2e-3 — outside north_star's 1e-3
# (4.bias, which depends on the forward alone, showed exactly that; profiles/r04_f32_conv_ab.txt).  conv1 stays fp32 as well.
F32_CONV = _os.environ.get("DS2_F32_CONV", "split")
# fp32 mode, conv2's FORWARD (DS2_F32_CONV_FWD): "f32" (default) = the fp32-input MFMA kernel; "split6" = three bf16 pieces per operand
# (a = h + m + l, each the bf16 of what the pieces before it leave: 24 significant bits) and the six products h.h, h.m, m.h, m.m, h.l, l.h on
# the bf16 conv kernel.  Measured: 4.8e-7 of fp64 (the fp32-input kernel: 1.5e-6; three terms: 4.6e-6), c2 f32 33.80 -> 33.16 ms — and NOT the
# default: whichever fp32-grade forward runs, its rounding flips a few Hardtanh branches against the fp64 oracle and the conv-stack gradients
# move by the square root of that fraction; with this form one oracle case lands at 1.14e-3 against north_star's 1e-3
# (test_step_vs_oracle_larger[gru-128-2-33-90], conv.seq_module.0.weight; the fp32-input kernel passes all of them), and 0.6 ms on a
# secondary configuration does not buy a parity case (profiles/r04_f32_conv_ab.txt).
F32_CONV_FWD = _tune("DS2_F32_CONV_FWD", "f32")
# bf16 mode, conv2's weight gradient (DS2_CONV2_WGRAD): "nhwc" (default, round 5) = from the channels-last operands conv2's forward / data
# gradient already take (time-major LDS images filled by DMA; no padded copies of a1 / dY2 are written any more); "pad" = the round-2 kernel
# on zero-padded (B,32,D,Tp) copies (eight pre-shifted dY copies in LDS).  A/B: profiles/r05_conv_ab.txt.
CONV2_WGRAD = _tune("DS2_CONV2_WGRAD", "nhwc")
# bf16 training, the x-projections of a recurrent layer (DS2_GX_BF16, default 0 — an experiment of round 6 that was measured and NOT taken): =1
# makes the projection GEMM round its fp32 accumulators (+ bias) to bf16 at the store and the forward recurrence read that — 394 MB written + read
# per c3 layer instead of 788 (ops.gemm_bf16_nt_obf16).  The projection gets as fast as dX (395 -> 367 us) but the recurrence pays +0.05 us per time
# step for the 2-byte operand: net -0.06 ms per step for one more rounding of every gate pre-activation (profiles/r06_experiments.txt).
GX_BF16 = _tune("DS2_GX_BF16", "0") != "0"
# the workspace of a recurrence call (exchange buffers of a persistent launch: every byte 0xff) is armed AHEAD of the GEMM in front of the
# recurrence (ops.rnn_ws) instead of by a fill launched between that GEMM and the recurrence (DS2_WS_PREARM=0: as rounds 1-5)
WS_PREARM = _tune("DS2_WS_PREARM", "1") != "0"
# bf16 training, BatchNorm1d of the recurrent layers 1 .. L-1 folded into their input projections (DS2_BN_FOLD): the direction sum + statistics
# pass of the layer in front writes ONLY the centred bf16 operand yc = y - m0 (ops.center_colstats: column means from the per-tile sums of h the
# forward recurrence emits, statistics of the centred values) and the normalisation goes into the weights, gx = yc (W_ih diag(s))^T + (b + W_ih c)
# (ops.wih_fold); backward: dW_ih = (dGx^T yc) diag(s) + db (x) c, BatchNorm backward on (yc, delta, var).  Neither y (fp32) nor BN(y) exists
# any more: 327 MB moved per layer in forward instead of 589, 196 instead of 262 for the backward sums.  0: the separate passes of rounds 1-5.
BN_FOLD = _tune("DS2_BN_FOLD", "1") != "0"
BN_FOLD_IDLE = _tune("DS2_BN_FOLD_IDLE", "1") != "0"      # (A/B: 0 = no fold for shapes under the idle-CU weight-gradient schedule, see forward())


def _f32_split_ok(M: int, N: int, K: int, H: int = 8) -> bool:
    """fp32 mode: do this layer's input-to-hidden products run as three-term split-bf16 GEMMs?  H: the layer's hidden size — backward's dW_hh
    problems have N = H and column offsets d * H, d * G * H, which the grouped TN launch wants 16-byte aligned (H % 8 == 0); hidden sizes
    with H % 8 == 4 (accepted by the recurrence entry points) keep the fp32-MFMA GEMM path."""
    return F32_GEMM == "split" and M >= 512 and N >= 256 and K >= 256 and K % 8 == 0 and N % 8 == 0 and H % 8 == 0


_BWD_PERSISTENT = {}      # (id of the recurrence context in use, gates, H, B) -> did the last backward recurrence of this shape run as a persistent launch?
_SIDE = {}


def _idle_cus_beside_bwd_recurrence(device, B: int, H: int) -> int:
    """compute units without a workgroup while a persistent backward recurrence runs: it is one workgroup per (direction, 16-row batch tile,
    32-unit slice) (csrc/rnn_bwd_ksplit.h, rnn_bwd_persistent_kernel)"""
    return torch.cuda.get_device_properties(device).multi_processor_count - 2 * ((B + 15) // 16) * ((H + 31) // 32)


def _wgrad_idle_schedule(device, B: int, H: int) -> bool:
    """does backward launch a layer's weight-gradient GEMMs on the side stream beside the NEXT layer's recurrence (the idle-CU schedule of
    _backward_rnn_deferred: B = 32 shapes whose recurrence leaves 64 ... 127 CUs without a workgroup — c4)?"""
    n_idle = _idle_cus_beside_bwd_recurrence(device, B, H)
    return WGRAD_IDLE_MIN_CUS <= n_idle < torch.cuda.get_device_properties(device).multi_processor_count // 2      # (the shape alone decides)


def _side_stream(device):
    import threading
    key = (device.type, device.index, threading.get_ident())      # one side stream per driving thread: two threads never share one
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


@dataclass
class BnGrad:
    """Gradient wrt a BatchNorm1d OUTPUT together with what its elementwise backward needs: consumed by ops.rnn_bwd_bn of the layer below."""
    dyn: Tensor          # (M, H) gradient wrt the BatchNorm output
    x: Tensor            # (M, H) the BatchNorm input (= the layer below's y)
    mean: Tensor
    var: Tensor
    gamma: Tensor
    sums: tuple          # two (H,) vectors: column sums of dyn and of dyn * xhat


def _bn_backward(dxn: Tensor, x: Tensor, mean, var, gamma, dgamma, dbeta, fuse: bool, private_sums: bool):
    """BatchNorm1d backward in front of a recurrent layer's backward: the materialised gradient, or (fuse) the column sums only + a BnGrad.
    private_sums: the gradient buffers may be all-reduced (asynchronously, in place) before the recurrence below has read the sums."""
    xbf = x.dtype == torch.bfloat16                      # BN_FOLD: x = the centred bf16 operand, mean = its delta
    if not fuse:
        return (ops.bn1d_bwd_xbf if xbf else ops.bn1d_bwd)(dxn, x, mean, var, gamma, dgamma, dbeta)
    sums_fn = ops.bn1d_bwd_sums_xbf if xbf else ops.bn1d_bwd_sums
    if private_sums:
        sums = sums_fn(dxn, x, mean, var, gamma)
        dbeta.copy_(sums[0])
        dgamma.copy_(sums[1])
    else:
        sums = sums_fn(dxn, x, mean, var, gamma, out=(dbeta, dgamma))
    return BnGrad(dxn, x, mean, var, gamma, sums)


@dataclass
class ModelCfg:
    rnn: str            # "gru" | "lstm"
    hidden: int
    layers: int
    classes: int
    freq: int = 161
    precision: str = "fp32"   # "fp32": everything fp32 (parity path) | "bf16": bf16 MFMA operands, fp32 accumulate/state

    @property
    def gates(self) -> int:
        return 3 if self.rnn == "gru" else 4


@dataclass
class LayerCtx:
    xin: Optional[Tensor] = None     # (M, I) raw layer input (previous y) — BN backward needs it
    xn: Optional[Tensor] = None      # (M, I) GEMM A operand (BN output, or the raw input for layer 0)
    gx: Optional[Tensor] = None      # (M, 2GH) gates -> overwritten by dGx in backward
    hbuf: Optional[Tensor] = None    # (M, 2H)
    aux: Optional[Tensor] = None     # (M, 2H)
    mean: Optional[Tensor] = None    # BN batch stats of this layer's INPUT (layers >= 1)
    var: Optional[Tensor] = None
    wpb: Optional[Tensor] = None     # W_hh^T in fragment order for the backward recurrence
    rec: Optional[Tensor] = None     # bf16 training: packed saved-gate records (M, 2H, 4) instead of gates in gx
    h_bf: Optional[Tensor] = None    # bf16 training, persistent forward recurrence: (M, 2H) bf16 copy of hbuf (operand of the TN-form dW_hh)
    wihT: Optional[Tensor] = None    # bf16 training: bf16 W_ih^T (I, pad8(2GH)), the B operand of dX (cast with the forward projection's operand)
    xs: Optional[Tensor] = None      # fp32 mode, split-bf16 GEMMs: (M, 3 I) [hi | hi | lo] split copy of xn (xn itself is then not kept)
    gshape: tuple = ()               # (M, 2GH) when gx itself was released
    fold: tuple = ()                 # BN_FOLD: (colscale s, colshift c) of this layer's folded BatchNorm; xin / xn are then the centred bf16 operand


@dataclass
class Ctx:
    B: int = 0
    T: int = 0
    D1: int = 0
    D2: int = 0
    x: Optional[Tensor] = None
    lens_dev: Optional[Tensor] = None
    y1: Optional[Tensor] = None
    a1: Optional[Tensor] = None      # fp32 Hardtanh(BN(conv1)) (fp32 mode; bf16 mode keeps only a1p)
    a1p: Optional[Tensor] = None     # bf16 mode: the bf16 operand of conv2's weight gradient (channels-last copy; DS2_CONV2_WGRAD=pad: zero-padded rows)
    y2: Optional[Tensor] = None
    st1: tuple = ()
    st2: tuple = ()
    packs: tuple = ()
    x16t: Optional[Tensor] = None      # bf16 mode: conv1's time-contiguous operand image (kept for its weight gradient)
    layers: List[LayerCtx] = field(default_factory=list)
    y_last: Optional[Tensor] = None
    fc_xn: Optional[Tensor] = None
    fc_stats: tuple = ()
    side_used: bool = False          # backward put weight-gradient work on the side stream: the compute stream joins it at the end


def forward(W: Dict[str, Tensor], cfg: ModelCfg, x: Tensor, lens_dev: Tensor, training: bool, save: bool = True, debug_acts: bool = False):
    """x (B,1,F,Tin) fp32 GPU; lens_dev int32 (B,) GPU = output frame counts.  Returns (logits (T,B,C), ctx)."""
    if not x.is_cuda:
        raise _lib.DS2LibraryError("asr_amd.engine.forward needs a GPU tensor: the MI355X kernels are the only implementation "
                                   "(the CPU restatement lives in oracle/ and is test-only)")
    x = x.contiguous().float()
    B, _, F, Tin = x.shape
    D1, D2, T = _lib.conv_dims(F, Tin)
    G, H, L = cfg.gates, cfg.hidden, cfg.layers
    M = T * B
    ctx = Ctx(B=B, T=T, D1=D1, D2=D2, x=x, lens_dev=lens_dev)
    cp = "conv.seq_module."

    def run(name):
        return (W[name + ".running_mean"], W[name + ".running_var"]) if training else (None, None)

    # ---- conv stack -----------------------------------------------------------------------------
    if cfg.precision != "bf16":                                  # (the bf16 mode has its own operand packs: conv1_pack_bf16 / conv2_pack_bf16)
        wpk1, wpk2, wpk2d = ops.conv_pack(W[cp + "0.weight"], W[cp + "3.weight"])
        ctx.packs = (wpk2d,)
    if cfg.precision == "bf16":
        # conv1 on the bf16 matrix cores: operand images gathered once from the spectrogram (the time-contiguous one is kept
        # for the weight gradient)
        X16, ctx.x16t = ops.conv1_gather_bf16(x, want_fwd=True, want_wgrad=save)
        # training: the BatchNorm2d statistics come out of the conv epilogues (per-block channel sums), not from another pass over y
        y1 = ops.conv1_fwd_bf16(X16, ops.conv1_pack_bf16(W[cp + "0.weight"]), W[cp + "0.bias"], lens_dev, Tin, stats=training and CONV_STATS)
        st_part1 = None
        if training and CONV_STATS:
            y1, st_part1 = y1
        del X16
    else:
        y1 = ops.conv1_fwd(x, wpk1, W[cp + "0.bias"], lens_dev)
        st_part1 = None
    if training and st_part1 is not None:
        m1, v1 = ops.chanstats_from_partials(st_part1, B * D1 * T, *run(cp + "1"))
    elif training:
        m1, v1 = ops.bn2d_stats(y1, *run(cp + "1"))
    else:
        m1, v1 = W[cp + "1.running_mean"], W[cp + "1.running_var"]
    if cfg.precision == "bf16":
        # one pass over y1 emits conv2's channels-last operand and (when backward follows) the zero-padded operand of conv2's weight
        # gradient; the fp32 activation itself has no consumer in this mode (debug_acts keeps it for the tests that inspect it)
        a1, a1p, a1n = ops.bn2d_act_fwd_fused(y1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"], want_f32=debug_acts,
                                              want_pad=save and CONV2_WGRAD != "nhwc", want_nhwc=True)
        cwf, cwd0, cwd1 = ops.conv2_pack_bf16(W[cp + "3.weight"])
        ctx.packs = (None, cwd0, cwd1)
        y2 = ops.conv2_fwd_bf16(a1n, cwf, W[cp + "3.bias"], lens_dev, stats=training and CONV_STATS)
        st_part2 = None
        if training and CONV_STATS:
            y2, st_part2 = y2
        if save and CONV2_WGRAD == "nhwc":
            a1p = a1n                                            # the channels-last copy IS the weight-gradient operand (kept for backward)
        del a1n
    elif F32_CONV_FWD == "split6":
        a1, a1p = ops.bn2d_act_fwd(y1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"]), None      # (the same a1 as the fp32-kernel path, bit for bit)
        ah = ops.nhwc_bf16(a1)
        r = ops.bf16_residual(a1)
        am = ops.nhwc_bf16(r)
        al = ops.nhwc_bf16(ops.bf16_residual(r))
        del r
        w2 = W[cp + "3.weight"]
        wr = ops.bf16_residual(w2)
        wh, wm, wl = ops.conv2_pack_bf16(w2)[0], ops.conv2_pack_bf16(wr)[0], ops.conv2_pack_bf16(ops.bf16_residual(wr))[0]
        zb = torch.zeros_like(W[cp + "3.bias"])
        # smallest terms first: (m.m + h.l + l.h), then + h.m + m.h, then h.h (+ bias)
        y2 = ops.conv2_fwd_bf16(am, wm, zb, lens_dev)
        ops.sum3_(y2, ops.conv2_fwd_bf16(ah, wl, zb, lens_dev), ops.conv2_fwd_bf16(al, wh, zb, lens_dev))
        t2 = ops.conv2_fwd_bf16(ah, wm, zb, lens_dev)
        ops.sum3_(y2, t2, ops.conv2_fwd_bf16(am, wh, zb, lens_dev))
        del t2
        y2 = ops.sum3_(ops.conv2_fwd_bf16(ah, wh, W[cp + "3.bias"], lens_dev), y2)
        del ah, am, al
        st_part2 = None
    else:
        a1, a1p = ops.bn2d_act_fwd(y1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"]), None
        y2 = ops.conv2_fwd(a1, wpk2, W[cp + "3.bias"], lens_dev)
        st_part2 = None
    if training and st_part2 is not None:
        m2, v2 = ops.chanstats_from_partials(st_part2, B * D2 * T, *run(cp + "4"))
    elif training:
        m2, v2 = ops.bn2d_stats(y2, *run(cp + "4"))
    else:
        m2, v2 = W[cp + "4.running_mean"], W[cp + "4.running_var"]
    xn0 = None
    if cfg.precision == "bf16":
        # BN + Hardtanh + mask + collapse + cast in one pass: layer 0 takes the bf16 operand; nobody needs the fp32 form
        # (row pitch rounded up to the GEMMs' 64-deep k-tile: 1312 -> 1344 zero-padded features put layer 0's projection on the four-wave kernel)
        xin, xn0 = ops.bn2d_act_collapse(y2, lens_dev, m2, v2, W[cp + "4.weight"], W[cp + "4.bias"], want_f32=debug_acts, pad_to=64)
    else:
        a2 = ops.bn2d_act_fwd(y2, lens_dev, m2, v2, W[cp + "4.weight"], W[cp + "4.bias"])
        xin = ops.transpose_bft(a2, B, 32 * D2, T, to_tbf=True).view(M, 32 * D2)   # (T*B, 1312), feature = c*D2 + d
        del a2
    if save:
        ctx.y1, ctx.a1, ctx.a1p, ctx.y2, ctx.st1, ctx.st2 = y1, a1, a1p, y2, (m1, v1), (m2, v2)

    # ---- recurrent stack ------------------------------------------------------------------------
    mean = var = delta = None
    for l in range(L):
        lc = LayerCtx()
        bf = cfg.precision == "bf16"
        rmode = 1 if bf else (2 if (F32_RNN == "split" and H % 32 == 0) else 0)
        ws_f = ops.rnn_ws("fwd", G, B, H, rmode, x.device) if WS_PREARM else None
        folded = l > 0 and xin.dtype == torch.bfloat16       # BN_FOLD: the layer in front left the centred bf16 operand (+ mean, var, delta)
        if folded:
            bp = f"rnns.{l}.batch_norm.module."
            xn = xin
            w_bf, bias_eff, colscale, colshift = ops.wih_fold(W[f"rnns.{l}.wih_cat"], W[f"rnns.{l}.bih_cat"], var, W[bp + "weight"], W[bp + "bias"], delta,
                                                              ld=xn.shape[1])
            lc.wihT = ops.cast_transpose_bf16(W[f"rnns.{l}.wih_cat"])          # (un-scaled: dX is the gradient wrt BN's OUTPUT, as before)
            lc.mean, lc.var, lc.fold = delta, var, (colscale, colshift)        # (the mean OF xin)
            gx = ops.gemm_bf16_nt(xn, w_bf, bias=bias_eff)
            del w_bf
        elif l > 0:
            bp = f"rnns.{l}.batch_norm.module."
            if not training:
                mean, var = W[bp + "running_mean"], W[bp + "running_var"]
            # bf16 mode: the normalised input exists only as the bf16 GEMM operand (kept for dW_ih in backward)
            xn = (ops.bn1d_apply_bf16 if cfg.precision == "bf16" else ops.bn1d_apply)(xin, mean, var, W[bp + "weight"], W[bp + "bias"])
            lc.mean, lc.var = mean, var
        else:
            xn = xn0 if cfg.precision == "bf16" else xin
        if folded:
            pass
        elif cfg.precision == "bf16":
            if save:
                # training: ONE read of W_ih gives the row-major bf16 operand of this projection and the transposed one of dX (kept for backward)
                w_bf, lc.wihT = ops.cast_bf16_both(W[f"rnns.{l}.wih_cat"], ld_r=xn.shape[1])
            else:
                w_bf = ops.cast_bf16(W[f"rnns.{l}.wih_cat"], ld=xn.shape[1])
            gx = None
            if GX_BF16 and save and B % 8 == 0:                  # (the conditions of `pack` below: the only consumer of a bf16 gx)
                gx = ops.gemm_bf16_nt_obf16(xn, w_bf, bias=W[f"rnns.{l}.bih_cat"])
            if gx is None:
                gx = ops.gemm_bf16_nt(xn, w_bf, bias=W[f"rnns.{l}.bih_cat"])
            del w_bf
        elif _f32_split_ok(M, 2 * G * H, xn.shape[1], H):
            # three-term split-bf16 product as ONE NT GEMM over a reduction index 3 I long: [hi | hi | lo] x [hi | lo | hi]^T
            # (rows padded with zeros to a multiple of the TN kernels' 64-deep k-tile: as the K-row-major operand of dW_ih in backward the split
            #  copy then has a reduction length the four-wave kernel takes — T * B = 16032 at c2, 24032 at c4)
            lc.xs = ops.split_bf16(xn, 0, pad_rows=64 if save else 0)
            gx = ops.gemm_bf16_nt(lc.xs[:M], ops.split_bf16(W[f"rnns.{l}.wih_cat"], 1), bias=W[f"rnns.{l}.bih_cat"])
            if save:
                xn = None                                        # backward takes dW_ih from the split copy (6 bytes per element instead of 4 + 6)
        else:
            gx = ops.gemm(xn, W[f"rnns.{l}.wih_cat"], transB=True, bias=W[f"rnns.{l}.bih_cat"])  # (M, 2GH)
        wpf, wpb = ops.rnn_pack(G, W[f"rnns.{l}.whh_cat"], bf16=rmode)
        # bf16 training (B % 8 == 0, the condition of the bf16 dGx path in backward): the saved gates are ONE packed bf16 record per
        # hidden unit; the fp32 x-projection buffer is then dead after the recurrence
        pack = bf and save and B % 8 == 0
        if pack:
            # (not for shapes whose backward recurrence is known to run one launch per step - LSTM H = 1280: nobody would read the copy)
            h_bf = (torch.empty(M, 2 * H, dtype=torch.bfloat16, device=x.device)
                    if (WGRAD_TN and OVERLAP_MODE == "2" and T > 1 and _BWD_PERSISTENT.get((ops.rnn_ctx_key(x.device), G, H, B), True)) else None)
            # BN_FOLD: the layer behind this one folds its BatchNorm — this recurrence emits the per-tile column sums of h it needs
            hsum = (torch.empty(2, (B + 15) // 16, H, dtype=torch.float32, device=x.device)
                    if (BN_FOLD and training and l + 1 < L and H % 8 == 0 and OVERLAP_MODE == "2" and T > 1
                        # (c4 — LSTM 1280 under the idle-CU schedule — first LOST 1.1 ms with the fold: not the epilogue beside the recurrence,
                        #  but a branch the bf16 BatchNorm input had put into the LSTM instance's per-step operand fetch; with ONE load either
                        #  way the fold is worth -0.2 ms there too, profiles/r06_experiments.txt)
                        and (BN_FOLD_IDLE or not _wgrad_idle_schedule(x.device, B, H))) else None)
            hbuf, aux, rec = ops.rnn_fwd(G, gx, wpf, W[f"rnns.{l}.bhh_cat"], lens_dev, T, B, H, bf16=True, packed_gates=True, h_bf16=h_bf, ws=ws_f, hsum=hsum)
            if hsum is not None and not (ops.rnn_last_path(x.device) & 1):
                hsum = None                                      # (only a persistent launch writes it)
            gx = None
            lc.rec, lc.gshape = rec, (M, 2 * G * H)
            if h_bf is not None and (ops.rnn_last_path(x.device) & 1):
                lc.h_bf = h_bf                                   # (only a persistent launch writes it)
        else:
            hsum = None
            hbuf, aux = ops.rnn_fwd(G, gx, wpf, W[f"rnns.{l}.bhh_cat"], lens_dev, T, B, H, bf16=rmode, ws=ws_f)
        lc.wpb = wpb
        nxt = f"rnns.{l + 1}.batch_norm.module" if l + 1 < L else "fc.0.module.0"
        if hsum is not None:
            y, mean, var, delta = ops.center_colstats(hbuf[:, :H], hbuf[:, H:], hsum, *run(nxt))     # y: the CENTRED bf16 operand
        else:
            y, mean, var = ops.add_colstats(hbuf[:, :H], hbuf[:, H:], *run(nxt))
        if save:
            lc.xin, lc.xn, lc.gx, lc.hbuf, lc.aux = xin, xn, gx, hbuf, aux
            ctx.layers.append(lc)
        xin = y

    # ---- fc block -------------------------------------------------------------------------------
    fp = "fc.0.module."
    if not training:
        mean, var = W[fp + "0.running_mean"], W[fp + "0.running_var"]
    xn = ops.bn1d_apply(xin, mean, var, W[fp + "0.weight"], W[fp + "0.bias"])
    logits = ops.gemm(xn, W[fp + "1.weight"], transB=True)                                        # (M, C)
    if save:
        ctx.y_last, ctx.fc_xn, ctx.fc_stats = xin, xn, (mean, var)
    if training:
        if "_bn_counters" in W:
            ops.add_i64(W["_bn_counters"], 1)          # all num_batches_tracked (views of one buffer: asr_amd/params.py) in one launch
        else:
            for name in ([cp + "1", cp + "4", fp + "0"] + [f"rnns.{l}.batch_norm.module" for l in range(1, L)]):
                W[name + ".num_batches_tracked"] += 1
    return logits.view(T, B, cfg.classes), ctx


def _backward_rnn_deferred(W, Gr, cfg: ModelCfg, ctx: Ctx, dy, done, private: bool, serial_buckets: bool = False):
    """Backward of the recurrent stack in the bf16 training mode (packed gate records, bf16 dGx), DS2_OVERLAP=2 schedule: per layer
        compute stream:  recurrence(l) | weight-gradient GEMMs of layer l+1 | dXn(l) = dGx W_ih | BatchNorm1d backward(l)
        side stream:     [from the start of recurrence(l)] transposing casts of dGx(l+1) (+ db_ih), d(hn)(l+1) (+ db_hn), h(l+1), Xn(l+1)
    Same kernels, same operands, same results as the one-stream schedule: only the streams and the issue order differ.
    Layers whose two recurrences ran as persistent launches (the default for c2 / c3 / c5 shapes) need none of those passes: the kernels
    wrote bf16 h / d(hn) / dGx and the bias partial sums themselves and the weight gradients are TN-form GEMMs on exactly those buffers
    (`weight_gradients_tn`, DS2_WGRAD_TN) — one stream, nothing to overlap."""
    G, H, L = cfg.gates, cfg.hidden, cfg.layers
    B, T = ctx.B, ctx.T
    M = T * B
    lens_dev = ctx.lens_dev
    main = torch.cuda.current_stream()
    dev = ctx.lens_dev.device
    side = _side_stream(dev)
    pending = None
    report = done

    def done(name):
        """A reducer may release MORE than the named bucket from the calling stream (the "conv" schedule holds fc and every recurrent layer
        until rnns.0 and then records ONE "gradients final" event on the stream it is called from): weight gradients that earlier layers
        left on the side stream must be ordered before that event — the calling stream joins the side stream first."""
        if name == "rnns.0" and getattr(ctx, "side_used", False) and not serial_buckets:
            torch.cuda.current_stream().wait_stream(side)
        report(name)

    def fold_epilogue(l):
        """BN_FOLD: the projection's operand was the centred yc, not BN(y): dW_ih = (dGx^T yc) diag(s) + db_ih (x) c, applied to the product in
        place on the stream that formed it (db_ih is final before any weight-gradient product of its layer is launched)"""
        f = fold_of.get(l)
        if f:
            ops.scale_rank1_(Gr[f"rnns.{l}.wih_cat"], f[0], Gr[f"rnns.{l}.bih_cat"].view(-1), f[1])

    def weight_gradients(p):
        """layer p's dW_hh / dW_ih GEMMs on the compute stream, behind the event of its operand passes"""
        l, dgxT, hT, auxT, xnT, ready, hold = p
        main.wait_event(ready)
        dwhh = Gr[f"rnns.{l}.whh_cat"]                                                            # (2, GH, H)
        rows = 2 * H if G == 3 else 4 * H
        # both directions per launch: direction 0 pairs rows t of dGh with h[t-1], direction 1 rows t with h[t+1] (a column offset of B)
        ka = (slice(B, M), slice(0, M - B))
        kb = (slice(0, M - B), slice(B, M))
        ops.gemm_bf16_nt_pair(dgxT[0:rows, ka[0]], dgxT[G * H:G * H + rows, ka[1]], hT[0:H, kb[0]], hT[H:2 * H, kb[1]], dwhh[:, :rows])
        if G == 3:
            ops.gemm_bf16_nt_pair(auxT[0:H, ka[0]], auxT[H:2 * H, ka[1]], hT[0:H, kb[0]], hT[H:2 * H, kb[1]], dwhh[:, 2 * H:])
        ops.gemm_bf16_nt(dgxT, xnT, out=Gr[f"rnns.{l}.wih_cat"])
        fold_epilogue(l)
        for t in (dgxT, hT, auxT, xnT) + hold:                   # allocated / last used on the other stream: tell the caching allocator
            if t is not None:
                t.record_stream(main)
        done(f"rnns.{l}")

    def weight_gradients_tn(l, dgx_bf, dhn_bf, h_bf, xn, bias_part):
        """layer l's bias / weight gradients from the row-major bf16 buffers: TN-form GEMMs (reduction index T*B on the rows of both
        operands), the time shift of dW_hh is a ROW offset of B"""
        ops.rnn_bias_grads(G, bias_part, Gr[f"rnns.{l}.bih_cat"], Gr[f"rnns.{l}.bhh_cat"])           # sums over the batch rows
        dwhh = Gr[f"rnns.{l}.whh_cat"]                                                            # (2, GH, H)
        rows = 2 * H if G == 3 else 4 * H
        # direction 0 pairs dGh[t] with h[t-1], direction 1 dGh[t] with h[t+1]
        ra, rb = (slice(B, M), slice(0, M - B)), (slice(0, M - B), slice(B, M))
        ops.gemm_bf16_tn_pair(dgx_bf[ra[0], 0:rows], dgx_bf[ra[1], G * H:G * H + rows], h_bf[rb[0], 0:H], h_bf[rb[1], H:2 * H], dwhh[:, :rows])
        if G == 3:
            ops.gemm_bf16_tn_pair(dhn_bf[ra[0], 0:H], dhn_bf[ra[1], H:2 * H], h_bf[rb[0], 0:H], h_bf[rb[1], H:2 * H], dwhh[:, 2 * H:])
        dwih = Gr[f"rnns.{l}.wih_cat"]                                                            # (2GH, I)
        I = dwih.shape[1]
        if I % 8 == 0:
            ops.gemm_bf16_tn(dgx_bf, xn[:, :I], out=dwih)
        else:                                                    # xn is zero-padded to a multiple of 8 columns
            dwih.copy_(ops.gemm_bf16_tn(dgx_bf, xn)[:, :I])
        fold_epilogue(l)
        done(f"rnns.{l}")

    def weight_gradients_group(l, dgx_bf, dhn_bf, h_bf, xn, on_side, start):
        """the products of weight_gradients_tn as ONE launch of the co-resident grouped kernel; on_side: on the side stream, not before
        `start` (an event of the compute stream) — the bucket is reported from that stream, behind the launch"""
        dwhh, dwih = Gr[f"rnns.{l}.whh_cat"], Gr[f"rnns.{l}.wih_cat"]
        rows = 2 * H if G == 3 else 4 * H
        probs = [(dgx_bf, xn[:, :dwih.shape[1]], dwih),
                 (dgx_bf[B:M, 0:rows], h_bf[0:M - B, 0:H], dwhh[0, :rows]),
                 (dgx_bf[0:M - B, G * H:G * H + rows], h_bf[B:M, H:2 * H], dwhh[1, :rows])]
        if G == 3:
            probs += [(dhn_bf[B:M, 0:H], h_bf[0:M - B, 0:H], dwhh[0, 2 * H:]), (dhn_bf[0:M - B, H:2 * H], h_bf[B:M, H:2 * H], dwhh[1, 2 * H:])]
        f = fold_of.get(l)
        if WGRAD_SIDE == "sk" and f:
            # BN_FOLD: the epilogue of dW_ih (problem 0) rides in the reduce launch of the grouped split-K product
            def launch(pr):
                ops.gemm_bf16_tn_splitk_group(pr, epilogue=(0, f[0], Gr[f"rnns.{l}.bih_cat"].view(-1), f[1]))
        elif WGRAD_SIDE == "sk":
            launch = ops.gemm_bf16_tn_splitk_group
        else:
            def launch(pr):
                ops.gemm_bf16_tn_group(pr)
                fold_epilogue(l)
        if not on_side:
            launch(probs)
            done(f"rnns.{l}")
            return
        with torch.cuda.stream(side):
            side.wait_event(start)
            launch(probs)
            if not serial_buckets:
                done(f"rnns.{l}")                                # (a reducer records its "gradients final" event on the current = side stream)
        if serial_buckets:                                       # the "serial" data-parallel schedule orders every collective INTO the compute stream
            main.wait_stream(side)
            done(f"rnns.{l}")
        for t in (dgx_bf, dhn_bf, h_bf, xn):
            if t is not None:
                t.record_stream(side)
        ctx.side_used = True

    # the side stream only pays beside a PERSISTENT recurrence (its resident workgroups leave registers and LDS for light kernels); beside
    # one-launch-per-step kernels (shapes whose W_hh^T slice does not fit: LSTM H = 1280) co-running passes delay every launch
    # (c4: 94.6 -> 99.5 ms per step), so there the passes stay on the compute stream.  What the library did for this shape is known from
    # the previous call (ds2_rnn_last_path); the first call of a shape assumes persistent.
    shape_key = (ops.rnn_ctx_key(dev), G, H, B)

    def operand_passes(l, lc_t, dgx_bf, start):
        """side stream, not before `start`: transposing casts of layer l's dGx (+ db_ih), d(hn) (+ db_hn), h and Xn"""
        aux, hbuf, xn = lc_t
        on_side = _BWD_PERSISTENT.get(shape_key, True)
        with torch.cuda.stream(side if on_side else main):
            if on_side:
                side.wait_event(start)
            dgxT, _ = ops.transpose_bf16(dgx_bf, colsum=Gr[f"rnns.{l}.bih_cat"].view(-1))          # + db_ih = column sums of dGx
            dbhh = Gr[f"rnns.{l}.bhh_cat"]                                                        # (2, GH)
            dbhh.copy_(Gr[f"rnns.{l}.bih_cat"].view(2, G * H))
            auxT = None
            if G == 3:
                dbn = torch.empty(2 * H, dtype=torch.float32, device=dev)
                auxT = ops.cast_transpose_bf16(aux, colsum=dbn)                                   # + d(b_hn) = column sums of d(hn)
                dbhh[:, 2 * H:] = dbn.view(2, H)
            hT = ops.cast_transpose_bf16(hbuf)                                                    # (2H, M)
            xnT = ops.transpose_bf16(xn[:, :W[f"rnns.{l}.wih_cat"].shape[1]])                     # xn is bf16 (M, pad8(I)) in this mode
            ready = torch.cuda.Event()
            ready.record(side if on_side else main)
        if on_side:
            for t in (dgx_bf, aux, hbuf, xn):                    # read on the side stream
                t.record_stream(side)
        return (l, dgxT, hT, auxT, xnT, ready, (dgx_bf,))

    queued = None                                                # layer whose operand passes wait for the next recurrence launch
    queued_tn = None
    queued_side = None                                           # layer whose grouped weight-gradient launch waits for the next recurrence
    queued_idle = None                                           # the same for the idle-CU schedule (DS2_WGRAD_IDLE)
    group_ok = WGRAD_SIDE != "0" and W[f"rnns.0.wih_cat"].shape[1] % 8 == 0
    side_ok = group_ok and WGRAD_SIDE == "1" and ops.wgrad_fits_beside_bwd_recurrence(G, H)
    # "sk" + DS2_WGRAD_IDLE: a persistent backward recurrence is one workgroup per (direction, 16-row batch tile, 32-unit slice) — at B = 32
    # that is 96 (H = 768) or 160 (H = 1280) of the 256 CUs.  The grouped split-K launch of the layer above then runs on the side stream BEHIND
    # the recurrence launch: its 256 x 256 workgroups (2 x 256 registers per SIMD lane) cannot share a CU with a recurrence workgroup, so they
    # take exactly the CUs the recurrence leaves idle — no CU's memory path is shared (what sank the co-resident kernel, DS2_WGRAD_SIDE=1)
    # Measured (profiles/r04_wgrad_idle_ab.txt): c4 (LSTM 1280, 160 of 256 CUs in the recurrence) 55.6 -> 51.7 ms — the recurrence pays 3.93 ->
    # 4.52 us per time step for 1.07 ms of hidden GEMM per layer; c2 (GRU 768, 96 CUs in the recurrence) 15.08 -> 15.07: with 160 CUs the GEMM
    # is over in 0.4 ms and doubles the step time meanwhile; at an even 128 / 128 split (GRU 1024 at B = 32: what a 32-row backward kernel
    # would create at c3's B = 64) the recurrence pays MORE than is hidden — 5.61 -> 7.63 ms of recurrences for 4 x 0.45 ms of GEMM, the step
    # 18.2 -> 19.2 ms (profiles/r05_idle_split_proxy.txt) — so only where the recurrence is the clear majority tenant (idle < half the chip).
    n_idle = _idle_cus_beside_bwd_recurrence(dev, B, H)
    idle_ok = (group_ok and WGRAD_SIDE == "sk" and WGRAD_IDLE and T > 1
               and WGRAD_IDLE_MIN_CUS <= n_idle < torch.cuda.get_device_properties(dev).multi_processor_count // 2)
    fold_of = {l: ctx.layers[l].fold for l in range(L)}          # (kept apart: the layer contexts are cleared as backward moves down)
    ws_b = ops.rnn_ws("bwd", G, B, H, 1, dev) if WS_PREARM else None      # (the top layer's: behind the fc block's backward, which is short)
    for l in range(L - 1, -1, -1):
        lc = ctx.layers[l]
        if queued_side is not None:
            start = torch.cuda.Event()                           # everything of the layer above that is on the compute stream is behind this
            start.record(main)
            weight_gradients_group(*queued_side, True, start)
            queued_side = None
        if queued is not None:
            # the compute stream is about to start this layer's recurrence: the operand passes of the layer above start WITH it (started
            # earlier they would only take CUs from the GEMMs in between, which fill the register file and leave them no room)
            start = torch.cuda.Event()
            start.record(main)
            pending = operand_passes(*queued, start)
            queued = None
        start_idle = None
        if queued_idle is not None:
            start_idle = torch.cuda.Event()                      # the layer above's operands are final; the recurrence launch follows
            start_idle.record(main)
        dgx_bf = torch.empty(lc.gshape if lc.rec is not None else lc.gx.shape, dtype=torch.bfloat16, device=dev)
        want_tn = lc.h_bf is not None
        dhn_bf = torch.empty(M, 2 * H, dtype=torch.bfloat16, device=dev) if (want_tn and G == 3) else None
        bias_part = torch.empty(B, 2, 4, H, dtype=torch.float32, device=dev) if want_tn else None
        if isinstance(dy, BnGrad):
            ops.rnn_bwd_bn(G, dy.dyn, dy.x, dy.mean, dy.var, dy.gamma, dy.sums, lc.gx, lc.aux, lc.hbuf, lc.wpb, lens_dev, T, B, H, bf16=True,
                           dgx_bf16=dgx_bf, gates_bf16=lc.rec, dhn_bf16=dhn_bf, bias_part=bias_part, ws=ws_b)
        else:
            ops.rnn_bwd(G, dy, lc.gx, lc.aux, lc.hbuf, lc.wpb, lens_dev, T, B, H, bf16=True, dgx_bf16=dgx_bf, gates_bf16=lc.rec, dhn_bf16=dhn_bf,
                        bias_part=bias_part, ws=ws_b)
        _BWD_PERSISTENT[shape_key] = bool(ops.rnn_last_path(dev) & 2)
        if queued_idle is not None:
            # enqueued BEHIND the recurrence launch: the recurrence's workgroups are dispatched first, the GEMM's fill what is left
            weight_gradients_group(*queued_idle, _BWD_PERSISTENT[shape_key], start_idle)
            queued_idle = None
        tn = want_tn and _BWD_PERSISTENT[shape_key]              # (only a persistent launch writes d(hn) in bf16 and the bias sums)
        lc.rec = None
        if pending is not None:
            weight_gradients(pending)                            # heavy work of the layer above: on the compute stream, behind this launch
            pending = None
        if tn:
            queued_tn = (l, dgx_bf, dhn_bf, lc.h_bf, lc.xn, bias_part)
        else:
            queued = (l, (lc.aux, lc.hbuf, lc.xn), dgx_bf)
        # ---- critical path: dXn = dGx W_ih -> BatchNorm1d backward -> the next layer's dy
        ws_b = ops.rnn_ws("bwd", G, B, H, 1, dev) if (WS_PREARM and l > 0) else None     # the layer below's recurrence workspace, armed ahead of this GEMM
        dxn = ops.gemm_bf16_nt(dgx_bf, (lc.wihT if lc.wihT is not None else ops.cast_transpose_bf16(W[f"rnns.{l}.wih_cat"])))
        if l > 0:
            bp = f"rnns.{l}.batch_norm.module."
            dy = _bn_backward(dxn, lc.xin, lc.mean, lc.var, W[bp + "weight"], Gr[bp + "weight"], Gr[bp + "bias"], FUSE_BN_BWD, private)
        else:
            dy = dxn
        del dxn
        if tn and group_ok:
            ops.rnn_bias_grads(G, bias_part, Gr[f"rnns.{l}.bih_cat"], Gr[f"rnns.{l}.bhh_cat"])
            if side_ok:
                queued_side = queued_tn[:5]                      # released with the next layer's recurrence launch (layer 0: below)
            elif idle_ok and l > 0:
                queued_idle = queued_tn[:5]                      # launched behind the next layer's recurrence launch, on the CUs it leaves idle
            else:
                weight_gradients_group(*queued_tn[:5], False, None)
            queued_tn = None
        elif tn:
            weight_gradients_tn(*queued_tn)                      # nothing to prepare: straight behind the critical-path work of the layer
            queued_tn = None
        lc.gx = lc.aux = lc.hbuf = lc.xn = lc.wpb = lc.h_bf = lc.wihT = None
    if queued_side is not None:
        start = torch.cuda.Event()                               # layer 0: beside the conv-stack backward
        start.record(main)
        weight_gradients_group(*queued_side, True, start)
    if queued is not None:
        start = torch.cuda.Event()                               # layer 0: nothing latency-bound follows; run its passes now
        start.record(main)
        pending = operand_passes(*queued, start)
        weight_gradients(pending)
    return dy


def backward(W: Dict[str, Tensor], Gr: Dict[str, Tensor], cfg: ModelCfg, ctx: Ctx, dlogits: Tensor, on_bucket=None):
    """dlogits (T,B,C) contiguous.  Writes every parameter gradient into Gr[name] (same keys as W for
    parameters, plus the *_cat views).  Consumes ctx (gate buffers are overwritten in place).
    `on_bucket(name)` is called as soon as the gradients of 'fc', 'rnns.<l>', 'conv' are final (all
    kernels enqueued) — the data-parallel reducer launches that bucket's RCCL all-reduce there, so
    communication overlaps the rest of the backward pass."""
    done = on_bucket if on_bucket is not None else (lambda name: None)
    G, H, L, Cn = cfg.gates, cfg.hidden, cfg.layers, cfg.classes
    # Weight gradients (dW_ih, dW_hh and their casts) are not on the critical path of backward: they run on a side
    # stream and fill the CUs the latency-bound recurrent step kernels of the NEXT layer leave idle.
    main = torch.cuda.current_stream()
    # fp32 mode (this function's own loop): the same side stream when the backward recurrences of this shape run as persistent launches that
    # leave >= DS2_WGRAD_IDLE_MIN_CUS compute units without a workgroup (c2: 96 of 256 used) — the split-bf16 weight-gradient launch of a layer
    # then runs beside the next layer's recurrence on those CUs: c2 f32 33.9 -> 32.6 ms (the recurrence pays 3.40 -> 3.75 us per time step,
    # profiles/r04_wgrad_idle_ab.txt).  What the library did for the shape is known from the previous step (first step: one stream).
    f32_key = (ops.rnn_ctx_key(dlogits.device), G, H, ctx.B, "f32")
    idle_f32 = (cfg.precision != "bf16" and WGRAD_IDLE and _BWD_PERSISTENT.get(f32_key, False)
                and _idle_cus_beside_bwd_recurrence(dlogits.device, ctx.B, H) >= WGRAD_IDLE_MIN_CUS)
    side = _side_stream(dlogits.device) if (OVERLAP_WGRAD or idle_f32) else main
    keep = []      # tensors used on the side stream must outlive it (caching-allocator reuse is per stream)
    B, T, D1, D2 = ctx.B, ctx.T, ctx.D1, ctx.D2
    M = T * B
    lens_dev = ctx.lens_dev
    dl = dlogits.reshape(M, Cn)
    fp = "fc.0.module."
    # ---- fc -------------------------------------------------------------------------------------
    ops.gemm(dl, ctx.fc_xn, transA=True, out=Gr[fp + "1.weight"])                                 # (C, H)
    dxn = ops.gemm(dl, W[fp + "1.weight"])                                                        # (M, H)
    mean, var = ctx.fc_stats
    deferred = OVERLAP_MODE == "2" and cfg.precision == "bf16" and B % 8 == 0 and T > 1
    dy = _bn_backward(dxn, ctx.y_last, mean, var, W[fp + "0.weight"], Gr[fp + "0.weight"], Gr[fp + "0.bias"], FUSE_BN_BWD and deferred,
                      on_bucket is not None)
    del dxn
    done("fc")
    # ---- recurrent stack ------------------------------------------------------------------------
    if deferred:
        # a reducer that orders its collectives into the compute stream ("serial") must be called from that stream
        serial = getattr(getattr(on_bucket, "__self__", None), "mode", None) == "serial"
        dy = _backward_rnn_deferred(W, Gr, cfg, ctx, dy, done, on_bucket is not None, serial)
        first_layer = -1          # the loop below has nothing left to do
    else:
        first_layer = L - 1
    defer = idle_f32 and not OVERLAP_WGRAD
    pending_off = None
    for l in range(first_layer, -1, -1):
        lc = ctx.layers[l]
        bf = cfg.precision == "bf16"
        split = (not bf) and lc.xs is not None                   # fp32 mode, three-term split-bf16 products (forward decided: _f32_split_ok)
        # bf16 mode: the step kernels write dGx in bf16 (row-major) into a side buffer that the GEMMs consume directly (needs
        # B % 8 == 0 for the 16-byte aligned time-shifted operand views below; other batch sizes keep fp32 dGx + a cast pass)
        bfd = bf and B % 8 == 0
        gshape = lc.gshape if lc.rec is not None else lc.gx.shape
        dgx_bf = torch.empty(gshape, dtype=torch.bfloat16, device=dy.device) if bfd else None
        rmode = 1 if bf else (2 if (F32_RNN == "split" and H % 32 == 0) else 0)          # (the mode lc.wpb was packed for in forward)
        ops.rnn_bwd(G, dy, lc.gx, lc.aux, lc.hbuf, lc.wpb, lens_dev, T, B, H, bf16=rmode, dgx_bf16=dgx_bf, gates_bf16=lc.rec)
        if not bf:
            _BWD_PERSISTENT[f32_key] = bool(ops.rnn_last_path(dy.device) & 2)
        if pending_off is not None:
            fn, ev, pl = pending_off
            pending_off = None
            fn(after=ev)
            with torch.cuda.stream(side):
                done(f"rnns.{pl}")
        lc.rec = None
        dgx = dgx_bf if bfd else lc.gx                                                            # dGx (M, 2GH)
        # ---- critical path: dXn = dGx W_ih (feeds the next layer's backward) ---------------------------------------
        dgxT = dgs = dbih_sum = None
        if bfd:
            dxn = ops.gemm_bf16_nt(dgx_bf, (lc.wihT if lc.wihT is not None else ops.cast_transpose_bf16(W[f"rnns.{l}.wih_cat"])))
            # dW = dGx^T [Xn | h] needs the transposed copy; the same read gives db_ih = column sums of dGx
            dgxT, dbih_sum = ops.transpose_bf16(dgx_bf, colsum=Gr[f"rnns.{l}.bih_cat"].view(-1))       # sums land in the gradient buffer
        elif bf:
            dgx_r, dgxT, dbih_sum = ops.cast_bf16_both(dgx, colsum=True)
            dxn = ops.gemm_bf16_nt(dgx_r, (lc.wihT if lc.wihT is not None else ops.cast_transpose_bf16(W[f"rnns.{l}.wih_cat"])))
            del dgx_r
        elif split:
            # dXn = dGx W_ih as [hi | hi | lo](dGx) x [hi | lo | hi](W_ih^T)^T; the split copy of dGx also feeds both weight-gradient products
            dgs = ops.split_bf16(dgx, 0, pad_rows=64)                                             # (M rounded up to 64 like lc.xs, 3 * 2GH)
            wT = ops.transpose_batched(W[f"rnns.{l}.wih_cat"].unsqueeze(0))[0]                    # (I, 2GH)
            dxn = ops.gemm_bf16_nt(dgs[:M], ops.split_bf16(wT, 1))
            del wT
        else:
            dxn = ops.gemm(dgx, W[f"rnns.{l}.wih_cat"])                                           # (M, I)
        # ---- off the critical path: bias and weight gradients ------------------------------------------------------
        def off_path(l=l, lc=lc, dgx=dgx, dgs=dgs, dgxT=dgxT, dbih_sum=dbih_sum, bf=bf, bfd=bfd, split=split, after=None):
            if after is not None:
                side.wait_event(after)
            else:
                side.wait_stream(main)
            keep.append((dgx, lc.aux, lc.hbuf, lc.xn))
            with torch.cuda.stream(side):
                dbih = Gr[f"rnns.{l}.bih_cat"]
                if not bfd:
                    dbih.copy_(dbih_sum if bf else ops.colsum(dgx))
                dbhh = Gr[f"rnns.{l}.bhh_cat"]                                                        # (2, GH)
                dbhh.copy_(dbih.view(2, G * H))
                auxT = None
                if G == 3 and T > 1 and bfd:
                    # d(b_hn) = column sums of d(hn): taken from the read that produces the transposed bf16 copy for dW_hh below
                    dbn = torch.empty(2 * H, dtype=torch.float32, device=dgx.device)
                    auxT = ops.cast_transpose_bf16(lc.aux, colsum=dbn)
                    dbhh[:, 2 * H:] = dbn.view(2, H)
                elif G == 3:
                    dbhh[:, 2 * H:] = ops.colsum(lc.aux).view(2, H)
                # dW_hh[dir] = sum_t dGh[t]^T h_prev[t]  (h_prev = h[t-1] fwd / h[t+1] reverse)
                dwhh = Gr[f"rnns.{l}.whh_cat"]                                                        # (2, GH, H)
                if T > 1 and bfd:
                    # bf16 MFMA path: transposed bf16 copies (dgxT: (2GH, M)), the time shift is a column offset of B elements
                    hT = ops.cast_transpose_bf16(lc.hbuf)                                             # (2H, M)
                    rows = 2 * H if G == 3 else 4 * H
                    # both directions per launch: direction 0 pairs rows t of dGh with h[t-1], direction 1 rows t with h[t+1]
                    ka = (slice(B, M), slice(0, M - B))
                    kb = (slice(0, M - B), slice(B, M))
                    ops.gemm_bf16_nt_pair(dgxT[0:rows, ka[0]], dgxT[G * H:G * H + rows, ka[1]], hT[0:H, kb[0]], hT[H:2 * H, kb[1]], dwhh[:, :rows])
                    if G == 3:
                        ops.gemm_bf16_nt_pair(auxT[0:H, ka[0]], auxT[H:2 * H, ka[1]], hT[0:H, kb[0]], hT[H:2 * H, kb[1]], dwhh[:, 2 * H:])
                    keep.append((dgxT, hT, auxT))
                elif T > 1 and split:
                    # every product = three terms (hi.hi, hi.lo, lo.hi) on row-pitched views of the split copies; ALL terms of ALL products of the
                    # layer (dW_hh of both directions, a GRU's n rows, dW_ih) in ONE grouped split-K launch + one reduce (terms of a product
                    # name the same output and are summed: ds2_gemm_bf16_tn_splitk_group)
                    C2, Iw = 2 * G * H, W[f"rnns.{l}.wih_cat"].shape[1]
                    Ip = lc.xs.shape[1] // 3
                    hs = ops.split_bf16(lc.hbuf, 2)                                                   # (M, 4H) = [hi | lo]
                    d_hi, d_lo, h_hi, h_lo = dgs[:, :C2], dgs[:, 2 * C2:], hs[:, :2 * H], hs[:, 2 * H:]
                    rows = 2 * H if G == 3 else 4 * H
                    ra, rb = (slice(B, M), slice(0, M - B)), (slice(0, M - B), slice(B, M))
                    probs = []
                    for d in (0, 1):
                        for a, b in ((d_hi, h_hi), (d_hi, h_lo), (d_lo, h_hi)):
                            probs.append((a[ra[d], d * G * H:d * G * H + rows], b[rb[d], d * H:(d + 1) * H], dwhh[d, :rows]))
                    if G == 3:                                                                        # n-gate rows use d(hn) (aux) instead of dGx_n
                        axs = ops.split_bf16(lc.aux, 2)                                               # (M, 4H)
                        a_hi, a_lo = axs[:, :2 * H], axs[:, 2 * H:]
                        for d in (0, 1):
                            for a, b in ((a_hi, h_hi), (a_hi, h_lo), (a_lo, h_hi)):
                                probs.append((a[ra[d], d * H:(d + 1) * H], b[rb[d], d * H:(d + 1) * H], dwhh[d, 2 * H:]))
                        keep.append((axs,))
                    x_hi, x_lo = lc.xs[:, :Iw], lc.xs[:, 2 * Ip:2 * Ip + Iw]
                    for a, b in ((d_hi, x_hi), (d_hi, x_lo), (d_lo, x_hi)):
                        probs.append((a, b, Gr[f"rnns.{l}.wih_cat"]))
                    if len(probs) <= 16 and Iw % 8 == 0:
                        ops.gemm_bf16_tn_splitk_group(probs)
                    else:                                                                             # (never at the reference's shapes) term by term
                        seen = set()
                        for a, b, o in probs:
                            ops.gemm_bf16_tn(a, b, out=o, accumulate=o.data_ptr() in seen)
                            seen.add(o.data_ptr())
                    keep.append((hs, dgs, lc.xs))
                elif T > 1:
                    K = (T - 1) * B
                    ldg, ldh = 2 * G * H, 2 * H
                    a0 = dgx.data_ptr() + 4 * (B * ldg)                 # dir 0: rows t >= 1
                    b0 = lc.hbuf.data_ptr()                             #        h[t-1]
                    a1 = dgx.data_ptr() + 4 * (G * H)                   # dir 1: rows t <= T-2, column block of dir 1
                    b1 = lc.hbuf.data_ptr() + 4 * (H + B * ldh)         #        h[t+1]
                    sA, sB = (a1 - a0) // 4, (b1 - b0) // 4
                    rows = 2 * H if G == 3 else 4 * H
                    ops.gemm_raw(True, False, rows, H, K, a0, ldg, sA, b0, ldh, sB, dwhh.data_ptr(), H, G * H * H, dgx.device, batch=2)
                    if G == 3:  # n-gate rows use d(hn) (aux) instead of dGx_n
                        x0 = lc.aux.data_ptr() + 4 * (B * ldh)
                        x1 = lc.aux.data_ptr() + 4 * H
                        ops.gemm_raw(True, False, H, H, K, x0, ldh, (x1 - x0) // 4, b0, ldh, sB, dwhh.data_ptr() + 4 * (2 * H * H), H,
                                     G * H * H, dgx.device, batch=2)
                else:
                    dwhh.zero_()
                # dW_ih (2GH, I) = dGx^T Xn
                if bf:
                    xnT = ops.transpose_bf16(lc.xn[:, :W[f"rnns.{l}.wih_cat"].shape[1]])              # lc.xn is bf16 (M, pad8(I)) in this mode
                    ops.gemm_bf16_nt(dgxT, xnT, out=Gr[f"rnns.{l}.wih_cat"])
                    keep.append((dgxT, xnT))
                elif split and T > 1:
                    pass                                                                              # (dW_ih went out with the grouped launch above)
                elif split:
                    C2, I = 2 * G * H, W[f"rnns.{l}.wih_cat"].shape[1]
                    Ip = lc.xs.shape[1] // 3
                    for k, (a, b) in enumerate(((dgs[:, :C2], lc.xs[:, :I]), (dgs[:, :C2], lc.xs[:, 2 * Ip:2 * Ip + I]), (dgs[:, 2 * C2:], lc.xs[:, :I]))):
                        ops.gemm_bf16_tn(a, b, out=Gr[f"rnns.{l}.wih_cat"], accumulate=k > 0)
                    keep.append((dgs, lc.xs))
                else:
                    ops.gemm(dgx, lc.xn, transA=True, out=Gr[f"rnns.{l}.wih_cat"])
            lc.gx = lc.aux = lc.hbuf = lc.xn = lc.wpb = lc.xs = lc.wihT = None
        if not defer:
            off_path()
        if l > 0:
            bp = f"rnns.{l}.batch_norm.module."
            dy = ops.bn1d_bwd(dxn, lc.xin, lc.mean, lc.var, W[bp + "weight"], Gr[bp + "weight"], Gr[bp + "bias"])
        else:
            dy = dxn
        del dxn
        if defer and l > 0:
            # launched behind the NEXT layer's recurrence launch (the loop's top): dispatched first, the GEMM's 256 x 256 workgroups would hold
            # every CU for their first round and the recurrence would wait for them
            ev = torch.cuda.Event()
            ev.record(main)                 # the layer's operands and its BatchNorm gradients are final
            pending_off = (off_path, ev, l)
            continue
        if defer:
            off_path()
        if side is not main:
            side.wait_stream(main)          # the bucket also holds this layer's BN grads (main stream)
        with torch.cuda.stream(side):
            done(f"rnns.{l}")
    # ---- conv stack -----------------------------------------------------------------------------
    cp = "conv.seq_module."
    da2 = ops.transpose_bft(dy, B, 32 * D2, T, to_tbf=False).view(B, 32, D2, T)
    m2, v2 = ctx.st2
    bf = cfg.precision == "bf16"
    if bf:
        # dY2 leaves the BatchNorm backward directly as the two bf16 operands conv2's weight gradient / input gradient take, together
        # with conv2's bias gradient (its per-channel sums); conv1's stage keeps fp32 dY1 (cast on the fly by conv1's weight gradient)
        _, dy2p, dy2n = ops.bn2d_act_bwd_fused(ctx.y2, da2, lens_dev, m2, v2, W[cp + "4.weight"], W[cp + "4.bias"], Gr[cp + "4.weight"],
                                               Gr[cp + "4.bias"], Gr[cp + "3.bias"], want_pad=CONV2_WGRAD != "nhwc", want_nhwc=True)
        del da2
        if CONV2_WGRAD == "nhwc":
            ops.conv2_wgrad_nhwc_bf16(ctx.a1p, dy2n, lens_dev, Gr[cp + "3.weight"])
        else:
            ops.conv2_wgrad_bf16(ctx.a1p, dy2p, lens_dev, Gr[cp + "3.weight"], T)
        da1 = ops.conv2_dgrad_bf16(dy2n, ctx.packs[1], ctx.packs[2], D1)
        del dy2p, dy2n
        m1, v1 = ctx.st1
        dy1, _, _ = ops.bn2d_act_bwd_fused(ctx.y1, da1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"], Gr[cp + "1.weight"],
                                           Gr[cp + "1.bias"], Gr[cp + "0.bias"], want_f32=True)
        del da1
        ops.conv1_wgrad_bf16(ctx.x16t, dy1, lens_dev, Gr[cp + "0.weight"], ctx.x.shape[3])
        ctx.x16t = None
    elif F32_CONV == "split":
        dy2, dy2p, dy2n = ops.bn2d_act_bwd_fused(ctx.y2, da2, lens_dev, m2, v2, W[cp + "4.weight"], W[cp + "4.bias"], Gr[cp + "4.weight"],
                                                 Gr[cp + "4.bias"], Gr[cp + "3.bias"], want_f32=True, want_pad=True, want_nhwc=True)
        del da2
        r2 = ops.bf16_residual(dy2)
        del dy2
        dy2p_lo, dy2n_lo = ops.padcast_bf16(r2), ops.nhwc_bf16(r2)
        r2 = ops.bf16_residual(ctx.a1)
        a1p = (ops.padcast_bf16(ctx.a1), ops.padcast_bf16(r2))
        del r2
        ph = ops.conv2_pack_bf16(W[cp + "3.weight"])
        pl = ops.conv2_pack_bf16(ops.bf16_residual(W[cp + "3.weight"]))
        # dW2 = a_hi (x) dy_hi + a_lo (x) dy_hi + a_hi (x) dy_lo;  da1 = dy_hi * w_hi + dy_lo * w_hi + dy_hi * w_lo
        g2 = Gr[cp + "3.weight"]
        gb, gc = torch.empty_like(g2), torch.empty_like(g2)
        ops.conv2_wgrad_bf16(a1p[0], dy2p, lens_dev, g2, T)
        ops.conv2_wgrad_bf16(a1p[1], dy2p, lens_dev, gb, T)
        ops.conv2_wgrad_bf16(a1p[0], dy2p_lo, lens_dev, gc, T)
        ops.sum3_(g2, gb, gc)
        del a1p
        wd0, wd1, wl0, wl1 = ph[1], ph[2], pl[1], pl[2]
        da1 = ops.conv2_dgrad_bf16(dy2n, wd0, wd1, D1)
        ops.sum3_(da1, ops.conv2_dgrad_bf16(dy2n_lo, wd0, wd1, D1), ops.conv2_dgrad_bf16(dy2n, wl0, wl1, D1))
        del dy2p, dy2n, dy2p_lo, dy2n_lo
        m1, v1 = ctx.st1
        dy1 = ops.bn2d_act_bwd(ctx.y1, da1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"], Gr[cp + "1.weight"], Gr[cp + "1.bias"])
        del da1
        Gr[cp + "0.bias"].copy_(ops.chan_sum(dy1))
        ops.conv1_wgrad(ctx.x, dy1, lens_dev, Gr[cp + "0.weight"])
    else:
        dy2 = ops.bn2d_act_bwd(ctx.y2, da2, lens_dev, m2, v2, W[cp + "4.weight"], W[cp + "4.bias"], Gr[cp + "4.weight"], Gr[cp + "4.bias"])
        del da2
        Gr[cp + "3.bias"].copy_(ops.chan_sum(dy2))
        ops.conv2_wgrad(ctx.a1, dy2, lens_dev, Gr[cp + "3.weight"])
        da1 = ops.conv2_dgrad(dy2, ctx.packs[0], D1)
        del dy2
        m1, v1 = ctx.st1
        dy1 = ops.bn2d_act_bwd(ctx.y1, da1, lens_dev, m1, v1, W[cp + "1.weight"], W[cp + "1.bias"], Gr[cp + "1.weight"], Gr[cp + "1.bias"])
        del da1
        Gr[cp + "0.bias"].copy_(ops.chan_sum(dy1))
        ops.conv1_wgrad(ctx.x, dy1, lens_dev, Gr[cp + "0.weight"])
    done("conv")
    if side is not main:
        main.wait_stream(side)              # all weight gradients are final for whoever runs next on the main stream
    if getattr(ctx, "side_used", False):
        main.wait_stream(_side_stream(dlogits.device))
    keep.clear()
