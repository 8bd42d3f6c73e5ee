"""CPU oracle for the DeepSpeech2 train-step hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product (`asr_amd`) never imports it and has no CPU fallback.

This is an independent *restatement* (padded + masked formulation, explicit loops over time, no
`pack_padded_sequence`, no `nn.GRU`) of what the reference computes on the path

    DeepSpeechTrainer.fit      asr_deepspeech/trainers/deepspeech_trainer.py:102-117
    DeepSpeech.forward         asr_deepspeech/modules/deepspeech.py:130-149
    MaskConv.forward           asr_deepspeech/modules/blocks.py:42-56
    BatchRNN.forward           asr_deepspeech/modules/blocks.py:84-93
    torch.nn.CTCLoss(sum)      asr_deepspeech/trainers/__main__.py:53 (third-party: torch,
                               pinned torch 2.12.1 / 2.8.0 in uv.lock; algorithm = Graves et al.
                               2006 alpha/beta recursion in log space, blank = 0)

Parity pinning: the reference's own tests hold no numeric known answers for this path
(SURVEY.md §4), so this oracle is pinned against golden vectors produced by importing the
unmodified reference in the build container (`tests/golden/make_golden.py`), see
`tests/test_oracle_golden.py`.

Everything is a pure function of a `state_dict`-keyed dict of tensors (reference key names,
SURVEY.md Appendix A.1), so the same dict drives the reference, the oracle and the HIP path.
All functions are differentiable through torch autograd except `ctc_*_np` (numpy, explicit grad).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# lengths  (deepspeech.py:275-288, deepspeech_trainer.py:104, functional.py:19,28)
# --------------------------------------------------------------------------------------------
def seq_lens_after_conv(input_lengths: Tensor) -> Tensor:
    """deepspeech.py:275-288: per Conv2d (L + 2p - d(k-1) - 1)/s + 1 in TRUE division on the time
    axis, a single truncation at the end.  conv1: k=11,s=2,p=5; conv2: k=11,s=1,p=5."""
    L = input_lengths.to(torch.int32)
    L = (L + 2 * 5 - 1 * (11 - 1) - 1) / 2 + 1  # float tensor after '/'
    L = (L + 2 * 5 - 1 * (11 - 1) - 1) / 1 + 1
    return L.int()


def lengths_from_percentages(input_percentages: Tensor, t_max: int) -> Tensor:
    """deepspeech_trainer.py:104 — float32 multiply then truncation; pure here
    (the reference mutates the percentages in place)."""
    return input_percentages.to(torch.float32).mul(int(t_max)).int()


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _time_mask(lengths: Tensor, t: int, dtype, device) -> Tensor:
    """(B, T) 1/0 mask, 1 where t < length[b]."""
    ar = torch.arange(t, device=device).unsqueeze(0)
    return (ar < lengths.to(device).unsqueeze(1)).to(dtype)


def batch_norm_train(x: Tensor, gamma: Tensor, beta: Tensor, reduce_dims, shape) -> Tuple[Tensor, Tensor, Tensor]:
    """Training-mode batch norm: biased variance for normalisation.
    Returns (y, mean, biased_var)."""
    mean = x.mean(dim=reduce_dims)
    var = ((x - mean.view(shape)) ** 2).mean(dim=reduce_dims)
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)
    return y, mean, var


def batch_norm_eval(x, gamma, beta, rmean, rvar, shape):
    return (x - rmean.view(shape)) * torch.rsqrt(rvar.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)


def _update_running(stats: Optional[dict], key: str, mean: Tensor, var_b: Tensor, n: int):
    """running = (1-m)*running + m*batch; running_var uses the UNBIASED batch variance."""
    if stats is None:
        return
    stats[key + ".batch_mean"] = mean.detach().clone()
    stats[key + ".batch_var"] = var_b.detach().clone()
    stats[key + ".count"] = n


def conv_stack(x: Tensor, out_lens: Tensor, sd: Dict[str, Tensor], training: bool = True,
               stats: Optional[dict] = None, taps: Optional[dict] = None) -> Tensor:
    """MaskConv over Conv-BN-Hardtanh-Conv-BN-Hardtanh (deepspeech.py:60-67, blocks.py:42-56).
    x: (B,1,161,T_in) -> (B,32,41,T).  The mask (t >= out_len[b] -> 0) is applied after EACH of the
    six sub-modules; BN statistics therefore include the zeros of masked positions (A.3/A.4)."""
    B = x.shape[0]

    def mask(v):
        m = _time_mask(out_lens, v.shape[3], v.dtype, v.device).view(B, 1, 1, -1)
        return v * m

    p = "conv.seq_module."
    y = torch.nn.functional.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=(2, 2), padding=(20, 5))
    y = mask(y)
    if taps is not None:
        taps["conv1"] = y
    if training:
        z, mu, var = batch_norm_train(y, sd[p + "1.weight"], sd[p + "1.bias"], (0, 2, 3), (1, -1, 1, 1))
        _update_running(stats, p + "1", mu, var, y.numel() // y.shape[1])
    else:
        z = batch_norm_eval(y, sd[p + "1.weight"], sd[p + "1.bias"], sd[p + "1.running_mean"],
                            sd[p + "1.running_var"], (1, -1, 1, 1))
    if taps is not None:
        taps["bn1"] = z
    z = mask(z)
    a = mask(torch.clamp(z, 0.0, 20.0))
    if taps is not None:
        taps["act1"] = a
    y = torch.nn.functional.conv2d(a, sd[p + "3.weight"], sd[p + "3.bias"], stride=(2, 1), padding=(10, 5))
    y = mask(y)
    if taps is not None:
        taps["conv2"] = y
    if training:
        z, mu, var = batch_norm_train(y, sd[p + "4.weight"], sd[p + "4.bias"], (0, 2, 3), (1, -1, 1, 1))
        _update_running(stats, p + "4", mu, var, y.numel() // y.shape[1])
    else:
        z = batch_norm_eval(y, sd[p + "4.weight"], sd[p + "4.bias"], sd[p + "4.running_mean"],
                            sd[p + "4.running_var"], (1, -1, 1, 1))
    if taps is not None:
        taps["bn2"] = z
    z = mask(z)
    a = mask(torch.clamp(z, 0.0, 20.0))
    if taps is not None:
        taps["act2"] = a
    return a


def hardtanh_kink_margin(sd: Dict[str, Tensor], x: Tensor, lengths: Tensor) -> float:
    """Smallest distance of any un-masked BatchNorm2d output to a Hardtanh kink (0 or 20).
    d/dz Hardtanh is discontinuous there: an element within fp32 round-off (~1e-6) of a kink takes
    either branch depending on summation order — in the reference's own CPU kernels too — and moves
    the gradients by O(1/sqrt(#elements)).  Parity fixtures are chosen with margin >= 4e-6 (10x round-off) so that
    they test the kernels, not this ill-conditioning ('Numerics')."""
    out_lens = seq_lens_after_conv(lengths.cpu().int())
    taps: dict = {}
    with torch.no_grad():
        conv_stack(x.double(), out_lens, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()
                                           if k.startswith("conv.")}, True, None, taps)
    B = x.shape[0]
    m = 1e30
    for key in ("bn1", "bn2"):
        z = taps[key]
        msk = _time_mask(out_lens, z.shape[3], torch.bool, z.device).view(B, 1, 1, -1).expand_as(z)
        zz = z[msk]
        m = min(m, float(torch.minimum(zz.abs(), (zz - 20.0).abs()).min()))
    return m


def collapse_to_tbf(a: Tensor) -> Tensor:
    """deepspeech.py:135-137: (B,C,D,T) -> (T,B,C*D), feature index = c*D + d."""
    B, C, D, T = a.shape
    return a.reshape(B, C * D, T).permute(2, 0, 1).contiguous()


def gru_direction(gx: Tensor, w_hh: Tensor, b_hh: Tensor, lens: Tensor, reverse: bool) -> Tensor:
    """One direction of a 1-layer GRU in padded+masked form.
    gx: (T,B,3H) = x W_ih^T + b_ih, gate order r,z,n.  h0 = 0.  Rows t >= len[b] output 0 and do
    not advance the state; the reverse direction therefore starts at each sample's own last frame."""
    T, B, H3 = gx.shape
    H = H3 // 3
    h = gx.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    lens_d = lens.to(gx.device)
    for t in order:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gx[t, :, :H] + gh[:, :H])
        z = torch.sigmoid(gx[t, :, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[t, :, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1.0 - z) * n + z * h
        m = (t < lens_d).to(gx.dtype).unsqueeze(1)
        h = m * h_new + (1.0 - m) * h
        outs[t] = m * h_new
    return torch.stack(outs, 0)


def lstm_direction(gx: Tensor, w_hh: Tensor, b_hh: Tensor, lens: Tensor, reverse: bool) -> Tensor:
    """One direction of a 1-layer LSTM, gate order i,f,g,o, h0 = c0 = 0."""
    T, B, H4 = gx.shape
    H = H4 // 4
    h = gx.new_zeros(B, H)
    c = gx.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    lens_d = lens.to(gx.device)
    for t in order:
        g = gx[t] + h @ w_hh.t() + b_hh
        i = torch.sigmoid(g[:, :H])
        f = torch.sigmoid(g[:, H:2 * H])
        gg = torch.tanh(g[:, 2 * H:3 * H])
        o = torch.sigmoid(g[:, 3 * H:])
        c_new = f * c + i * gg
        h_new = o * torch.tanh(c_new)
        m = (t < lens_d).to(gx.dtype).unsqueeze(1)
        h = m * h_new + (1.0 - m) * h
        c = m * c_new + (1.0 - m) * c
        outs[t] = m * h_new
    return torch.stack(outs, 0)


def batch_rnn(x: Tensor, out_lens: Tensor, sd: Dict[str, Tensor], prefix: str, rnn_type: str,
              batch_norm: bool, training: bool = True, stats: Optional[dict] = None) -> Tensor:
    """BatchRNN.forward (blocks.py:84-93): [BN1d over all T*B rows incl. padding] -> bi-RNN ->
    sum of the two directions (blocks.py:92).  x: (T,B,I) -> (T,B,H)."""
    T, B, I = x.shape
    if batch_norm:
        bp = prefix + "batch_norm.module."
        flat = x.reshape(T * B, I)
        if training:
            flat, mu, var = batch_norm_train(flat, sd[bp + "weight"], sd[bp + "bias"], (0,), (1, -1))
            _update_running(stats, bp[:-1], mu, var, T * B)
        else:
            flat = batch_norm_eval(flat, sd[bp + "weight"], sd[bp + "bias"], sd[bp + "running_mean"],
                                   sd[bp + "running_var"], (1, -1))
        x = flat.reshape(T, B, I)
    rp = prefix + "rnn."
    step = gru_direction if rnn_type == "gru" else lstm_direction
    y = None
    for sfx, rev in (("", False), ("_reverse", True)):
        gx = x @ sd[rp + "weight_ih_l0" + sfx].t() + sd[rp + "bias_ih_l0" + sfx]
        yd = step(gx, sd[rp + "weight_hh_l0" + sfx], sd[rp + "bias_hh_l0" + sfx], out_lens, rev)
        y = yd if y is None else y + yd
    return y


def fc_block(x: Tensor, sd: Dict[str, Tensor], training: bool = True, stats: Optional[dict] = None) -> Tensor:
    """deepspeech.py:103-109: SequenceWise(BatchNorm1d(H) -> Linear(H, C, bias=False)). (T,B,H)->(T,B,C)."""
    T, B, H = x.shape
    flat = x.reshape(T * B, H)
    p = "fc.0.module."
    if training:
        flat, mu, var = batch_norm_train(flat, sd[p + "0.weight"], sd[p + "0.bias"], (0,), (1, -1))
        _update_running(stats, p + "0", mu, var, T * B)
    else:
        flat = batch_norm_eval(flat, sd[p + "0.weight"], sd[p + "0.bias"], sd[p + "0.running_mean"],
                               sd[p + "0.running_var"], (1, -1))
    return (flat @ sd[p + "1.weight"].t()).reshape(T, B, -1)


def num_layers(sd: Dict[str, Tensor]) -> int:
    n = 0
    while f"rnns.{n}.rnn.weight_ih_l0" in sd:
        n += 1
    return n


def rnn_kind(sd: Dict[str, Tensor]) -> str:
    w = sd["rnns.0.rnn.weight_hh_l0"]
    g = w.shape[0] // w.shape[1]
    return {3: "gru", 4: "lstm"}[g]


def forward(sd: Dict[str, Tensor], x: Tensor, lengths: Tensor, training: bool = True,
            stats: Optional[dict] = None, taps: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """DeepSpeech.forward (deepspeech.py:130-149) -> (logits (B,T,C) [softmax(-1) in eval], out_lens int32 CPU)."""
    out_lens = seq_lens_after_conv(lengths.cpu().int())
    a = conv_stack(x, out_lens, sd, training, stats, taps)
    h = collapse_to_tbf(a)
    kind = rnn_kind(sd)
    for l in range(num_layers(sd)):
        h = batch_rnn(h, out_lens, sd, f"rnns.{l}.", kind, batch_norm=(l > 0), training=training, stats=stats)
        if taps is not None:
            taps[f"rnn{l}"] = h
    logits = fc_block(h, sd, training, stats).transpose(0, 1)
    if not training:
        logits = torch.softmax(logits, dim=-1)  # InferenceBatchSoftmax, blocks.py:59-64
    return logits, out_lens


# --------------------------------------------------------------------------------------------
# CTC (numpy, explicit)  — restates aten::_ctc_loss semantics used via nn.CTCLoss(reduction="sum")
# --------------------------------------------------------------------------------------------
def _lse(*vals):
    m = max(vals)
    if m == -math.inf:
        return -math.inf
    return m + math.log(sum(math.exp(v - m) for v in vals))


def ctc_nll_and_grad_np(log_probs: np.ndarray, targets: np.ndarray, in_lens, tgt_lens, blank: int = 0):
    """log_probs (T,B,C) float64/32 (already log-softmaxed), flat targets.  Returns
    (nll per utterance (B,), grad wrt *logits* (T,B,C)) where grad = softmax - posterior occupancy
    for t < in_len, 0 beyond; an infeasible alignment gives nll=+inf (zero_infinity=False) and an
    unspecified (here: zero) gradient row-block."""
    lp = np.asarray(log_probs, dtype=np.float64)
    T, B, C = lp.shape
    nll = np.zeros(B)
    grad = np.zeros_like(lp)
    off = 0
    for b in range(B):
        Tb, U = int(in_lens[b]), int(tgt_lens[b])
        lab = [int(v) for v in targets[off:off + U]]
        off += U
        ext = [blank]
        for v in lab:
            ext += [v, blank]
        S = len(ext)
        NEG = -math.inf
        alpha = np.full((Tb, S), NEG)
        beta = np.full((Tb, S), NEG)
        if Tb > 0:
            alpha[0, 0] = lp[0, b, blank]
            if S > 1:
                alpha[0, 1] = lp[0, b, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                a = [alpha[t - 1, s]]
                if s >= 1:
                    a.append(alpha[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    a.append(alpha[t - 1, s - 2])
                v = _lse(*a)
                alpha[t, s] = v + lp[t, b, ext[s]] if v != NEG else NEG
        if Tb > 0:
            ll = _lse(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2]) if S > 1 else alpha[Tb - 1, S - 1]
        else:
            ll = 0.0 if S == 1 else NEG
        nll[b] = -ll
        if ll == NEG:
            nll[b] = math.inf
            continue
        beta[Tb - 1, S - 1] = lp[Tb - 1, b, blank]
        if S > 1:
            beta[Tb - 1, S - 2] = lp[Tb - 1, b, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                a = [beta[t + 1, s]]
                if s + 1 < S:
                    a.append(beta[t + 1, s + 1])
                if s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]:
                    a.append(beta[t + 1, s + 2])
                v = _lse(*a)
                beta[t, s] = v + lp[t, b, ext[s]] if v != NEG else NEG
        for t in range(Tb):
            occ = np.zeros(C)
            for s in range(S):
                ab = alpha[t, s] + beta[t, s]
                if ab != NEG:
                    occ[ext[s]] += math.exp(ab - lp[t, b, ext[s]] - ll)
            grad[t, b, :] = np.exp(lp[t, b, :]) - occ
    return nll, grad


def ctc_loss_sum(log_probs: Tensor, targets: Tensor, in_lens: Tensor, tgt_lens: Tensor) -> Tensor:
    """Differentiable CTC 'sum' loss for the whole-model oracle (uses aten's CPU kernel — the very
    third-party function the reference calls, trainers/__main__.py:53 — so that whole-model oracle
    grads flow through autograd).  `ctc_nll_and_grad_np` is the independent restatement used to
    check the HIP CTC kernel and is itself pinned against this function in tests."""
    return torch.nn.functional.ctc_loss(log_probs, targets, in_lens, tgt_lens, blank=0,
                                        reduction="sum", zero_infinity=False)


# --------------------------------------------------------------------------------------------
# fit() + backward as one pure function  (deepspeech_trainer.py:102-117, :86-95)
# --------------------------------------------------------------------------------------------
def check_loss_value(loss_value: float) -> bool:
    """functional.py:45-61 on the scalar: invalid if +-inf, NaN, or negative."""
    return not (math.isinf(loss_value) or math.isnan(loss_value) or loss_value < 0)


def fit_and_grads(sd: Dict[str, Tensor], inputs: Tensor, targets: Tensor, input_percentages: Tensor,
                  target_sizes: Tensor, dtype=torch.float32):
    """Runs the reference's statement sequence on the oracle model.  Returns dict with logits,
    out_lens, loss (= CTC sum / B), grads {param key: grad}, stats (batch mean/var per BN)."""
    params = {}
    for k, v in sd.items():
        v = v.detach().to(dtype) if v.is_floating_point() else v.detach()
        if v.is_floating_point() and "running_" not in k:
            v = v.clone().requires_grad_(True)
        params[k] = v
    stats: dict = {}
    input_sizes = lengths_from_percentages(input_percentages, inputs.size(3))
    out, out_lens = forward(params, inputs.to(dtype), input_sizes, training=True, stats=stats)
    lp = out.transpose(0, 1).log_softmax(2)
    loss = ctc_loss_sum(lp, targets, out_lens, target_sizes) / inputs.size(0)
    grads = {}
    if check_loss_value(float(loss.detach())):
        keys = [k for k, v in params.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(keys, gs)}
    return {"logits": out.detach(), "out_lens": out_lens, "loss": float(loss.detach()), "grads": grads,
            "stats": stats, "input_sizes": input_sizes}


def adamw_step_np(p, g, m, v, step, lr=1.5e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """torch.optim.AdamW single-tensor update (trainers/__main__.py:41-47 hyper-parameters from
    asr_deepspeech/config.yml:41-47).  step is 1-based.  Returns (p, m, v)."""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
