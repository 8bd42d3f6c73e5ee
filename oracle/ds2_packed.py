"""CPU baseline in the reference's OWN formulation.  TEST / BENCH INFRASTRUCTURE — NOT PRODUCT CODE.

Only `tests/` and `bench.py`'s `cpu_baseline` leg import this module.

`oracle/ds2_oracle.py` restates the hot path in padded + masked form with explicit time loops — the right shape for a
parity oracle, but not what the reference's CPU trainer executes.  SURVEY.md §8(d) asks for the CPU baseline in the
form the reference itself runs, so that it pays the same costs:

    BatchRNN.forward        asr_deepspeech/modules/blocks.py:84-93
        [SequenceWise BN1d] -> pack_padded_sequence(x, lengths) -> one-layer bidirectional aten::gru / aten::lstm
        -> pad_packed_sequence -> sum of the two directions
    MaskConv.forward        asr_deepspeech/modules/blocks.py:42-56   (mask after EACH of Conv, BN, Hardtanh, twice)
    DeepSpeech.forward      asr_deepspeech/modules/deepspeech.py:130-149
    fit + train-step tail   asr_deepspeech/trainers/deepspeech_trainer.py:86-97, 102-117   (log_softmax, CTCLoss(sum) / B,
                            zero_grad -> backward -> AdamW.step)

written here functionally over the same `state_dict`-keyed tensors the other oracle uses (torch.nn.functional +
`torch.nn.utils.rnn` + the fused `torch._VF.gru/lstm` kernels on packed data — the aten ops `nn.GRU` / `nn.LSTM` dispatch to),
so one weight dict drives the reference, both CPU restatements and the HIP path.  Pinned by tests/test_oracle_golden.py:
logits / loss / gradients / 3-step AdamW loss curve against the golden vectors generated from the imported reference.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import ds2_oracle as O

Tensor = torch.Tensor


def _mask_time(x: Tensor, lengths: Tensor) -> Tensor:
    """blocks.py:50-55: zero [b, :, :, T_b:] (lengths are the FINAL conv-stack output lengths for every stage)."""
    t = x.size(3)
    keep = (torch.arange(t).view(1, 1, 1, t) < lengths.view(-1, 1, 1, 1).to(torch.long))
    return x * keep.to(x.dtype)


def _bn(x, sd, key, training):
    """native_batch_norm in training mode updates sd[key.running_*] in place, like nn.BatchNorm (momentum 0.1, eps 1e-5)."""
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=training, momentum=O.BN_MOMENTUM, eps=O.BN_EPS)


def forward(sd: Dict[str, Tensor], x: Tensor, lengths: Tensor, training: bool = True, packed: bool = True):
    """x (B,1,F,T_in), lengths (B,) input frames, sorted descending (pack_padded_sequence enforces it, blocks.py:87).
    Returns (logits (B,T,C), output_lengths int32).
    packed=False: the recurrent layers run the same fused aten gru / lstm on the PADDED (T,B,I) tensor, without pack / unpack — exactly
    equivalent when every utterance of the batch has the full length (nothing is padded), which is asserted; this is
    the 'fair CPU' form of the bench's baseline leg (no per-time-step packed-sequence slices in autograd)."""
    out_lens = O.seq_lens_after_conv(lengths.cpu().int())
    cp = "conv.seq_module."
    # ---- MaskConv over Conv, BN, Hardtanh, Conv, BN, Hardtanh (deepspeech.py:60-67)
    a = _mask_time(F.conv2d(x, sd[cp + "0.weight"], sd[cp + "0.bias"], stride=(2, 2), padding=(20, 5)), out_lens)
    a = _mask_time(_bn(a, sd, cp + "1", training), out_lens)
    a = _mask_time(F.hardtanh(a, 0.0, 20.0), out_lens)
    a = _mask_time(F.conv2d(a, sd[cp + "3.weight"], sd[cp + "3.bias"], stride=(2, 1), padding=(10, 5)), out_lens)
    a = _mask_time(_bn(a, sd, cp + "4", training), out_lens)
    a = _mask_time(F.hardtanh(a, 0.0, 20.0), out_lens)
    b, c, d, t = a.shape
    h = a.view(b, c * d, t).transpose(1, 2).transpose(0, 1).contiguous()          # (T, B, 1312)  deepspeech.py:135-137
    kind = O.rnn_kind(sd)
    for l in range(O.num_layers(sd)):
        p = f"rnns.{l}."
        if l > 0:                                                                 # SequenceWise(BatchNorm1d) over (T*B, H)
            tt, bb = h.size(0), h.size(1)
            h = _bn(h.view(tt * bb, -1), sd, p + "batch_norm.module", training).view(tt, bb, -1)
        flat: List[Tensor] = []
        for suffix in ("", "_reverse"):
            flat += [sd[p + "rnn.weight_ih_l0" + suffix], sd[p + "rnn.weight_hh_l0" + suffix], sd[p + "rnn.bias_ih_l0" + suffix],
                     sd[p + "rnn.bias_hh_l0" + suffix]]
        hid = flat[1].size(1)
        if packed:
            pk = pack_padded_sequence(h, out_lens)                                # blocks.py:87
            h0 = torch.zeros(2, int(pk.batch_sizes[0]), hid, dtype=h.dtype)
            if kind == "gru":
                y, _ = torch._VF.gru(pk.data, pk.batch_sizes, h0, flat, True, 1, 0.0, training, True)
            else:
                y, _, _ = torch._VF.lstm(pk.data, pk.batch_sizes, (h0, h0.clone()), flat, True, 1, 0.0, training, True)
            y, _ = pad_packed_sequence(torch.nn.utils.rnn.PackedSequence(y, pk.batch_sizes, None, None))   # blocks.py:89
        else:
            assert int(out_lens.min()) == h.size(0), "the un-packed form is only equivalent for batches without padding"
            h0 = torch.zeros(2, h.size(1), hid, dtype=h.dtype)
            if kind == "gru":
                y, _ = torch._VF.gru(h, h0, flat, True, 1, 0.0, training, True, False)
            else:
                y, _, _ = torch._VF.lstm(h, (h0, h0.clone()), flat, True, 1, 0.0, training, True, False)
        h = y.view(y.size(0), y.size(1), 2, -1).sum(2)                            # blocks.py:91-92: fwd + bwd, not concat
    tt, bb = h.size(0), h.size(1)
    z = _bn(h.view(tt * bb, -1), sd, "fc.0.module.0", training)                   # deepspeech.py:103-109
    logits = F.linear(z, sd["fc.0.module.1.weight"]).view(tt, bb, -1)
    return logits.transpose(0, 1), out_lens


def leaf_params(sd: Dict[str, Tensor], dtype=torch.float32) -> Dict[str, Tensor]:
    """Trainable tensors become autograd leaves; buffers (running stats, counters) stay plain tensors updated in place."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            v = v.detach().to(dtype).clone()
            if "running_" not in k:
                v.requires_grad_(True)
        else:
            v = v.detach().clone()
        out[k] = v
    return out


def fit(params: Dict[str, Tensor], inputs: Tensor, targets: Tensor, input_percentages: Tensor, target_sizes: Tensor, packed: bool = True):
    """deepspeech_trainer.py:102-117 on the packed-form model: returns (logits (B,T,C), out_lens, loss tensor with graph)."""
    input_sizes = O.lengths_from_percentages(input_percentages, inputs.size(3))
    out, out_lens = forward(params, inputs, input_sizes, training=True, packed=packed)
    lp = out.transpose(0, 1).float().log_softmax(2)
    loss = F.ctc_loss(lp, targets, out_lens, target_sizes, blank=0, reduction="sum", zero_infinity=False) / inputs.size(0)
    return out, out_lens, loss


def make_optimizer(params: Dict[str, Tensor]):
    """trainers/__main__.py:41-47"""
    return torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)


def train_step(params, optimizer, batch, packed: bool = True):
    """One reference train step (deepspeech_trainer.py:86-97): fit -> zero_grad -> backward -> step.  Returns the loss value."""
    inputs, targets, pct, tsz = batch
    _, _, loss = fit(params, inputs, targets, pct, tsz, packed=packed)
    value = float(loss.detach())
    if O.check_loss_value(value):
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    return value
