"""`CTCLoss` — call-compatible with `torch.nn.CTCLoss` as the reference wires it
(trainers/__main__.py:53, called at trainers/deepspeech_trainer.py:111) and backed by the fused
log-softmax + CTC HIP kernels (asr_amd/csrc/ctc.hip).

Input may be log-probabilities (T,B,C) *or* raw logits: the kernel re-normalises each row (for
log-softmaxed input the row log-sum-exp is 0, so results are identical to torch's).  The gradient
returned w.r.t. the input is `softmax - occupancy`, exactly what aten's `_ctc_loss_backward`
produces for log-softmaxed input.
"""
from __future__ import annotations

import torch

from . import _lib, ops


def _prep_targets_host(targets, target_lengths):
    """CPU int32 tensors (flat targets, per-utterance offsets, target lengths) and the longest target."""
    tl = torch.as_tensor(target_lengths).to(torch.int32).cpu()
    t = torch.as_tensor(targets).to(torch.int32).cpu()
    if t.dim() == 2:  # padded (B, S) form of torch.nn.CTCLoss
        t = torch.cat([t[i, : int(tl[i])] for i in range(t.size(0))]) if t.numel() else t.reshape(-1)
    off = torch.zeros(tl.numel(), dtype=torch.int32)
    if tl.numel() > 1:
        off[1:] = torch.cumsum(tl, 0)[:-1].to(torch.int32)
    max_u = int(tl.max()) if tl.numel() else 0
    if t.numel() == 0:
        t = torch.zeros(1, dtype=torch.int32)
    return t, off, tl, max_u


def _prep_targets(targets, target_lengths, device):
    tl = torch.as_tensor(target_lengths).to(torch.int32).cpu()
    t = torch.as_tensor(targets).to(torch.int32).cpu()
    if t.dim() == 2:  # padded (B, S) form of torch.nn.CTCLoss
        t = torch.cat([t[i, : int(tl[i])] for i in range(t.size(0))]) if t.numel() else t.reshape(-1)
    off = torch.zeros(tl.numel(), dtype=torch.int32)
    if tl.numel() > 1:
        off[1:] = torch.cumsum(tl, 0)[:-1].to(torch.int32)
    max_u = int(tl.max()) if tl.numel() else 0
    if t.numel() == 0:
        t = torch.zeros(1, dtype=torch.int32)
    return t.to(device), off.to(device), tl.to(device), max_u


class _CTCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, targets, input_lengths, target_lengths, want_grad):
        T, B, C = acts.shape
        dev = acts.device
        il = torch.as_tensor(input_lengths).to(torch.int32).to(dev)
        tg, off, tl, max_u = _prep_targets(targets, target_lengths, dev)
        x = acts.float()
        if not (x.stride(2) == 1 and x.stride(0) == B * x.stride(1)):
            x = x.contiguous()
        nll, grad = ops.ctc_loss(x, tg, off, il, tl, max_u, 1.0, want_grad=want_grad)
        ctx.grad = grad
        return nll

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        if grad is None:
            raise RuntimeError("CTC gradient was not requested in forward")
        return grad * g.view(1, -1, 1), None, None, None, None


class CTCLoss(torch.nn.Module):
    def __init__(self, blank: int = 0, reduction: str = "mean", zero_infinity: bool = False):
        super().__init__()
        if blank != 0:
            raise ValueError("the HIP CTC kernel fixes blank = 0 (the reference's only configuration)")
        if reduction not in ("none", "mean", "sum"):
            raise ValueError(reduction)
        self.blank, self.reduction, self.zero_infinity = blank, reduction, zero_infinity

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        if not log_probs.is_cuda:
            raise _lib.DS2LibraryError("asr_amd.CTCLoss needs GPU input (no CPU fallback; the CPU oracle is test-only)")
        want_grad = torch.is_grad_enabled() and log_probs.requires_grad
        nll = _CTCFunction.apply(log_probs, targets, input_lengths, target_lengths, want_grad)
        if self.zero_infinity:
            nll = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
        if self.reduction == "none":
            return nll
        if self.reduction == "sum":
            return nll.sum()
        tl = torch.as_tensor(target_lengths).to(nll.device).clamp_min(1).to(nll.dtype)
        return (nll / tl).mean()
