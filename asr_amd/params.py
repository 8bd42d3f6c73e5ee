"""Flat parameter / gradient storage for the DeepSpeech module.

All parameters live in ONE contiguous fp32 buffer (and all gradients in a second one with the same
offsets), laid out for the hardware rather than in `state_dict` order:

  * the forward- and reverse-direction RNN tensors of a layer are adjacent, so `[W_ih ; W_ih_reverse]`
    is a zero-copy (2GH, I) view -> both directions' input projections are one MFMA GEMM, and
    `(2, GH, H)` W_hh feeds the two-direction recurrence launches directly;
  * layer blocks are contiguous -> one RCCL all-reduce bucket per layer, in backward order;
  * the whole buffer is one fused AdamW launch.

`state_dict()` keys/shapes stay exactly the reference's: every
`nn.Parameter` of the module is re-pointed at a view of the flat buffer.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

ALIGN = 4  # floats (16 bytes): keeps every view float4-aligned and fwd/reverse tensors adjacent


def layout_order(param_names: List[str], layers: int) -> List[str]:
    names = set(param_names)
    order: List[str] = []
    cp = "conv.seq_module."
    order += [cp + "0.weight", cp + "0.bias", cp + "1.weight", cp + "1.bias", cp + "3.weight", cp + "3.bias", cp + "4.weight",
              cp + "4.bias"]
    for l in range(layers):
        p = f"rnns.{l}."
        if l > 0:
            order += [p + "batch_norm.module.weight", p + "batch_norm.module.bias"]
        for base in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            order += [p + "rnn." + base, p + "rnn." + base + "_reverse"]
    order += ["fc.0.module.0.weight", "fc.0.module.0.bias", "fc.0.module.1.weight"]
    extra = sorted(names - set(order))      # e.g. lookahead params of the unidirectional variant
    missing = [n for n in order if n not in names]
    if missing:
        raise KeyError(f"parameters missing from module: {missing}")
    return order + extra


class FlatParams:
    def __init__(self, module: torch.nn.Module, layers: int, device):
        named = dict(module.named_parameters())
        self.order = layout_order(list(named.keys()), layers)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n in self.order:
            sz = named[n].numel()
            self.offsets[n] = (off, sz)
            off += (sz + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.device = torch.device(device)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.layers = layers
        with torch.no_grad():
            for n in self.order:
                o, sz = self.offsets[n]
                p = named[n]
                self.flat[o:o + sz].copy_(p.detach().reshape(-1).to(self.device, torch.float32))
                p.data = self.flat[o:o + sz].view(p.shape)
        self._named = named
        # BatchNorm buffers: the running statistics live in ONE contiguous fp32 buffer (a train step snapshots / restores them with a single
        # copy: trainers.DeepSpeechTrainer) and the `num_batches_tracked` counters in one int64 buffer (one increment launch per step
        # instead of one per BatchNorm).  The module's buffers are re-pointed at views, so state_dict() / load_state_dict() are unchanged.
        fbufs = [(n, b) for n, b in module.named_buffers() if b.dtype == torch.float32]
        ibufs = [(n, b) for n, b in module.named_buffers() if b.dtype == torch.int64 and b.numel() == 1]
        self.stats = torch.zeros(sum(b.numel() for _, b in fbufs), dtype=torch.float32, device=self.device)
        self.counters = torch.zeros(len(ibufs), dtype=torch.int64, device=self.device)
        self._buf_views: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            off = 0
            for n, b in fbufs:
                v = self.stats[off:off + b.numel()].view(b.shape)
                v.copy_(b.detach().to(self.device))
                b.data = v
                self._buf_views[n] = v
                off += b.numel()
            for i, (n, b) in enumerate(ibufs):
                v = self.counters[i:i + 1].view(b.shape)
                v.copy_(b.detach().to(self.device))
                b.data = v
                self._buf_views[n] = v

    # ------------------------------------------------------------------------------------------
    def owns(self, module: torch.nn.Module) -> bool:
        """True while every parameter still aliases this flat buffer (module.to()/load may break it)."""
        for n, p in module.named_parameters():
            if n not in self.offsets:
                return False
            o, sz = self.offsets[n]
            if p.data_ptr() != self.flat.data_ptr() + 4 * o or p.device != self.flat.device:
                return False
        for n, b in module.named_buffers():
            v = self._buf_views.get(n)
            if v is not None and (b.data_ptr() != v.data_ptr() or b.device != v.device):
                return False
        return True

    def _view(self, buf: torch.Tensor, name: str, shape=None) -> torch.Tensor:
        o, sz = self.offsets[name]
        v = buf[o:o + sz]
        return v.view(shape if shape is not None else self._named[name].shape)

    def _cat(self, buf: torch.Tensor, first: str, shape) -> torch.Tensor:
        o, sz = self.offsets[first]
        o2, sz2 = self.offsets[first + "_reverse"]
        assert o2 == o + sz and sz2 == sz, "fwd/reverse tensors must be adjacent"
        return buf[o:o + 2 * sz].view(shape)

    def tensors(self, module: torch.nn.Module, grads: bool = False) -> Dict[str, torch.Tensor]:
        """{state_dict key -> tensor} incl. buffers (weights case) plus the *_cat views."""
        buf = self.flat_grad if grads else self.flat
        out: Dict[str, torch.Tensor] = {n: self._view(buf, n) for n in self.order}
        for l in range(self.layers):
            p = f"rnns.{l}.rnn."
            wih = self._named[p + "weight_ih_l0"]
            gh, i = wih.shape
            h = self._named[p + "weight_hh_l0"].shape[1]
            out[f"rnns.{l}.wih_cat"] = self._cat(buf, p + "weight_ih_l0", (2 * gh, i))
            out[f"rnns.{l}.whh_cat"] = self._cat(buf, p + "weight_hh_l0", (2, gh, h))
            out[f"rnns.{l}.bih_cat"] = self._cat(buf, p + "bias_ih_l0", (2 * gh,))
            out[f"rnns.{l}.bhh_cat"] = self._cat(buf, p + "bias_hh_l0", (2, gh))
        if not grads:
            for n, b in module.named_buffers():
                out[n] = b
            out["_bn_counters"] = self.counters            # every num_batches_tracked at once (engine.forward)
        return out

    def layer_buckets(self) -> List[Tuple[str, int, int]]:
        """[(bucket name, start, end)] contiguous slices of the flat buffers, in BACKWARD completion
        order: fc, rnns.L-1 ... rnns.0, conv — the order their gradients become final."""
        def span(prefix):
            offs = [(o, o + (sz + ALIGN - 1) // ALIGN * ALIGN) for n, (o, sz) in self.offsets.items() if n.startswith(prefix)]
            return min(a for a, _ in offs), max(b for _, b in offs)
        out = [("fc",) + span("fc.")]
        for l in range(self.layers - 1, -1, -1):
            out.append((f"rnns.{l}",) + span(f"rnns.{l}."))
        out.append(("conv",) + span("conv."))
        return out
