"""`DeepSpeech` with the reference's constructor, `forward`, `get_seq_lens`, `get_loader`, eval
`__call__`, `finetune_from` and state_dict keys (asr_deepspeech/modules/deepspeech.py:24-288), whose
forward/backward run on the hand-written MI355X kernels (asr_amd/engine.py, libds2hip).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from .. import _lib, engine, ops
from ..device import resolve_device
from ..params import FlatParams
from ..vars import resolve_rnn_type
from .blocks import BatchRNN, InferenceBatchSoftmax, Lookahead, MaskConv, SequenceWise


def _read_labels(label_path):
    """deepspeech.py:48 — {char: row index} from the `label` column of labels.csv (row 0 = CTC blank)."""
    import pandas as pd
    return dict([(v, k) for k, v in pd.read_csv(label_path).to_dict()["label"].items()])


class _DS2Function(torch.autograd.Function):
    """Whole-network autograd node: forward = kernel schedule, backward = hand-derived schedule.
    Parameters are passed as inputs so that `loss.backward()` populates `p.grad` like the reference."""

    @staticmethod
    def forward(ctx, model, x, lens_dev, *params):
        W = model._flat.tensors(model)
        logits, saved = engine.forward(W, model._cfg, x, lens_dev, training=model.training, save=True)
        ctx.model, ctx.saved = model, saved
        # backward runs on autograd's worker thread: it must go through the recurrence context of THIS thread (whose enable switches the
        # trainer set, and whose starvation record / cooldown the trainer reads), not through a fresh one of the worker's own
        ctx.rnn_ctx, ctx.device = ops.rnn_ctx(x.device), x.device
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        W = model._flat.tensors(model)
        flat = model._flat
        # autograd installs the returned tensors as `p.grad` without a copy, i.e. p.grad then ALIASES the flat
        # gradient buffer (the zero-copy normal case: zero_grad(set_to_none=True) -> backward -> step).  If some
        # p.grad is still alive (gradient accumulation, zero_grad(set_to_none=False)) the kernels must not
        # overwrite it: compute into a scratch buffer and let autograd add.
        accumulating = any(p.grad is not None for p in model.parameters())
        if accumulating:
            if model._on_bucket is not None:
                # the data-parallel reducer all-reduces the flat gradient buffer bucket by bucket as backward fills it; gradients that
                # autograd adds up outside that buffer would silently stay un-reduced
                raise RuntimeError("asr_amd.DeepSpeech: gradient accumulation (p.grad still set when backward runs) is not supported "
                                   "under data parallelism; call optimizer.zero_grad() (set_to_none=True) before every backward")
            keep, flat.flat_grad = flat.flat_grad, torch.empty_like(flat.flat_grad)
        try:
            Gr = flat.tensors(model, grads=True)
            with ops.use_rnn_ctx(ctx.rnn_ctx, ctx.device), torch.cuda.device(ctx.device):
                engine.backward(W, Gr, model._cfg, ctx.saved, dlogits.contiguous(),
                                on_bucket=None if accumulating else model._on_bucket)
        finally:
            if accumulating:
                flat.flat_grad = keep
        ctx.saved = None
        grads = tuple(Gr[n] for n in model._param_names)
        return (None, None, None) + grads


class DeepSpeech(nn.Module):
    def __init__(self, audio_conf, decoder, label_path, id="asr", rnn_type="nn.LSTM", rnn_hidden_size=768, rnn_hidden_layers=5,
                 bidirectional=True, context=20, version="0.0.1", model_path=None, restart_from=None):
        super().__init__()
        self.version = version
        self.id = id
        self.decoder, self.audio_conf = decoder, audio_conf
        self.context = context
        self.rnn_hidden_size = rnn_hidden_size
        self.rnn_hidden_layers = rnn_hidden_layers
        self.rnn_type = resolve_rnn_type(rnn_type)
        self.labels = _read_labels(label_path)
        self.bidirectional = bidirectional
        self.sample_rate = self.audio_conf.sample_rate
        self.window_size = self.audio_conf.window_size
        self.num_classes = len(self.labels)
        self.model_path = model_path
        self.build_network()
        from ..decoders import GreedyDecoder
        self.decoder = GreedyDecoder(self.labels)
        self._flat: Optional[FlatParams] = None
        self._on_bucket = None          # DP hook: called as each layer's gradients become final
        self._param_names = [n for n, _ in self.named_parameters()]
        kind = {nn.GRU: "gru", nn.LSTM: "lstm"}.get(self.rnn_type)
        self._cfg = engine.ModelCfg(rnn=kind or "unsupported", hidden=rnn_hidden_size, layers=rnn_hidden_layers,
                                    classes=self.num_classes, freq=int(math.floor(self.sample_rate * self.window_size / 2) + 1))

    # -- structure (deepspeech.py:58-110) -----------------------------------------------------------
    def build_network(self):
        self.conv = MaskConv(nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=(41, 11), stride=(2, 2), padding=(20, 5)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20, inplace=True),
            nn.Conv2d(32, 32, kernel_size=(21, 11), stride=(2, 1), padding=(10, 5)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20, inplace=True),
        ))
        f = int(math.floor((self.sample_rate * self.window_size) / 2) + 1)
        f = int(math.floor(f + 2 * 20 - 41) / 2 + 1)
        f = int(math.floor(f + 2 * 10 - 21) / 2 + 1)
        rnn_input_size = f * 32
        rnns = [("0", BatchRNN(rnn_input_size, self.rnn_hidden_size, rnn_type=self.rnn_type, bidirectional=self.bidirectional,
                               batch_norm=False))]
        for i in range(self.rnn_hidden_layers - 1):
            rnns.append((str(i + 1), BatchRNN(self.rnn_hidden_size, self.rnn_hidden_size, rnn_type=self.rnn_type,
                                              bidirectional=self.bidirectional)))
        self.rnns = nn.Sequential(OrderedDict(rnns))
        self.lookahead = (nn.Sequential(Lookahead(self.rnn_hidden_size, context=self.context), nn.Hardtanh(0, 20, inplace=True))
                          if not self.bidirectional else None)
        fully_connected = nn.Sequential(nn.BatchNorm1d(self.rnn_hidden_size),
                                        nn.Linear(self.rnn_hidden_size, self.num_classes, bias=False))
        self.fc = nn.Sequential(SequenceWise(fully_connected))
        self.inference_softmax = InferenceBatchSoftmax()

    # -- flat parameter storage -------------------------------------------------------------------
    @property
    def precision(self) -> str:
        """"fp32" (default; the parity path) or "bf16" (bf16 MFMA operands for the input-to-hidden GEMMs, fp32
        accumulation / state / BN / CTC — the BASELINE configs[2],[4] setting)."""
        return self._cfg.precision

    @precision.setter
    def precision(self, value: str):
        if value not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._cfg.precision = value

    def _ensure_flat(self, device):
        device = torch.device(device)
        if self._flat is None or self._flat.device != device or not self._flat.owns(self):
            old = self._flat
            self._flat = FlatParams(self, self.rnn_hidden_layers, device)
            for b_name, b in list(self.named_buffers()):
                if b.device != device:
                    mod, _, leaf = b_name.rpartition(".")
                    setattr(self.get_submodule(mod), leaf, b.to(device))
            del old
        return self._flat

    def flat_parameters(self):
        """(flat params, flat grads) fp32 buffers — the fused optimizer / RCCL all-reduce operate on these."""
        assert self._flat is not None, "call model.to(device) and run one forward (or _ensure_flat) first"
        return self._flat.flat, self._flat.flat_grad

    def finetune_from(self, model_path, nlayers=1):
        """deepspeech.py:112-128: shape-checked partial load, then freeze all but the last n tensors."""
        state_dict = self.state_dict()
        loaded = torch.load(model_path, map_location="cpu")
        loaded = loaded.get("state_dict", loaded) if isinstance(loaded, dict) else loaded
        for k, v in loaded.items():
            if k in state_dict and state_dict[k].shape == v.shape:
                state_dict[k] = v
            else:
                print(k, state_dict.get(k, torch.empty(0)).shape, v.shape)
        self.load_state_dict(state_dict)
        print(f"finetune from {model_path} (last {nlayers} layers)")
        if nlayers is not None:
            for m in list(self.parameters())[:-nlayers]:
                m.requires_grad = False

    # -- forward (deepspeech.py:130-149) ----------------------------------------------------------
    def forward(self, x: torch.Tensor, lengths: torch.Tensor):
        lengths = torch.as_tensor(lengths).cpu().int()
        output_lengths = self.get_seq_lens(lengths)
        if not self.bidirectional:
            return self._forward_unidirectional(x, output_lengths)
        if self._cfg.rnn == "unsupported":
            raise NotImplementedError("only GRU / LSTM cells have MI355X kernels (asr_deepspeech.vars.supported_rnns lists nn.RNN too: use "
                                      "asr_deepspeech.modules.DeepSpeech for that cell)")
        if not x.is_cuda:
            raise _lib.DS2LibraryError(
                "asr_amd.DeepSpeech.forward needs GPU input: the MI355X HIP kernels are the only implementation. "
                "(The CPU restatement used for parity checks lives in oracle/ and is test infrastructure.)")
        _lib.load()
        self._ensure_flat(x.device)
        lens_dev = output_lengths.to(x.device, non_blocking=True)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            params = [p for _, p in self.named_parameters()]
            logits = _DS2Function.apply(self, x, lens_dev, *params)
        else:
            if not self.training and ops.rnn_poison_seen(x.device):
                # an EARLIER inference forward of this process was handed NaN logits (below) and nobody has settled the starvation since:
                # settle it now — raises DS2LibraryError naming the launch, clears the record and moves the next recurrence calls onto the
                # step kernels, so a caller that only ever calls forward() sees the failure once and then keeps working
                ops.rnn_persistent_check(x.device)
            W = self._flat.tensors(self)
            logits, _ = engine.forward(W, self._cfg, x, lens_dev, training=self.training, save=False)
            if not self.training:
                # A persistent recurrence launch that could not get all of its workgroups resident leaves invalid activations.  Inference
                # must never return them, and must not pay a device synchronisation per forward either (streaming / batched inference
                # would lose all host-device overlap), nor consume the starvation record of a trainer's un-settled step: a kernel in
                # stream order turns the logits into NaN if a launch before it starved (fp32 shapes take the persistent kernels too),
                # and the record stays for the next check at a natural sync point (evaluate() below, the trainer's step / synchronize,
                # the decoders' device-to-host copy) — or, for callers that only ever call forward(), the NEXT forward: the poison kernel
                # raises a pinned host flag that the test above reads without synchronising.
                ops.rnn_poison_if_starved(logits)
        out = logits.transpose(0, 1)            # (B,T,C) view, like the reference's x.transpose(0, 1)
        out = self.inference_softmax(out)       # identity in train, HIP softmax in eval
        return out, output_lengths

    def _forward_unidirectional(self, x, output_lengths):
        """`bidirectional=False` (+ Lookahead, deepspeech.py:83-101, :142-143): not on any BASELINE configuration and not a kernel target
        ("keep as PyTorch fallback") — the reference's op sequence on torch ops with torch autograd, on whatever device
        the module lives on, so that a unidirectional checkpoint of the reference loads, evaluates and fine-tunes behind the same class.
        The MI355X kernels, the fused `trainer.step` and the data-parallel reducer are for the bidirectional model only."""
        if not getattr(self, "_uni_note", False):
            self._uni_note = True
            print("[asr_amd] unidirectional DeepSpeech: this variant runs on torch ops (no MI355X kernels; SURVEY.md §2 row 1)", flush=True)
        keep = (torch.arange(int(output_lengths.max()) if output_lengths.numel() else 0, device=x.device).view(1, 1, 1, -1)
                < output_lengths.to(x.device).view(-1, 1, 1, 1))
        h = x
        for m in self.conv.seq_module:                       # MaskConv (blocks.py:42-56): every stage's output is zeroed beyond each utterance
            h = m(h)
            t = h.size(3)
            live = keep[..., :t] if keep.size(3) >= t else torch.nn.functional.pad(keep, (0, t - keep.size(3)))
            h = h * live
        b, c, d, t = h.shape
        h = h.reshape(b, c * d, t).permute(2, 0, 1).contiguous()                     # (T, N, c*D + d)
        for rnn in self.rnns:
            h = rnn(h, output_lengths)
        h = self.lookahead(h)
        h = self.fc(h).transpose(0, 1)
        if not self.training:
            h = torch.softmax(h, dim=-1)
        return h, output_lengths

    def get_loader(self, manifest, batch_size, num_workers, caching=False):
        from ..data import get_loader
        return get_loader(self.audio_conf, self.labels, manifest, batch_size, num_workers, caching=caching)

    def get_seq_lens(self, input_length: torch.Tensor) -> torch.Tensor:
        """deepspeech.py:275-288: true division per Conv2d on the time axis, one truncation at the end."""
        seq_len = input_length
        for m in self.conv.modules():
            if isinstance(m, nn.modules.conv.Conv2d):
                seq_len = (seq_len + 2 * m.padding[1] - m.dilation[1] * (m.kernel_size[1] - 1) - 1) / m.stride[1] + 1
        return seq_len.int()

    # -- evaluation loop (deepspeech.py:161-273) --------------------------------------------------
    def evaluate(self, loader=None, manifest=None, batch_size=None, device="auto", num_workers=32, verbose=False, half=False,
                 output_file=None, main_proc=True, **_unused):
        device = resolve_device(device)
        with torch.no_grad():
            if loader is None:
                loader, _ = self.get_loader(manifest=manifest, batch_size=batch_size, num_workers=num_workers)
            decoder = self.decoder
            self.eval()
            self.to(device)
            total_cer = total_wer = num_tokens = num_chars = 0
            output_data = []
            # per-utterance report (deepspeech.py:224-244): best / last / worst transcript by CER and a 10-bucket CER histogram
            hist = [0] * 10
            best = (float("inf"), "")
            worst = (-1.0, "")
            last_str = ""
            for data in loader:
                inputs, targets, input_percentages, target_sizes = data
                input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
                inputs = inputs.to(device)
                split_targets, offset = [], 0
                for size in target_sizes:
                    split_targets.append(targets[offset:offset + size])
                    offset += size
                out, output_sizes = self.forward(inputs, input_sizes)
                decoded_output, _ = decoder.decode(out, output_sizes)   # (copies to the host: the device is idle behind it)
                ops.rnn_persistent_check(inputs.device)                 # raise if a persistent recurrence of this batch starved (the logits are NaN then)
                target_strings = decoder.convert_to_strings(split_targets)
                if output_file is not None:
                    output_data.append((out.detach().cpu().numpy(), output_sizes.numpy(), target_strings))
                for i in range(len(target_strings)):
                    transcript, reference = decoded_output[i][0], target_strings[i][0]
                    wer_inst, cer_inst = decoder.wer(transcript, reference), decoder.cer(transcript, reference)
                    total_wer += wer_inst
                    total_cer += cer_inst
                    n_tok, n_chr = len(reference.split()), len(reference.replace(" ", ""))
                    num_tokens += n_tok
                    num_chars += n_chr
                    wer_pct = min(100.0 * wer_inst / max(n_tok, 1), 100.0)
                    cer_pct = min(100.0 * cer_inst / max(n_chr, 1), 100.0)
                    hist[min(int(cer_pct // 10), 9)] += 1
                    last_str = f"Ref:{reference.lower()}\nHyp:{transcript.lower()}\nWER:{wer_pct}  - CER:{cer_pct}"
                    if cer_pct < best[0]:
                        best = (cer_pct, last_str)
                    if cer_pct > worst[0]:
                        worst = (cer_pct, last_str)
                    if verbose:
                        print(last_str)
            wer = float(total_wer) / max(num_tokens, 1)
            cer = float(total_cer) / max(num_chars, 1)
            if main_proc and output_file is not None:
                # same sections as the reference's report (deepspeech.py:249-271); its histogram is drawn by the third-party ascii_graph
                # package, this one by the loop below (same buckets and counts)
                peak = max(max(hist), 1)
                bars = [f"{k * 10:>3}-{k * 10 + 10:<3} | {'#' * round(40 * v / peak):<40} {v}" for k, v in enumerate(hist)]
                with open(output_file, "w") as f:
                    f.write("\n".join([f"===== {wer * 100:.2f}/{cer * 100:.2f} =====", "----- BEST -----", best[1], "----- LAST -----", last_str,
                                       "----- WORST -----", worst[1], "CER histogram"] + bars
                                      + ["=============================================\n"]))
                print(f"saved output to {output_file}")
            return wer * 100, cer * 100, output_data

    def __call__(self, *args, **kwargs):
        """The reference overrides __call__ with its eval loop (deepspeech.py:161).  Keep that calling
        convention (`model(loader=..., device=...)`) and fall through to nn.Module for tensors."""
        if args and torch.is_tensor(args[0]):
            return super().__call__(*args, **kwargs)
        return self.evaluate(*args, **kwargs)
