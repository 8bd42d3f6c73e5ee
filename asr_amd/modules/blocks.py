"""Building blocks with the reference's names and parameter layout (asr_deepspeech/modules/blocks.py).

Inside `DeepSpeech` these classes are *parameter containers*: `DeepSpeech.forward` runs the fused
MI355X kernel schedule (asr_amd/engine.py) over their parameters and never calls their `forward`.
Stand-alone `forward` of a block is provided where a HIP kernel exists (MaskConv masking,
SequenceWise reshape, InferenceBatchSoftmax, BatchRNN in no-grad mode); there is no CPU path for the
bidirectional model.  The unidirectional variant (`bidirectional=False` + `Lookahead`; no BASELINE
config, SURVEY.md §2 row 1 "keep as PyTorch fallback, not a kernel target") runs on torch ops.
"""
import torch
import torch.nn as nn

from .. import _lib, ops


class SequenceWise(nn.Module):
    """blocks.py:6-27: (T,N,H) -> (T*N,H) -> module -> (T,N,·)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        t, n = x.size(0), x.size(1)
        x = self.module(x.reshape(t * n, -1))
        return x.view(t, n, -1)

    def __repr__(self):
        return self.__class__.__name__ + " (\n" + self.module.__repr__() + ")"


class MaskConv(nn.Module):
    """blocks.py:30-56: apply each sub-module, then zero every frame t >= lengths[b].  Inside DeepSpeech the mask is fused into the conv /
    BatchNorm epilogues of the kernel schedule (asr_amd/engine.py).  Stand-alone `forward` (forward-only, like BatchRNN's) runs the SAME HIP
    kernels for the one stack they are written for — DeepSpeech's Conv2d(1,32,(41,11),s(2,2),p(20,5)) / BatchNorm2d / Hardtanh(0,20) /
    Conv2d(32,32,(21,11),s(2,1),p(10,5)) / BatchNorm2d / Hardtanh(0,20) (deepspeech.py:60-67) — and raises for anything else: there is no
    torch (MIOpen) fallback behind this class either."""

    def __init__(self, seq_module):
        super().__init__()
        self.seq_module = seq_module

    def _is_ds2_stack(self):
        m = list(self.seq_module)
        if len(m) != 6 or not (isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.BatchNorm2d) and isinstance(m[2], nn.Hardtanh)
                               and isinstance(m[3], nn.Conv2d) and isinstance(m[4], nn.BatchNorm2d) and isinstance(m[5], nn.Hardtanh)):
            return False
        c1, c2 = m[0], m[3]
        return ((c1.in_channels, c1.out_channels, tuple(c1.kernel_size), tuple(c1.stride), tuple(c1.padding)) == (1, 32, (41, 11), (2, 2), (20, 5))
                and (c2.in_channels, c2.out_channels, tuple(c2.kernel_size), tuple(c2.stride), tuple(c2.padding)) == (32, 32, (21, 11), (2, 1), (10, 5))
                and all((h.min_val, h.max_val) == (0, 20) for h in (m[2], m[5])) and c1.bias is not None and c2.bias is not None)

    def forward(self, x, lengths):
        """x (B,1,F,T_in) fp32 on the GPU, lengths (B,) = frames to keep in the OUTPUT (blocks.py:42-56) -> ((B,32,D2,T), lengths)."""
        if not self._is_ds2_stack():
            # any OTHER stack: the reference container's own semantics (blocks.py:42-56: run each module, zero everything beyond each
            # utterance's length after it) in plain torch — a generic container, not the train-step path (DeepSpeech never builds one)
            for module in self.seq_module:
                x = module(x)
                mask = torch.zeros(x.size(), dtype=torch.bool, device=x.device)
                for i, length in enumerate(lengths):
                    length = int(length)
                    if mask[i].size(2) - length > 0:
                        mask[i].narrow(2, length, mask[i].size(2) - length).fill_(True)
                x = x.masked_fill(mask, 0)
            return x, lengths
        # DeepSpeech's own stack: the HIP kernels, never a torch / MIOpen fallback
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("MaskConv.forward stand-alone has no autograd; train through DeepSpeech.forward")
        if not x.is_cuda:
            raise _lib.DS2LibraryError("MaskConv.forward: GPU tensor required (no CPU fallback for DeepSpeech's conv stack)")
        m = list(self.seq_module)
        lens_dev = torch.as_tensor(lengths).to(torch.int32).to(x.device)
        x = x.contiguous().float()
        wpk1, wpk2, _ = ops.conv_pack(m[0].weight.detach(), m[3].weight.detach())

        def stats(y, bn):
            if self.training:
                return ops.bn2d_stats(y, bn.running_mean, bn.running_var)
            return bn.running_mean, bn.running_var
        y1 = ops.conv1_fwd(x, wpk1, m[0].bias.detach(), lens_dev)
        a1 = ops.bn2d_act_fwd(y1, lens_dev, *stats(y1, m[1]), m[1].weight.detach(), m[1].bias.detach())
        y2 = ops.conv2_fwd(a1, wpk2, m[3].bias.detach(), lens_dev)
        a2 = ops.bn2d_act_fwd(y2, lens_dev, *stats(y2, m[4]), m[4].weight.detach(), m[4].bias.detach())
        return a2, lengths


class InferenceBatchSoftmax(nn.Module):
    """blocks.py:59-64: identity in training, softmax over the last dim in eval."""

    def forward(self, input_):
        if self.training:
            return input_
        if not input_.is_cuda:
            raise _lib.DS2LibraryError("InferenceBatchSoftmax: GPU tensor required (no CPU fallback)")
        shp = input_.shape
        flat = input_.reshape(-1, shp[-1]).float().contiguous()
        return ops.softmax_rows(flat).view(shp)


class BatchRNN(nn.Module):
    """blocks.py:67-93.  Holds `batch_norm` (SequenceWise(BatchNorm1d)) and `rnn` (nn.GRU / nn.LSTM
    used purely as the parameter container => identical init and state_dict keys)."""

    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM, bidirectional=False, batch_norm=True):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self._bidirectional = bidirectional
        self.batch_norm = SequenceWise(nn.BatchNorm1d(input_size)) if batch_norm else None
        self.rnn = rnn_type(input_size=input_size, hidden_size=hidden_size, bidirectional=bidirectional, bias=True)
        self.num_directions = 2 if bidirectional else 1

    def flatten_parameters(self):
        pass  # parameters are already one flat buffer (asr_amd/params.py)

    def forward(self, x, output_lengths):
        """Stand-alone inference/forward-only use: (T,N,I) -> (T,N,H) through the HIP kernels (bidirectional); the unidirectional variant —
        outside every BASELINE config, not a kernel target — runs the reference's op sequence on torch ops, autograd
        included: BatchNorm1d over all rows, packed sequence through the cell, zero rows beyond each length (blocks.py:84-93)."""
        if not self._bidirectional:
            if self.batch_norm is not None:
                x = self.batch_norm(x)
            packed = nn.utils.rnn.pack_padded_sequence(x, torch.as_tensor(output_lengths).cpu())
            return nn.utils.rnn.pad_packed_sequence(self.rnn(packed)[0])[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("BatchRNN.forward stand-alone has no autograd; train through DeepSpeech.forward")
        if not x.is_cuda:
            raise _lib.DS2LibraryError("BatchRNN.forward: GPU tensor required (no CPU fallback)")
        T, N, I = x.shape
        H = self.hidden_size
        gates = {nn.GRU: 3, nn.LSTM: 4}[type(self.rnn)]
        lens_dev = torch.as_tensor(output_lengths, dtype=torch.int32).to(x.device)
        flat = x.reshape(T * N, I).float().contiguous()
        if self.batch_norm is not None:
            bn = self.batch_norm.module
            if self.training:
                mean, var = ops.colstats(flat, bn.running_mean, bn.running_var)
            else:
                mean, var = bn.running_mean, bn.running_var
            flat = ops.bn1d_apply(flat, mean, var, bn.weight.detach(), bn.bias.detach())
        r = self.rnn
        wih = torch.cat([r.weight_ih_l0.detach(), r.weight_ih_l0_reverse.detach()], 0).contiguous()
        bih = torch.cat([r.bias_ih_l0.detach(), r.bias_ih_l0_reverse.detach()], 0).contiguous()
        whh = torch.stack([r.weight_hh_l0.detach(), r.weight_hh_l0_reverse.detach()], 0).contiguous()
        bhh = torch.stack([r.bias_hh_l0.detach(), r.bias_hh_l0_reverse.detach()], 0).contiguous()
        gx = ops.gemm(flat, wih, transB=True, bias=bih)
        wpf, _ = ops.rnn_pack(gates, whh)
        hbuf, _ = ops.rnn_fwd(gates, gx, wpf, bhh, lens_dev, T, N, H)
        y, _, _ = ops.add_colstats(hbuf[:, :H], hbuf[:, H:])
        return y.view(T, N, H)


class Lookahead(nn.Module):
    """blocks.py:96-132 (only built when bidirectional=False): y[t] = sum_{k < context} w[:, k] * x[t + k] per feature, zeros beyond the end
    (Wang et al. 2016).  The unidirectional variant is outside every BASELINE config and is not a kernel target (SURVEY.md §2 row 1:
    "keep as PyTorch fallback"): a depthwise torch convolution over time, as in the reference."""

    def __init__(self, n_features, context):
        super().__init__()
        assert context > 0
        self.context = context
        self.n_features = n_features
        self.pad = (0, self.context - 1)
        self.conv = nn.Conv1d(self.n_features, self.n_features, kernel_size=self.context, stride=1, groups=self.n_features,
                              padding=0, bias=None)

    def forward(self, x):
        """(T, N, H) -> (T, N, H)"""
        nht = torch.nn.functional.pad(x.permute(1, 2, 0), self.pad, value=0)          # (N, H, T + context - 1): future frames, zero-padded
        return self.conv(nht).permute(2, 0, 1).contiguous()

    def __repr__(self):
        return f"{self.__class__.__name__}(n_features={self.n_features}, context={self.context})"
