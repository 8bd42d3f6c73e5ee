"""Building blocks with the reference's names and parameter layout (asr_deepspeech/modules/blocks.py).

Inside `DeepSpeech` these classes are *parameter containers*: `DeepSpeech.forward` runs the fused
MI355X kernel schedule (asr_amd/engine.py) over their parameters and never calls their `forward`.
Stand-alone `forward` of a block is provided where a HIP kernel exists (MaskConv masking,
SequenceWise reshape, InferenceBatchSoftmax, BatchRNN in no-grad mode); there is no CPU path.
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
    """blocks.py:30-56: apply each sub-module, then zero every frame t >= lengths[b].  Inside
    DeepSpeech the mask is fused into the conv / BN epilogues; stand-alone it is a vectorised
    `masked_fill` (no per-sample host sync, unlike the reference's `.item()` loop at :52)."""

    def __init__(self, seq_module):
        super().__init__()
        self.seq_module = seq_module

    def forward(self, x, lengths):
        lens = torch.as_tensor(lengths).to(x.device)
        for module in self.seq_module:
            x = module(x)
            t = torch.arange(x.size(3), device=x.device).view(1, 1, 1, -1)
            x = x.masked_fill(t >= lens.view(-1, 1, 1, 1), 0)
        return x, lengths


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
        """Stand-alone inference/forward-only use: (T,N,I) -> (T,N,H) through the HIP kernels."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("BatchRNN.forward stand-alone has no autograd; train through DeepSpeech.forward")
        if not self._bidirectional:
            raise NotImplementedError("unidirectional BatchRNN has no HIP kernel and asr_amd has no torch fallback by design: see INTEGRATION.md, \"Unidirectional models\" (use asr_deepspeech.modules for this variant)")
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
    """blocks.py:96-132 (only built when bidirectional=False).  Parameter container only: the
    unidirectional variant is outside every BASELINE config and has no HIP kernel."""

    def __init__(self, n_features, context):
        super().__init__()
        assert context > 0
        self.context = context
        self.n_features = n_features
        self.pad = (0, self.context - 1)
        self.conv = nn.Conv1d(self.n_features, self.n_features, kernel_size=self.context, stride=1, groups=self.n_features,
                              padding=0, bias=None)

    def forward(self, x):
        raise NotImplementedError("Lookahead (unidirectional DeepSpeech) has no HIP kernel and asr_amd has no torch fallback by design: see INTEGRATION.md, \"Unidirectional models\" (use asr_deepspeech.modules for this variant)")

    def __repr__(self):
        return f"{self.__class__.__name__}(n_features={self.n_features}, context={self.context})"
