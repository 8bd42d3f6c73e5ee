"""RNN-type registry (mirror of asr_deepspeech/vars.py:17-35, same accepted spellings / errors).

Unlike the reference, importing this module does NOT reseed torch's global RNG (vars.py:13-15 does
that as an import side effect); call `asr_amd.seed_like_reference()` for the same 123456 seed.
"""
from torch import nn

supported_rnns = {"lstm": nn.LSTM, "rnn": nn.RNN, "gru": nn.GRU}
supported_rnns_inv = dict((v, k) for k, v in supported_rnns.items())


def resolve_rnn_type(spec):
    """'nn.LSTM' | 'LSTM' | 'lstm' (and gru/rnn variants) or an nn.Module class -> class."""
    if isinstance(spec, type):
        return spec
    key = str(spec).split(".")[-1].lower()
    try:
        return supported_rnns[key]
    except KeyError as exc:
        raise ValueError(f"Unsupported rnn_type {spec!r}; expected one of {sorted(supported_rnns)}") from exc
