"""Greedy CTC decoder + WER/CER (host-side mirror of asr_deepspeech/decoders/{decoder,greedy_decoder}.py).
`decode()` runs on the GPU (csrc/decode.hip); process_string / convert_to_strings are the host utilities the
reference uses for TARGET strings.  Eval-only, not on the train step.
The edit distance is a small pure-Python DP (the reference imports the `Levenshtein` C package)."""
from __future__ import annotations

import torch


def _edit_distance(a, b) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class Decoder:
    """decoder.py:4-72: labels is {char: index}; index 0 is the CTC blank."""

    def __init__(self, labels, blank_index=0):
        self.labels = labels
        self.int_to_char = dict((i, c) for c, i in labels.items()) if isinstance(labels, dict) else dict(enumerate(labels))
        self.blank_index = blank_index
        space_index = len(self.int_to_char)
        if isinstance(labels, dict) and " " in labels:
            space_index = labels[" "]
        self.space_index = space_index

    def wer(self, s1, s2):
        b = set(s1.split() + s2.split())
        word2char = dict(zip(b, range(len(b))))
        w1 = [chr(word2char[w]) for w in s1.split()]
        w2 = [chr(word2char[w]) for w in s2.split()]
        return _edit_distance(w1, w2)

    def cer(self, s1, s2):
        s1, s2 = s1.replace(" ", ""), s2.replace(" ", "")
        return _edit_distance(s1, s2)

    def decode(self, probs, sizes=None):
        raise NotImplementedError


class GreedyDecoder(Decoder):
    """greedy_decoder.py:6-68: per-frame argmax, collapse repeats, drop blanks."""

    def convert_to_strings(self, sequences, sizes=None, remove_repetitions=False, return_offsets=False):
        """Restates the reference's loop (greedy_decoder.py:12-26) statement for statement: same signature, same nested-list results — the
        evaluation loop and the reference's tests index them as `out[i][0]`."""
        strings, offsets = [], []
        for x in range(len(sequences)):
            seq_len = sizes[x] if sizes is not None else len(sequences[x])
            string, string_offsets = self.process_string(sequences[x], seq_len, remove_repetitions)
            strings.append([string])
            if return_offsets:
                offsets.append([string_offsets])
        return (strings, offsets) if return_offsets else strings

    def process_string(self, sequence, size, remove_repetitions=False):
        string, offsets = "", []
        seq = [int(v) for v in (sequence.tolist() if torch.is_tensor(sequence) else sequence)][: int(size)]
        for i, idx in enumerate(seq):
            char = self.int_to_char.get(idx, "")
            if idx != self.blank_index:
                if remove_repetitions and i != 0 and idx == seq[i - 1]:
                    continue
                string += " " if idx == self.space_index else char
                offsets.append(i)
        return string, torch.tensor(offsets, dtype=torch.int)

    def decode(self, probs, sizes=None):
        """probs (B,T,C) -> ([[str]], [[offsets]]) (greedy_decoder.py:48-68).

        Arg-max, repeat collapse and blank removal run as two HIP kernels (`ds2_greedy_decode_f32`); the host does
        ONE device->host copy of the compacted ids and maps them to characters.  Host tensors are uploaded first —
        there is no CPU implementation of decode()."""
        from .. import ops
        probs = torch.as_tensor(probs)
        if not probs.is_cuda:
            probs = probs.to(_device())
        probs = probs.float()
        if probs.stride(2) != 1:
            probs = probs.contiguous()
        if sizes is not None:
            sizes = torch.as_tensor(sizes)
        ids, offs, lens = ops.greedy_decode(probs, sizes, self.blank_index)
        B, T = ids.shape
        host = torch.cat((ids.reshape(-1), offs.reshape(-1), lens)).cpu()
        if ops.rnn_poison_seen(ids.device):
            # the copy above synchronised with the stream: if the forward that produced `probs` was poisoned (a persistent recurrence
            # launch starved: NaN logits -> empty transcripts), say so here instead of returning garbage silently
            ops.rnn_persistent_check(ids.device)
        ids_h, offs_h, lens_h = host[:B * T].view(B, T), host[B * T:2 * B * T].view(B, T), host[2 * B * T:].tolist()
        strings, offsets = [], []
        for b in range(B):
            n = lens_h[b]
            strings.append(["".join(" " if i == self.space_index else self.int_to_char.get(i, "") for i in ids_h[b, :n].tolist())])
            offsets.append([offs_h[b, :n].clone()])
        return strings, offsets


def _device():
    from .._lib import DS2LibraryError
    if not torch.cuda.is_available():
        raise DS2LibraryError("GreedyDecoder.decode needs a GPU (no CPU fallback exists)")
    return torch.device("cuda", torch.cuda.current_device())
