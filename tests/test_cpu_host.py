"""CPU (-m "not gpu"): host logic, API surface, C-ABI export check, data-parallel reducer over gloo."""
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audio_conf():
    return SimpleNamespace(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming", speed_volume_perturb=False,
                           spec_augment=False, noise_dir=None, noise_prob=0.4, noise_levels=(0.0, 0.5))


def make(rnn="gru", hidden=32, layers=2, classes=7):
    import pandas as pd
    from asr_amd import DeepSpeech
    chars = ["_", "'"] + list("abcdefghijklmnopqrstuvwxyz") + ["|"]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "labels.csv")
        pd.DataFrame({"label": chars[:classes]}).to_csv(path, index=False)
        return DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=path, rnn_type=rnn, rnn_hidden_size=hidden,
                          rnn_hidden_layers=layers, bidirectional=True)


def test_library_loads_and_exports_every_declared_symbol():
    """The in-tree libds2hip.so loads (no GPU needed) and exports exactly what include/ds2hip.h declares."""
    from asr_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ds2hip.h")).read()
    declared = set(re.findall(r"\b(ds2_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.ds2_version()
    assert _lib.conv_dims(161, 1001) == (81, 41, 501)
    assert lib.ds2_ctc_workspace_bytes(10, 2, 3) > 0 and lib.ds2_conv_packed_floats(1) == 32 * 232 * 32


def test_workspace_queries_and_splitk_policy():
    """Host-side sizing logic (no GPU): workspace queries are consistent and the split-K policy is sane on the train step's shapes."""
    from asr_amd import _lib
    from asr_amd.ops import _pick_splitk
    lib = _lib.load()
    assert lib.ds2_greedy_decode_workspace_bytes(64, 1001) == 64 * 1001 * 4
    assert lib.ds2_cast_bf16_both_workspace_bytes(32064, 6144) == 501 * 6144 * 2 * 4
    assert lib.ds2_gemm_bf16_workspace_bytes(100, 200, 2, 4) == 2 * 4 * 100 * 200 * 4 and lib.ds2_gemm_bf16_workspace_bytes(100, 200, 2, 1) == 0
    assert _pick_splitk(32064, 6144, 1024) == 1 and _pick_splitk(32064, 1024, 6144) == 1      # output-heavy: never split
    for m, n, k in [(6144, 1024, 32064), (2048, 1024, 32000), (1024, 1024, 32000), (6144, 1312, 32064)]:
        s = _pick_splitk(m, n, k)
        tiles = -(-m // 256) * -(-n // 256) * s
        assert 1 < s <= 16 and k // s >= 1024 and tiles >= 192, (m, n, k, s)                    # long-K weight gradients fill the chip
    assert _pick_splitk(70, 45, 33) == 1 and _pick_splitk(29, 96, 1000) == 1


def test_state_dict_matches_reference_manifest():
    man = json.load(open(f"{GOLDEN}/state_manifest.json"))
    for key, (rnn, h, l, c) in {"gru_32x2_c7": ("gru", 32, 2, 7), "lstm_24x2_c7": ("lstm", 24, 2, 7)}.items():
        m = make(rnn, h, l, c)
        sd = m.state_dict()
        assert list(sd.keys()) == list(man[key]["keys"].keys())
        assert all(list(v.shape) == man[key]["keys"][k] for k, v in sd.items())
        assert [n for n, _ in m.named_parameters()] == man[key]["param_order"]
        assert sum(p.numel() for p in m.parameters()) == man[key]["param_count"]
        assert m.num_classes == c


def test_get_seq_lens_and_length_recovery_golden():
    z = np.load(f"{GOLDEN}/lengths.npz")
    m = make()
    assert np.array_equal(m.get_seq_lens(torch.arange(1, 2002, dtype=torch.int32)).numpy(), z["seq_lens"])


def test_collate_golden():
    from asr_amd import _collate_fn
    import det
    z = np.load(f"{GOLDEN}/collate.npz")
    specs = [torch.from_numpy(det.unitvar((161, t), 20 + i)) for i, t in enumerate((15, 20, 9))]
    inputs, targets, pct, sizes = _collate_fn(list(zip(specs, [[4, 5], [1, 2, 3], [6]])))
    assert torch.equal(inputs, torch.from_numpy(z["inputs"])) and np.array_equal(targets.numpy(), z["targets"])
    assert np.array_equal(pct.numpy(), z["pct"]) and np.array_equal(sizes.numpy(), z["sizes"])
    assert targets.dtype == torch.int32 and sizes.dtype == torch.int32


def test_check_loss_and_resolvers():
    from asr_amd import check_loss, resolve_device, resolve_rnn_type
    assert check_loss(torch.tensor(1.0), 1.0)[0]
    assert not check_loss(torch.tensor(float("inf")), float("inf"))[0]
    assert not check_loss(torch.tensor(float("nan")), float("nan"))[0]
    assert not check_loss(torch.tensor(-1.0), -1.0)[0]
    assert resolve_rnn_type("nn.GRU") is torch.nn.GRU and resolve_rnn_type("lstm") is torch.nn.LSTM
    assert resolve_rnn_type(torch.nn.GRU) is torch.nn.GRU
    with pytest.raises(ValueError):
        resolve_rnn_type("transformer")
    assert resolve_device("cpu").type == "cpu" and resolve_device(None).type in ("cpu", "cuda")
    with pytest.raises(ValueError):
        resolve_device("tpu")


def test_product_has_no_cpu_path():
    """forward / CTCLoss on CPU tensors must fail loudly (the CPU restatement lives in oracle/, test-only)."""
    from asr_amd import CTCLoss, _lib
    m = make()
    with pytest.raises(_lib.DS2LibraryError):
        m.forward(torch.zeros(2, 1, 161, 40), torch.tensor([40, 20]))
    with pytest.raises(_lib.DS2LibraryError):
        CTCLoss(reduction="sum")(torch.zeros(5, 2, 7), torch.tensor([1, 2], dtype=torch.int32), torch.tensor([5, 5]), torch.tensor([1, 1]))
    import asr_amd
    src = "".join(open(os.path.join(ROOT, "asr_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "asr_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_flat_params_layout_and_buckets():
    from asr_amd.params import FlatParams
    m = make("gru", 32, 3, 7)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m, 3, "cpu")
    assert fp.owns(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])                               # values preserved, keys unchanged
    W = fp.tensors(m)
    r = m.rnns[1].rnn
    assert torch.equal(W["rnns.1.wih_cat"], torch.cat([r.weight_ih_l0, r.weight_ih_l0_reverse], 0))
    assert torch.equal(W["rnns.1.whh_cat"], torch.stack([r.weight_hh_l0, r.weight_hh_l0_reverse], 0))
    assert W["rnns.1.wih_cat"].data_ptr() == r.weight_ih_l0.data_ptr()  # zero-copy
    b = fp.layer_buckets()
    assert [n for n, _, _ in b] == ["fc", "rnns.2", "rnns.1", "rnns.0", "conv"]
    covered = sorted((a, e) for _, a, e in b)
    assert covered[0][0] == 0 and covered[-1][1] == fp.total and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    m.load_state_dict(before)
    assert fp.owns(m)                                                  # load_state_dict copies in place


def test_samplers_partition_rule():
    from asr_amd.data import BucketingSampler, DistributedBucketingSampler
    data = list(range(23))
    s = BucketingSampler(data, batch_size=4)
    assert len(s) == 6 and sorted(i for b in s for i in b) == data
    parts = [DistributedBucketingSampler(data, batch_size=4, num_replicas=4, rank=r) for r in range(4)]
    seen = [tuple(b) for p in parts for b in p]
    assert all(len(p) == 2 for p in parts) and len(seen) == 8           # 6 bins wrap-padded to 8
    assert {i for b in seen for i in b} == set(data)
    for p in parts:
        p.shuffle(epoch=3)
    assert [tuple(b) for b in parts[0].bins] == [tuple(b) for b in parts[1].bins]   # same permutation on every rank


def test_greedy_decoder_host_utilities_golden():
    """process_string / convert_to_strings (host utilities, used for TARGET strings) against the reference's decode
    goldens: feed them the first-maximum arg-max path and expect the reference's strings + offsets."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import det as mgi
    from asr_amd.decoders import GreedyDecoder
    from asr_amd._lib import DS2LibraryError
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "decode.json")))
    for name, g in gold.items():
        probs = mgi.decode_probs(name, g["B"], g["T"], len(g["labels"]), g["levels"])
        path = torch.from_numpy(np.argmax(probs, axis=2))
        d = GreedyDecoder(g["labels"])
        sizes = g["sizes"]
        strings, offsets = d.convert_to_strings(path, sizes, remove_repetitions=True, return_offsets=True)
        assert [s[0] for s in strings] == g["strings"], name
        assert [o[0].tolist() for o in offsets] == g["offsets"], name
    d = GreedyDecoder({c: i for i, c in enumerate("_abc ")})
    assert d.process_string(torch.tensor([1, 1, 0, 2, 4, 3]), 6, remove_repetitions=True)[0] == "ab c"
    assert d.cer("abc", "abd") == 1 and d.wer("a b c", "a x c") == 1
    if not torch.cuda.is_available():
        with pytest.raises(DS2LibraryError):           # decode() itself is GPU-only: no CPU path
            d.decode(torch.zeros(1, 3, 5))


DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from asr_amd.parallel import BucketedAllReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 1000
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
buckets = [("fc", 900, 1000), ("rnns.1", 500, 900), ("rnns.0", 100, 500), ("conv", 0, 100)]
red = BucketedAllReducer(g, buckets)
for name, _, _ in buckets:          # backward order
    red.on_bucket(name)
order = list(red.launched)
red.finish()
expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(g, expect), (g[:5], expect[:5])
assert order == ["fc", "rnns.1", "rnns.0", "conv"]
assert red.all_valid(True, "cpu") is True
assert red.all_valid(rank != 1, "cpu") is False       # one rank invalid -> everyone skips
dist.destroy_process_group()
print("OK", rank)
'''


def test_bucketed_allreduce_gloo_world2():
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "w.py")
        open(script, "w").write(DP_WORKER)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
            procs.append(subprocess.Popen([sys.executable, script, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=180)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert all("OK" in o for o in outs), outs


def test_trainer_api_surface():
    from asr_amd.trainers import DeepSpeechTrainer, Epochs, asr_metrics
    for name in ("run", "checkpoint", "description", "train", "fit", "step", "test", "update", "optimizer_to", "load", "save"):
        assert callable(getattr(DeepSpeechTrainer, name))
    e = Epochs(3)
    assert list(e) == [0, 1, 2] and e.total == 3
    m = asr_metrics()
    assert m.train.current.loss == 0.0 and m.test.best.cer is None
