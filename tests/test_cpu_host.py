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


def test_shipped_library_has_no_work_skipping_switch():
    """ds2_debug_flags keeps its kernel-family SELECTORS (8 / 16 / 64 / 128: every selection computes the full result), but the ablation bits
    that skip the recurrent product / the gate epilogue (1 / 2, scripts/ablate_rnn.py) exist only in a -DDS2_ABLATE build: the shipped
    library masks them off at the ABI and compiles their tests out of the kernels."""
    from asr_amd import _lib
    lib = _lib.load()
    assert lib.ds2_ablation_build() == 0, "asr_amd/lib/libds2hip.so was built with ABLATE=1: not a shippable library"
    from asr_amd import ops
    ctx = ops.RnnCtx()                           # the selectors live in the caller's context (no device needed for this entry point)
    at = ctypes.addressof(ctx)
    assert lib.ds2_debug_flags(at, 3) == 0           # previous value
    assert lib.ds2_debug_flags(at, 1 | 2 | 64) == 0  # ... the request for 3 was dropped entirely
    assert lib.ds2_debug_flags(at, 0) == 64          # ... and only the selector survived of 67
    assert ctx.debug_flags == 0
    src = open(os.path.join(ROOT, "asr_amd", "csrc", "rnn.hip")).read()
    assert "#define DS2_ABLATE_BIT(flags, bit) false" in src and not re.search(r"dbg\s*&\s*[12]\b", src)


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
    # nothing under asr_amd/ (sub-packages and native sources included) may import, include or dlopen the oracle
    offenders = []
    for d, _, files in os.walk(os.path.join(ROOT, "asr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"^\s*(import\s+oracle|from\s+oracle)\b", src, re.M) or re.search(r"#include\s+[\"<][^\">]*oracle", src) \
                        or re.search(r"(dlopen|CDLL)\([^)]*oracle", src):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders


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
mode = os.environ.get("DS2_DP_MODE", "conv")
assert red.mode == mode
# "conv": fc + recurrent buckets are held and go out as ONE collective over their contiguous span once the last of them is final
assert order == (["fc+rnns.1+rnns.0", "conv"] if mode == "conv" else ["fc", "rnns.1", "rnns.0", "conv"]), order
assert red.all_valid(True, "cpu") is True
assert red.all_valid(rank != 1, "cpu") is False       # one rank invalid -> everyone skips
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("mode", ["conv", "serial", "overlap"])
def test_bucketed_allreduce_gloo_world2(mode):
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "w.py")
        open(script, "w").write(DP_WORKER)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + ["conv", "serial", "overlap"].index(mode)),
                       DS2_DP_MODE=mode)
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


def test_checkpoint_wire_format_and_finetune(tmp_path):
    """the `.pth` layout of DeepSpeechTrainer.save/load (deepspeech_trainer.py:154-188: keys epoch / metrics /
    optimizer / scheduler / state_dict, torch AdamW + StepLR objects as wired by trainers/__main__.py:41-52) and
    DeepSpeech.finetune_from (deepspeech.py:112-128).  Pure host logic: no kernels run."""
    from asr_amd.trainers import DeepSpeechTrainer
    torch.manual_seed(3)
    m = make("gru", 16, 2, 7)
    opt = torch.optim.AdamW(m.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.99)
    path = str(tmp_path / "ckpt" / "model.pth")
    tr = DeepSpeechTrainer(m, None, 5, None, opt, path, None, "cpu", "cpu", False, None, scheduler=sched)
    tr._epochs.best = 2
    tr._metrics.test.best.cer = 12.5
    tr.save()
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "metrics", "optimizer", "scheduler", "state_dict"} and raw["epoch"] == 2
    man = json.load(open(f"{GOLDEN}/state_manifest.json"))["gru_32x2_c7"]["keys"]
    assert list(raw["state_dict"].keys()) == list(man.keys())                     # the reference's key order
    assert set(raw["optimizer"]) == {"state", "param_groups"}
    # restore into a fresh model / optimizer
    torch.manual_seed(4)
    m2 = make("gru", 16, 2, 7)
    assert not torch.equal(m2.state_dict()["fc.0.module.1.weight"], m.state_dict()["fc.0.module.1.weight"])
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1.0)
    assert m.precision == "fp32"
    tr2 = DeepSpeechTrainer(m2, None, 5, None, opt2, path, None, "cpu", "cpu", True, None, overwrite_lr=7e-5)
    assert m2.precision == "bf16"                       # mixed_precision=True selects the bf16-operand mode (no GradScaler needed)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert tr2._epochs.start == tr2._epochs.current == 2 and tr2._metrics.test.best.cer == 12.5
    assert opt2.param_groups[0]["lr"] == 7e-5 and opt2.param_groups[0]["betas"] == (0.9, 0.999)
    assert tr2._scheduler is not None and tr2._scheduler.optimizer is opt2
    # finetune_from: shape-checked partial load (a 9-class head does not take the 7-class fc), all but the last tensor frozen
    sd_path = str(tmp_path / "sd.pth")
    torch.save(m.state_dict(), sd_path)
    m3 = make("gru", 16, 2, 9)
    before_fc = m3.state_dict()["fc.0.module.1.weight"].clone()
    m3.finetune_from(sd_path, nlayers=1)
    assert torch.equal(m3.state_dict()["conv.seq_module.0.weight"], m.state_dict()["conv.seq_module.0.weight"])
    assert torch.equal(m3.state_dict()["fc.0.module.1.weight"], before_fc)
    flags = [p.requires_grad for p in m3.parameters()]
    assert flags[-1] and not any(flags[:-1])


def test_manifest_label_formats_and_loader(tmp_path):
    """the on-disk formats the reference's ETL writes — manifest CSV `audio_filepath,duration,fq,text,text_size`
    (etl/jsut_dataset.py:36-42) and labels.csv with one `label` column (jsut_dataset.py:56-60) — through SpectrogramDataset /
    get_loader / _collate_fn.  Host logic only; the wav goes through the numpy STFT restatement."""
    import pandas as pd
    from scipy.io import wavfile
    from asr_amd.data import SpectrogramDataset, get_loader
    from oracle import stft_oracle as S
    sr = 16000
    waves = {}
    for i, n in enumerate((8000, 12000, 4000)):
        y = (0.3 * np.sin(2 * np.pi * (300 + 100 * i) * np.arange(n) / sr)).astype(np.float32)
        waves[i] = (np.round(y * 32767)).astype(np.int16)
        wavfile.write(str(tmp_path / f"u{i}.wav"), sr, waves[i])
    texts = ["ab c", "cab", "b?a"]                                             # '?' is not a label: dropped (spectrogram_dataset.py:70-73)
    pd.DataFrame({"audio_filepath": [str(tmp_path / f"u{i}.wav") for i in range(3)], "duration": [0.5, 0.75, 0.25], "fq": [sr] * 3,
                  "text": texts, "text_size": [len(t) for t in texts]}).to_csv(tmp_path / "manifest.csv", index=False)
    pd.DataFrame({"label": ["_", "a", "b", "c", " "]}).to_csv(tmp_path / "labels.csv", index=False)
    conf = audio_conf()
    ds = SpectrogramDataset(audio_conf=conf, manifest_filepath=str(tmp_path / "manifest.csv"), labels=str(tmp_path / "labels.csv"), normalize=False)
    assert len(ds) == 3
    spect, ids = ds[0]
    assert spect.dtype == torch.float32 and spect.shape == (161, 1 + 8000 // 160) and bool(torch.isfinite(spect).all())
    ref = S.stft_log_spectrogram(waves[0].astype(np.float32) / 32767.0, 320, 160, "hamming", "constant")
    assert np.allclose(spect.numpy(), ref, atol=2e-5)
    # space-row quirk PRESERVED: pandas.read_csv skips the whitespace-only row of labels.csv, so ' ' never
    # becomes a label (and rows after it would shift down by one) — same call as spectrogram_dataset.py:36 / deepspeech.py:48
    assert " " not in ds.labels_map and ds.labels_map == {"_": 0, "a": 1, "b": 2, "c": 3}
    assert ids == [1, 2, 3] and ds[2][1] == [2, 1]                               # "ab c" -> a b c ; "b?a" -> b a ; index-0 '_' is never emitted
    loader, sampler = get_loader(conf, str(tmp_path / "labels.csv"), str(tmp_path / "manifest.csv"), batch_size=2, num_workers=0)
    batches = list(loader)
    assert len(batches) == 2 and len(sampler) == 2
    for inputs, targets, pct, tsz in batches:
        assert inputs.dim() == 4 and inputs.size(1) == 1 and inputs.size(2) == 161
        assert float(pct.max()) == 1.0 and int(tsz.sum()) == targets.numel()


def test_data_formats_match_reference_golden(tmp_path):
    """SURVEY §8(f) rank 4, pinned: labels.csv / manifest.csv semantics, the two reference samplers and the ETL's on-disk layout against
    tests/golden/data_formats.json, which make_golden.py produces by running the reference's OWN SpectrogramDataset, BucketingSampler,
    DistributedBucketingSampler and JSUTDataset (data/dataset/spectrogram_dataset.py:36,70-73, data/samplers/*.py, etl/jsut_dataset.py:28-60)."""
    import json
    from asr_amd.data import (SpectrogramDataset, BucketingSampler, DistributedBucketingSampler, write_manifest, export_labels,
                              clean_jsut_text)
    import pandas as pd
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "data_formats.json"), encoding="utf-8"))
    # ---- the reference-written label and manifest files through this repo's dataset
    (tmp_path / "labels.csv").write_text(g["labels_csv_text"], encoding="utf-8")
    (tmp_path / "manifest.csv").write_text(g["manifest_csv_text"], encoding="utf-8")
    ds = SpectrogramDataset(audio_conf=audio_conf(), manifest_filepath=str(tmp_path / "manifest.csv"), labels=str(tmp_path / "labels.csv"))
    assert ds.labels_map == g["labels_map"] and " " not in ds.labels_map          # the space-row quirk, index shift included
    assert len(ds) == g["dataset_len"]
    assert [bool(x) for x in ds.df["text"].isnull().tolist()] == g["manifest_text_isnull"]
    for text, ids in g["parse_transcript"]:
        assert ds.parse_transcript(text) == ids, text

    # ---- samplers
    class N:
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
    for c in g["bucketing_sampler"]:
        smp = BucketingSampler(N(c["n"]), batch_size=c["batch_size"])
        assert smp.bins == c["bins"] and len(smp) == c["len"]
        np.random.seed(1234)
        assert [list(x) for x in smp] == c["iter_seed1234"]
        np.random.seed(99)
        smp.shuffle()
        assert smp.bins == c["bins_after_iter_then_shuffle_seed99"]
    for c in g["distributed_bucketing_sampler"]:
        if c.get("raises"):
            with pytest.raises(AssertionError):
                list(DistributedBucketingSampler(N(c["n"]), batch_size=c["batch_size"], num_replicas=c["world"], rank=0))
            continue
        for r in range(c["world"]):
            smp = DistributedBucketingSampler(N(c["n"]), batch_size=c["batch_size"], num_replicas=c["world"], rank=r)
            assert [list(x) for x in smp] == c["per_rank"][r] and len(smp) == c["len"][r]
            assert (smp.num_samples, smp.total_size, smp.num_replicas, smp.rank) == (c["len"][r], c["len"][r] * c["world"], c["world"], r)
            for ep, v in c["shuffle"].items():
                smp = DistributedBucketingSampler(N(c["n"]), batch_size=c["batch_size"], num_replicas=c["world"], rank=r)
                smp.shuffle(int(ep))
                assert smp.bins == v["bins"] and [list(x) for x in smp] == v["per_rank"][r]
        # every id exactly once per epoch over the ranks when nothing had to be padded
        if c["n"] % (c["batch_size"] * c["world"]) == 0:
            seen = sorted(i for r in range(c["world"]) for b in c["per_rank"][r] for i in b)
            assert seen == list(range(c["n"]))

    # ---- ETL layout: the same rows through this repo's writers give the reference's files byte for byte
    e = g["etl"]
    cleaned = dict(clean_jsut_text(l) for l in e["transcript_lines"])
    assert [[l, list(clean_jsut_text(l))] for l in e["transcript_lines"]] == e["clean_text"]
    rows = [(f"/bronze/basic5000/wav/{k}.wav", d, 16000, cleaned[k]) for k, d in e["durations"].items() if 1 <= d <= 5]
    write_manifest(rows, str(tmp_path / "etl_manifest.csv"))
    assert (tmp_path / "etl_manifest.csv").read_text(encoding="utf-8") == e["manifest_csv_text"]
    assert list(pd.read_csv(tmp_path / "etl_manifest.csv").columns) == e["manifest_columns"]
    export_labels(cleaned.values(), str(tmp_path / "etl_labels.csv"))
    lab = (tmp_path / "etl_labels.csv").read_text(encoding="utf-8").splitlines()
    assert lab[0] == e["labels_csv_header"] and sorted(pd.read_csv(tmp_path / "etl_labels.csv")["label"].tolist()) == e["labels_set"]


def test_length_bucketing_samplers_partition_and_order():
    """bins are homogeneous in length, cover every item exactly once, keep the reference's bin/shuffle interface;
    the distributed variant gives concurrent ranks neighbouring lengths (reference partition rule on length-sorted bins)."""
    from asr_amd.data import BucketingSampler, DistributedLengthBucketingSampler, LengthBucketingSampler
    rng = np.random.RandomState(0)
    n, bs, world = 1003, 64, 8
    dur = rng.uniform(3.0, 20.0, n)
    ds = list(range(n))
    s = LengthBucketingSampler(ds, bs, durations=dur)
    assert len(s) == -(-n // bs) == len(BucketingSampler(ds, bs))
    assert sorted(i for b in s for i in b) == ds                                   # a partition of the data set
    edges = [(dur[b].min(), dur[b].max()) for b in s.bins]
    assert all(edges[k][1] <= edges[k + 1][0] for k in range(len(edges) - 1))       # bins are consecutive runs of the sorted order
    spread_sorted = max(s.bin_spread())
    spread_manifest = max(float(dur[b].max() - dur[b].min()) for b in BucketingSampler(ds, bs).bins)
    assert spread_sorted < 0.15 * spread_manifest                                  # what bucketing buys: ~1.4 s vs ~17 s here
    before = [list(b) for b in s.bins]
    s.shuffle(epoch=3)
    assert sorted(map(sorted, s.bins)) == sorted(map(sorted, before)) and [sorted(b) for b in s.bins] != [sorted(b) for b in before]
    t = LengthBucketingSampler(ds, bs, durations=dur)
    t.shuffle(epoch=3)
    assert [sorted(b) for b in t.bins] == [sorted(b) for b in s.bins]               # epoch-seeded: every process agrees
    # distributed: same number of steps on every rank, every item covered, ranks of one step are neighbours in length
    ranks = [DistributedLengthBucketingSampler(ds, bs, world, r, durations=dur) for r in range(world)]
    for r in ranks:
        r.shuffle(7)
    its = [list(r) for r in ranks]
    steps = len(ranks[0])
    assert all(len(it) == steps for it in its) and steps == -(-len(s) // world)
    assert set(i for it in its for b in it for i in b) == set(ds)
    for k in range(steps):
        longest = [dur[it[k]].max() for it in its]
        assert max(longest) - min(longest) <= (world + 1) * spread_sorted + 1e-9     # within a round: adjacent bins of the sorted order
    with pytest.raises(ValueError):
        LengthBucketingSampler(ds, bs)                                              # no durations and no manifest to take them from


def test_fused_adamw_state_handling():
    """restored moments are moved, not re-zeroed; foreign state dicts are rejected; frozen parameters lie outside the
    launched spans; zero_grad clears the autograd views."""
    from asr_amd import FusedAdamW
    from asr_amd.params import FlatParams
    m = make("gru", 32, 2, 7)
    m._flat = FlatParams(m, 2, "cpu")
    opt = FusedAdamW(m)
    assert opt._trainable_spans() == [(0, m._flat.total)]
    for p in list(m.parameters())[:-1]:
        p.requires_grad = False                                                     # finetune_from(nlayers=1) (deepspeech.py:124-128)
    spans = opt._trainable_spans()
    o, sz = m._flat.offsets["fc.0.module.1.weight"]
    assert spans == [(o, o + (sz + 3) // 4 * 4)]
    for p in m.parameters():
        p.requires_grad = True
    assert opt._trainable_spans() == [(0, m._flat.total)]
    # restored state keeps its values
    opt.state = {"step": 5, "exp_avg": torch.full((m._flat.total,), 0.25, dtype=torch.float64), "exp_avg_sq": torch.ones(m._flat.total, dtype=torch.float64)}
    opt._ensure_state()
    assert opt.state["exp_avg"].dtype == torch.float32 and float(opt.state["exp_avg"][0]) == 0.25 and float(opt.state["exp_avg_sq"][-1]) == 1.0
    sd = opt.state_dict()
    opt2 = FusedAdamW(m)
    opt2.load_state_dict(sd)
    assert opt2.state["step"] == 5 and torch.equal(opt2.state["exp_avg"], opt.state["exp_avg"])
    ref_sd = torch.optim.AdamW(m.parameters(), lr=1e-3).state_dict()                # what the reference trainer checkpoints: converted
    opt2.load_state_dict(ref_sd)
    assert opt2.state["step"] == 0 and opt2.param_groups[0]["lr"] == 1e-3 and float(opt2.state["exp_avg"].abs().sum()) == 0.0
    ams = torch.optim.AdamW(m.parameters(), lr=1e-3, amsgrad=True).state_dict()
    with pytest.raises(ValueError):
        opt2.load_state_dict(ams)
    with pytest.raises(ValueError):
        opt2.load_state_dict({"state": {"step": 1, "exp_avg": torch.zeros(3), "exp_avg_sq": None}, "param_groups": opt.param_groups})
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    opt.zero_grad()
    assert all(p.grad is None for p in m.parameters())


def test_reference_written_checkpoint_resumes(tmp_path):
    """SURVEY §8(f) rank 3 against a file the REFERENCE's classes wrote (tests/golden/make_golden.py::gen_checkpoint: the reference's
    DeepSpeech + torch AdamW + StepLR after one real optimizer step, saved in the five-key layout of its trainer's save()): the trainer
    restores the model bit for bit, and BOTH optimizer kinds resume from it - torch AdamW natively, FusedAdamW by conversion of the
    per-parameter moments into its flat layout (same step count, same hyper-parameters)."""
    from asr_amd import FusedAdamW
    from asr_amd.trainers import DeepSpeechTrainer
    path = f"{GOLDEN}/ref_checkpoint_gru_16x2_c7.pth"
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "metrics", "optimizer", "scheduler", "state_dict"}
    for kind in ("torch", "fused"):
        torch.manual_seed(9)
        m = make("gru", 16, 2, 7)
        assert list(m.state_dict().keys()) == list(raw["state_dict"].keys())
        opt = torch.optim.AdamW(m.parameters(), lr=1.0) if kind == "torch" else FusedAdamW(m, lr=1.0)
        tr = DeepSpeechTrainer(m, None, 5, None, opt, path, None, "cpu", "cpu", False, None)
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), raw["state_dict"].values()))
        assert tr._epochs.start == 3 and tr._metrics["test"]["best"]["cer"] == 12.5
        assert isinstance(tr._scheduler, torch.optim.lr_scheduler.StepLR) and tr._scheduler.optimizer is opt
        g = opt.param_groups[0]
        assert g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8 and g["weight_decay"] == 1e-5
        assert abs(g["lr"] - 1.5e-4 * 0.99) < 1e-12                                # one StepLR step had been taken
        names = [n for n, _ in m.named_parameters()]
        if kind == "torch":
            st = opt.state_dict()["state"]
            assert int(st[0]["step"]) == 1 and torch.equal(st[3]["exp_avg"], raw["optimizer"]["state"][3]["exp_avg"])
        else:
            assert opt.state["step"] == 1
            for i, n in enumerate(names):
                o, sz = m._flat.offsets[n]
                assert torch.equal(opt.state["exp_avg"][o:o + sz], raw["optimizer"]["state"][i]["exp_avg"].reshape(-1)), n
                assert torch.equal(opt.state["exp_avg_sq"][o:o + sz], raw["optimizer"]["state"][i]["exp_avg_sq"].reshape(-1)), n


def test_gemm_isa_no_spills_and_no_copy_of_inflight_fragments():
    """Generated gfx950 code of the bf16 GEMM kernels (hipcc -S, no GPU): no register spills in the hot kernels, and in the TN kernel no
    instruction touches the destination of an inline-asm `ds_read_b64_tr_b16` before the wait that covers it (scripts/check_isa.py)."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 instruction(s) touching in-flight" in r.stdout


def test_committed_pmc_summary_belongs_to_this_tree_and_names_the_launched_instance():
    """bench.py only quotes `roofline.traffic` from a PMC summary whose recorded kernel-source hash equals the tree's and whose kernel is the
    template instance the metric config launches: the committed summary must satisfy both, or every bench line of this tree reports
    traffic null (it did once: a fourth template parameter changed the printed instance name)."""
    import bench
    pmc, why_not = bench.load_pmc_summary()
    if pmc is None:
        # the recurrence sources were edited after the last counter collection: bench.py then reports `traffic` null WITH this reason, which is
        # the designed behaviour — re-collect on a GPU box (scripts/gpu_pmc_persistent.sh); not a reason to fail the CPU suite
        pytest.skip(why_not)
    assert pmc["kernels"]["rnn_bwd_ksplit_kernel"]["kernel"] == bench.expected_ksplit_instance(3, 1024)
    assert pmc["kernels"]["rnn_bwd_ksplit_kernel"]["hbm_bytes_per_time_step"] > 0
    assert pmc["kernels"]["rnn_fwd_persistent_kernel"]["hbm_bytes_per_time_step"] > 0


def test_maskconv_generic_stack_keeps_the_reference_container_semantics():
    """MaskConv is also a generic container in the reference (blocks.py:42-56).  Any stack other than DeepSpeech's own runs
    module by module with everything beyond each length zeroed — on CPU tensors and under autograd too; DeepSpeech's stack stays HIP-only."""
    import torch.nn as nn
    from asr_amd.modules.blocks import MaskConv
    torch.manual_seed(0)
    m = MaskConv(nn.Sequential(nn.Conv2d(1, 4, (3, 3), padding=1), nn.ReLU()))
    x = torch.randn(2, 1, 8, 10)
    out, lens = m(x, [10, 6])
    assert out.shape == (2, 4, 8, 10) and float(out[1, :, :, 6:].abs().sum()) == 0.0 and float(out[0].abs().sum()) > 0
    out.sum().backward()
    assert m.seq_module[0].weight.grad is not None


def test_fp32_split_gemm_decision_needs_an_aligned_hidden_size():
    from asr_amd import engine
    if engine.F32_GEMM != "split":
        pytest.skip("DS2_F32_GEMM overridden")
    assert engine._f32_split_ok(648, 2 * 3 * 96, 1312, 96)
    assert not engine._f32_split_ok(648, 2 * 3 * 100, 1312, 100)          # dW_hh problems would have N = 100, column offsets 100, 300


def test_library_exports_no_writable_global_state():
    """SURVEY §8(b) "Threading": no mutable global state besides the thread-local error message.  The dynamic symbol table of the shipped
    library holds functions, kernel descriptors and the HIP fat-binary bookkeeping — no data object of the library's own — and the
    recurrence state that used to be process-global (kernel-path bits, cooldown, enable switches, debug selectors, poison word) is a
    caller-owned ds2_rnn_ctx whose layout the header, the library and the ctypes mirror agree on."""
    from asr_amd import _lib, ops
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    data = [l.split() for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "BDbdCGgSs"]
    own = [n for _, _, n in data if not (n.startswith("__hip_") or n.startswith("_Z") or n.startswith("__") or n in ("_edata", "_end", "_fini", "_init"))]
    assert own == [], f"writable data symbols exported by libds2hip.so: {own}"
    src = open(os.path.join(ROOT, "asr_amd", "csrc", "rnn.hip")).read() + open(os.path.join(ROOT, "asr_amd", "csrc", "api.cpp")).read()
    assert not re.search(r"^\s*(static\s+)?(int|bool)\s*\*?\s*g_(last|persist|poison|ds2_debug)\w*\s*(=|;)", src, re.M)
    hdr = open(os.path.join(ROOT, "include", "ds2hip.h")).read()
    fields = re.search(r"typedef struct ds2_rnn_ctx \{(.*?)\} ds2_rnn_ctx;", hdr, re.S).group(1)
    n_int = sum(int(m.group(2) or 1) for m in re.finditer(r"\bint\s+(\w+)(?:\[(\d+)\])?\s*[;,]", fields)) + fields.count("persist_fwd, persist_bwd") * 1
    assert ctypes.sizeof(ops.RnnCtx) == 16 * 4 + 3 * 8 and [f[0] for f in ops.RnnCtx._fields_][-3:] == ["status_dev", "poison_host", "poison_dev"]
    ctx = ops.RnnCtx()
    dummy = (ctypes.c_int * 8)()
    assert _lib.load().ds2_rnn_ctx_init(ctypes.addressof(ctx), ctypes.addressof(dummy), None, None) == 0
    assert ctx.size == ctypes.sizeof(ops.RnnCtx) and (ctx.persist_fwd, ctx.persist_bwd, ctx.cooldown, ctx.last_path) == (1, 1, 0, 0)
    assert ctx.rearm_calls == int(os.environ.get("DS2_RNN_REARM_CALLS", "64"))


def test_unidirectional_variant_matches_reference_golden():
    """VERDICT round 5 item 8 / SURVEY §2 row 1: `bidirectional=False` + `Lookahead` (deepspeech.py:83-101, blocks.py:96-132) is not a kernel
    target; asr_amd serves it on torch ops behind the same class names.  One step of the reference's statement sequence on the imported
    reference (tests/golden/make_golden.py --only-uni) against the same step here: logits, loss, every gradient, eval-mode probabilities."""
    import json
    import det
    from helpers import rel_l2, subsample
    from asr_amd import DeepSpeech
    import pandas as pd
    z = np.load(os.path.join(GOLDEN, "model_uni_gru_h16_l2.npz"))
    cfg = json.loads(str(z["cfg"]))
    chars = ["_", "'"] + list("abcdefghijklmnopqrstuvwxyz")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "labels.csv")
        pd.DataFrame({"label": chars[:cfg["classes"]]}).to_csv(path, index=False)
        from types import SimpleNamespace
        ac = SimpleNamespace(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming")
        model = DeepSpeech(audio_conf=ac, decoder=None, label_path=path, rnn_type="nn.GRU", rnn_hidden_size=cfg["hidden"],
                           rnn_hidden_layers=cfg["layers"], bidirectional=False, context=cfg["context"])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert {k: list(v) for k, v in shapes.items()} == cfg["shapes"]                     # the reference's parameter set, key for key
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in det.model_state(shapes, base_seed=0).items()})
    model.train()
    x, targets, pct, tsz = det.batch(len(cfg["t_ins"]), cfg["t_ins"], cfg["classes"], seed=cfg["seed"])
    inputs = torch.from_numpy(x)
    sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int()
    out, out_lens = model.forward(inputs, sizes)
    assert np.array_equal(out_lens.numpy(), z["output_sizes"])
    loss = torch.nn.CTCLoss(reduction="sum")(out.transpose(0, 1).float().log_softmax(2), torch.from_numpy(targets), out_lens, torch.from_numpy(tsz)) / inputs.size(0)
    loss.backward()
    assert rel_l2(out.detach().numpy(), z["logits"]) < 2e-5 and abs(float(loss) - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    gmax = max(float(z["gradnorm_" + k]) for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        got, ref = subsample(p.grad.numpy()), z["grad_" + k]
        assert np.linalg.norm(got.astype(np.float64) - ref) <= 2e-4 * max(np.linalg.norm(ref), 1e-3 * float(z["gradnorm_" + k]), 1e-6 * gmax), k
    model.eval()
    with torch.no_grad():
        probs, _ = model.forward(inputs, torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int())
    assert rel_l2(probs.numpy(), z["eval_probs"]) < 2e-5
    # the fused MI355X step is for the bidirectional model: it says so instead of failing somewhere inside
    from asr_amd.trainers import DeepSpeechTrainer
    tr = DeepSpeechTrainer(model, torch.nn.CTCLoss(reduction="sum"), 1, None, torch.optim.AdamW(model.parameters()), None, None, "cpu", "cpu", False, None)
    with pytest.raises(NotImplementedError):
        tr.step((inputs, torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)))
