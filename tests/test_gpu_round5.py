"""-m gpu, round 5: the data formats driven end to end on the GPU, the C-ABI's threading contract (§8(b))."""
import os
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from test_gpu_model import audio_conf, make_model

pytestmark = pytest.mark.gpu


def _write_corpus(tmp_path, n, rng, labels):
    """n pre-computed (161, T) spectrograms of 0.6 .. 2.4 s + a manifest and labels file in the reference's layout."""
    from asr_amd.data import write_manifest
    import pandas as pd
    rows, durs = [], []
    for i in range(n):
        T = int(rng.integers(60, 240))
        np.save(str(tmp_path / f"u{i:03d}.npy"), rng.standard_normal((161, T)).astype(np.float32))
        text = "".join(rng.choice(list("abcd"), size=int(rng.integers(2, 6))))
        if i % 7 == 0:
            text = text[:1] + " ?" + text[1:]                      # a space (no label: the space-row quirk) and an unknown character
        rows.append((str(tmp_path / f"u{i:03d}.npy"), T * 0.01, 16000, text))
        durs.append(T)
    write_manifest(rows, str(tmp_path / "manifest.csv"))
    pd.DataFrame({"label": labels}).to_csv(tmp_path / "labels.csv", index=False)
    return durs


def test_manifest_to_loader_to_step_for_an_epoch(tmp_path):
    """manifest.csv + labels.csv (the ETL's layout, written by asr_amd.data.write_manifest and held to the reference's bytes by
    test_data_formats_match_reference_golden) -> get_loader(length_bucketing=True) -> trainer.step for a whole epoch on the HIP path:
    every utterance is consumed exactly once, batches are homogeneous in length, the loss is finite and the weights move."""
    from asr_amd import CTCLoss
    from asr_amd.data import get_loader, LengthBucketingSampler
    from asr_amd.optim import FusedAdamW
    from asr_amd.trainers import DeepSpeechTrainer
    rng = np.random.default_rng(5)
    labels = ["_", "a", "b", " ", "c", "d"]                        # pandas drops the space row: 5 classes
    durs = _write_corpus(tmp_path, 37, rng, labels)
    conf = audio_conf()
    loader, sampler = get_loader(conf, str(tmp_path / "labels.csv"), str(tmp_path / "manifest.csv"), batch_size=8, num_workers=0,
                                 length_bucketing=True)
    assert isinstance(sampler, LengthBucketingSampler) and len(sampler) == 5
    assert loader.dataset.labels_map == {"_": 0, "a": 1, "b": 2, "c": 3, "d": 4}
    assert sorted(i for b in sampler.bins for i in b) == list(range(37))          # a partition of the manifest
    assert max(sampler.bin_spread()) <= (max(durs) - min(durs)) * 0.01 / 3        # bins homogeneous in length (durations in seconds)
    torch.manual_seed(0)
    model = make_model(dict(rnn="gru", hidden=64, layers=2, classes=5))
    w0 = dict(model.named_parameters())['fc.0.module.1.weight'].detach().clone()
    opt = FusedAdamW(model, lr=1e-3)
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, "cuda", "cuda", False, None)
    seen, losses = 0, []
    for batch in loader:
        inputs, targets, pct, tsz = batch
        assert inputs.size(2) == 161 and int(tsz.sum()) == targets.numel() and int(targets.min()) >= 1
        valid, loss = tr.step((inputs, targets, pct, tsz))
        assert valid
        losses.append(float(loss))
        seen += inputs.size(0)
    tr.synchronize()
    assert seen == 37 and len(losses) == 5 and all(np.isfinite(losses))
    assert not torch.equal(w0, dict(model.named_parameters())['fc.0.module.1.weight'].detach())


def test_distributed_length_bucketing_assignment_through_the_loader(tmp_path):
    """The per-rank bin assignment of the distributed sampler behind AudioDataLoader (no process group needed: ranks are constructed
    explicitly): over the ranks of a world every utterance of the manifest is delivered once per epoch, concurrent ranks hold neighbouring
    lengths, and a rank's batches run through trainer.step."""
    from asr_amd import CTCLoss
    from asr_amd.data import SpectrogramDataset, AudioDataLoader, DistributedLengthBucketingSampler
    from asr_amd.optim import FusedAdamW
    from asr_amd.trainers import DeepSpeechTrainer
    rng = np.random.default_rng(6)
    durs = np.asarray(_write_corpus(tmp_path, 32, rng, ["_", "a", "b", "c", "d"]))
    ds = SpectrogramDataset(audio_conf(), str(tmp_path / "manifest.csv"), str(tmp_path / "labels.csv"), normalize=True)
    world = 2
    samplers = [DistributedLengthBucketingSampler(ds, batch_size=4, num_replicas=world, rank=r) for r in range(world)]
    for s in samplers:
        s.shuffle(3)
    per_rank = [list(s) for s in samplers]
    assert sorted(i for r in per_rank for b in r for i in b) == list(range(32))
    for b0, b1 in zip(*per_rank):                                   # the two ranks of a round: adjacent bins of the length-sorted order
        assert abs(durs[b0].mean() - durs[b1].mean()) <= (durs.max() - durs.min()) / 3
    model = make_model(dict(rnn="gru", hidden=32, layers=2, classes=5))
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, FusedAdamW(model, lr=1e-3), None, None, "cuda", "cuda", False, None)
    loader = AudioDataLoader(ds, num_workers=0, batch_sampler=samplers[1])
    n = 0
    for batch in loader:
        valid, loss = tr.step(batch)
        assert valid and np.isfinite(float(loss))
        n += batch[0].size(0)
    assert n == 16


SIDE_ORDER_WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from test_gpu_model import make_model
from asr_amd import CTCLoss, engine
from asr_amd.optim import FusedAdamW
from asr_amd.trainers import DeepSpeechTrainer
torch.manual_seed(0)
B, T_in, H, L, C = 32, 161, 1280, 3, 29
model = make_model(dict(rnn="lstm", hidden=H, layers=L, classes=C))
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, FusedAdamW(model, lr=1e-4), None, None, "cuda", "cuda", True, None)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, 161, T_in, generator=g)
tsz = torch.full((B,), 6, dtype=torch.int32)
targets = torch.randint(1, C, (int(tsz.sum()),), generator=g, dtype=torch.int32)
pct = torch.ones(B)
snaps = []
real_all_reduce = dist.all_reduce
def spy(view, *a, **k):
    if view.numel() > 1000000:
        snaps.append((view, view.clone()))            # on the stream the collective is issued on, behind its "gradients final" event
    return real_all_reduce(view, *a, **k)
dist.all_reduce = spy
side = engine._side_stream(torch.device("cuda:0"))
out = {"steps": []}
for step in range(3):
    if step >= 1:
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(os.environ.get("SIDE_SLEEP", "200000000")))      # the side stream is LATE: its weight gradients lag the compute stream
    valid, loss = tr.step((x, targets, pct.clone(), tsz))
    tr.synchronize(); torch.cuda.synchronize()
    view, snap = snaps[-1]
    out["steps"].append({"loss": float(loss), "valid": bool(valid), "snapshot_equals_final": bool(torch.equal(view, snap)),
                         "max_abs_diff": float((view - snap).abs().max()), "side_used": True})
out["idle_schedule"] = bool(engine._BWD_PERSISTENT)
out["starved"] = DeepSpeechTrainer.starved_steps
print("SIDE_JSON " + json.dumps(out))
dist.destroy_process_group()
'''


def test_conv_schedule_releases_the_big_bucket_behind_late_side_stream_weight_gradients(tmp_path):
    """under the default data-parallel schedule ("conv": fc + recurrent buckets held and released as ONE collective at
    rnns.0, from the compute stream) the idle-CU weight-gradient schedule leaves the gradients of layers >= 1 on the SIDE stream; the
    release must be ordered behind them.  One rank, forced all-reduce, LSTM 3 x 1280 at B = 32 (160 of 256 CUs in the recurrence: the idle-CU
    schedule is on), the side stream artificially late: a snapshot of the gradient span taken where the collective is issued must equal
    the final gradients bit for bit."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = str(tmp_path / "side_order.py")
    open(script, "w").write(SIDE_ORDER_WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", DS2_FORCE_ALLREDUCE="1", DS2_DP_MODE="conv",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, script, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("SIDE_JSON ")][-1][len("SIDE_JSON "):])
    assert res["starved"] == 0
    for s in res["steps"]:
        assert s["valid"] and np.isfinite(s["loss"])
        assert s["snapshot_equals_final"], res


def test_fp32_step_with_hidden_size_not_a_multiple_of_8():
    """hidden sizes with H % 8 == 4 are accepted by the recurrence entry points; with T * B >= 512 the fp32 mode's
    split-bf16 GEMM path used to be taken in forward and then rejected by the grouped weight-gradient launch in backward.  They keep the
    fp32-MFMA GEMMs now: the step runs and meets the fp64 oracle."""
    from helpers import model_inputs, rel_l2
    from oracle import ds2_oracle as O
    from asr_amd import CTCLoss
    import det
    B, tmax = 8, 170                                               # T = 85 frames x B = 8 = 680 rows >= 512
    t_ins = sorted([int(v) for v in det.randint((B,), 61, tmax // 3, tmax + 1)], reverse=True)
    t_ins[0] = tmax
    cfg = dict(rnn="gru", hidden=100, layers=2, classes=29, t_ins=t_ins)
    sd, x, targets, pct, tsz = model_inputs(cfg)
    ref = O.fit_and_grads(sd, x, targets, pct, tsz, dtype=torch.float64)
    model = make_model(cfg, sd)
    out, out_lens = model.forward(x.cuda(), O.lengths_from_percentages(pct, x.size(3)))
    loss = CTCLoss(reduction="sum")(out.transpose(0, 1), targets, out_lens, tsz) / x.size(0)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_l2(out.detach().cpu().numpy(), ref["logits"].numpy()) < 1e-3
    assert abs(float(loss.detach()) - ref["loss"]) / ref["loss"] < 1e-3
    worst = max(rel_l2(p.grad.cpu().numpy(), ref["grads"][k].numpy()) for k, p in model.named_parameters())
    assert worst < 1e-3, worst


def _three_steps(cfg, stream, barrier=None):
    """three fused train steps of a c1-sized model on `stream`; returns (losses, flat weights, kernel-path bits, starved launches of this context)"""
    from helpers import model_inputs
    from asr_amd import CTCLoss, ops
    from asr_amd.optim import FusedAdamW
    from asr_amd.trainers import DeepSpeechTrainer
    torch.cuda.set_device(0)
    sd, x, targets, pct, tsz = model_inputs(cfg)
    with torch.cuda.stream(stream):
        model = make_model(cfg, sd)
        model.precision = cfg.get("precision", "fp32")
        tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, FusedAdamW(model, lr=1e-3), None, None, "cuda", "cuda", False, None)
        losses, paths = [], []
        for _ in range(3):
            if barrier is not None:
                barrier.wait(timeout=120)                       # both threads enqueue their step at the same moment
            valid, loss = tr.step((x, targets, pct.clone(), tsz))
            assert valid
            losses.append(float(loss))
            paths.append(ops.rnn_last_path())
        tr.synchronize()
        stream.synchronize()
        flat, _ = model.flat_parameters()
        return losses, flat.detach().cpu().clone(), paths, ops.rnn_persistent_counters()[0], ops.rnn_ctx_key()


def test_two_threads_on_two_streams_are_independent_and_bit_identical_to_serial_runs():
    """SURVEY §8(b) "Threading": the library keeps no mutable global state — what the recurrence entry points remember lives in a
    caller-owned ds2_rnn_ctx, one per (thread, device) in this binding.  Two Python threads drive c1-sized GRU and LSTM train steps (BASELINE
    configs[0]'s 2 x 256 width, bf16 and fp32) on two streams AT THE SAME TIME; each must end bit-identical to its own serial run, each sees
    its own kernel-path bits (GRU bf16: K-split backward = bits 1|2|4; LSTM fp32: another family), and the two threads hold different contexts."""
    t_ins = [201, 190, 171, 160]
    gru = dict(rnn="gru", hidden=256, layers=2, classes=29, t_ins=t_ins, precision="bf16")
    lstm = dict(rnn="lstm", hidden=256, layers=2, classes=29, t_ins=t_ins, precision="fp32")
    serial = {k: _three_steps(c, torch.cuda.Stream()) for k, c in (("gru", gru), ("lstm", lstm))}
    barrier = threading.Barrier(2)
    res, err = {}, []

    def work(name, cfg):
        try:
            res[name] = _three_steps(cfg, torch.cuda.Stream(), barrier)
        except BaseException as e:               # noqa: BLE001 - reported by the main thread
            err.append((name, repr(e)))
            barrier.abort()
    th = [threading.Thread(target=work, args=("gru", gru)), threading.Thread(target=work, args=("lstm", lstm))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not err, err
    for name in ("gru", "lstm"):
        sl, sw, sp, ss, sk = serial[name]
        cl, cw, cp, cs, ck = res[name]
        assert cl == sl, (name, cl, sl)                          # losses bit for bit
        assert torch.equal(cw, sw), name                         # every weight after three AdamW steps
        assert cp == sp, (name, cp, sp)                          # the same kernel families as in the serial run
        assert cs == 0 and ss == 0
    assert res["gru"][4] != res["lstm"][4] and res["gru"][4] != serial["gru"][4]   # three different contexts (main thread + two workers)
    assert res["gru"][2] != res["lstm"][2]                       # each thread read ITS OWN path bits, not the other's
