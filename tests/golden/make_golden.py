"""Generate golden vectors by running the UNMODIFIED reference (zakuro-ai/asr, /root/reference) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs (committed, data only — expected outputs, never reference source):
    tests/golden/ctc.npz            torch.nn.CTCLoss (the reference's criterion) known answers
    tests/golden/lengths.npz        get_seq_lens table, _collate_fn + fit() length recovery (A.5)
    tests/golden/collate.npz        _collate_fn on a 3-sample batch
    tests/golden/state_manifest.json  state_dict keys/shapes of the reference model (A.1)
    tests/golden/decode.json        GreedyDecoder.decode (argmax, collapse repeats, drop blanks) known answers
    tests/golden/data_formats.json  labels.csv / manifest.csv through the reference's SpectrogramDataset (labels_map, parse_transcript), the
                                    reference's BucketingSampler / DistributedBucketingSampler bins and shuffles, the JSUT ETL's manifest and
                                    labels layout (`--only-data-formats`)
    tests/golden/model_<name>.npz   whole model: logits, loss, grads (sub-sampled), BN running
                                    stats, 3 AdamW steps — reference statement sequence of
                                    DeepSpeechTrainer.fit + backward + AdamW.step
    tests/golden/model_c3_full.npz, model_c2_full.npz   BASELINE configs[2] / [1] at FULL size (5x1024 B=64, 5x768 B=32, T_in 1001, ragged): one step
                                    of the imported reference AND of the fp64 oracle — loss, per-utterance logit checksums, sub-sampled gradients
                                    (`--only-full c3_full`, tens of minutes each; not part of the default run)
    tests/golden/model_uni_gru_h16_l2.npz   the unidirectional variant (bidirectional=False + Lookahead), one step (`--only-uni`)
    tests/golden/ref_checkpoint_gru_16x2_c7.pth (+ _eval.npz)   a checkpoint in the reference trainer's wire format written from the
                                    reference's own model / torch AdamW / StepLR after one optimizer step, the reference model's eval
                                    probabilities and train-mode loss at those weights (`--only-checkpoint` regenerates it, byte-identical)
Inputs/weights are NOT stored: they are regenerated from integer hashes (tests/golden/det.py) — the checkpoint fixture excepted, whose
whole point is the file.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import det  # noqa: E402
from _ref_shim import import_reference  # noqa: E402

GRAD_STRIDE = 13
GRAD_FULL_MAX = 4096


def subsample(a: np.ndarray) -> np.ndarray:
    f = a.reshape(-1)
    return f if f.size <= GRAD_FULL_MAX else f[::GRAD_STRIDE]


def audio_conf():
    return SimpleNamespace(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming",
                           speed_volume_perturb=False, spec_augment=False, noise_dir=None,
                           noise_prob=0.4, noise_levels=(0.0, 0.5))


def label_csv(tmp, n):
    import pandas as pd
    chars = ["_", "'"] + list("abcdefghijklmnopqrstuvwxyz") + ["|"] + [chr(0x3041 + i) for i in range(100)]
    path = os.path.join(tmp, f"labels{n}.csv")
    pd.DataFrame({"label": chars[:n]}).to_csv(path, index=False)
    return path


MODELS = {
    # name: (rnn, hidden, layers, classes, t_ins)
    "gru_h32_l2": ("gru", 32, 2, 7, [40, 33, 21]),
    "lstm_h24_l2": ("lstm", 24, 2, 7, [40, 33, 21]),
    "gru_h48_l3": ("gru", 48, 3, 29, [90, 77, 64, 50, 31]),
    "lstm_h40_l3": ("lstm", 40, 3, 29, [61, 61, 47, 22]),
    # BASELINE.json configs[0] itself (C1: DS2-tiny 2x256 BiGRU, 29 labels, batch 4 of 2 s = 201 input frames)
    "c1_gru_h256_l2": ("gru", 256, 2, 29, [201, 201, 201, 201]),
}


def gen_ctc(out):
    torch.manual_seed(0)
    crit = torch.nn.CTCLoss(reduction="none")
    cases = {}
    # case A: all feasible, ragged, repeated labels
    T, B, C = 12, 4, 7
    logits = torch.from_numpy(det.unitvar((T, B, C), 11)).requires_grad_(True)
    targets = torch.tensor([1, 1, 2, 3, 3, 3, 4, 5, 6, 6, 1, 2], dtype=torch.int32)
    tl = torch.tensor([3, 4, 4, 1], dtype=torch.int32)
    il = torch.tensor([12, 10, 9, 5], dtype=torch.int32)
    lp = logits.log_softmax(2)
    nll = crit(lp, targets, il, tl)
    nll.sum().backward()
    cases["a"] = dict(seed=11, T=T, B=B, C=C, targets=targets.numpy(), tl=tl.numpy(), il=il.numpy(),
                      nll=nll.detach().numpy(), grad=logits.grad.numpy())
    # case B: row 1 infeasible (needs 2*3-? frames: "1 1 1" needs 5 frames, only 3 given)
    T, B, C = 9, 3, 5
    logits = torch.from_numpy(det.unitvar((T, B, C), 12)).requires_grad_(True)
    targets = torch.tensor([2, 3, 1, 1, 1, 4], dtype=torch.int32)
    tl = torch.tensor([2, 3, 1], dtype=torch.int32)
    il = torch.tensor([9, 3, 2], dtype=torch.int32)
    lp = logits.log_softmax(2)
    nll = crit(lp, targets, il, tl)
    cases["b"] = dict(seed=12, T=T, B=B, C=C, targets=targets.numpy(), tl=tl.numpy(), il=il.numpy(),
                      nll=nll.detach().numpy())
    # case C: longer, C=29, U up to 10, including length-0 target
    T, B, C = 50, 5, 29
    logits = torch.from_numpy(det.unitvar((T, B, C), 13) * 2.0).requires_grad_(True)
    tl = torch.tensor([10, 7, 5, 0, 3], dtype=torch.int32)
    targets = torch.from_numpy(det.randint((int(tl.sum()),), 14, 1, C).astype(np.int32))
    il = torch.tensor([50, 44, 30, 20, 7], dtype=torch.int32)
    lp = logits.log_softmax(2)
    nll = crit(lp, targets, il, tl)
    nll.sum().backward()
    cases["c"] = dict(seed=13, T=T, B=B, C=C, targets=targets.numpy(), tl=tl.numpy(), il=il.numpy(),
                      nll=nll.detach().numpy(), grad=logits.grad.numpy(), scale=2.0)
    flat = {}
    for k, d in cases.items():
        for kk, v in d.items():
            flat[f"{k}_{kk}"] = np.asarray(v)
    np.savez_compressed(os.path.join(out, "ctc.npz"), **flat)


def gen_lengths(out, DeepSpeech, functional, tmp):
    model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, 7),
                       rnn_type="nn.GRU", rnn_hidden_size=8, rnn_hidden_layers=1)
    L = torch.arange(1, 2002, dtype=torch.int32)
    seq = model.get_seq_lens(L).numpy().astype(np.int32)
    rec = {}
    for tmax in (201, 501, 1001, 1501, 2001):
        tb = np.arange(1, tmax + 1)
        pct = torch.zeros(tmax, dtype=torch.float32)
        for i, t in enumerate(tb):
            pct[i] = int(t) / float(tmax)          # functional.py:28
        rec[f"rec_{tmax}"] = pct.mul_(int(tmax)).int().numpy()  # deepspeech_trainer.py:104
    np.savez_compressed(os.path.join(out, "lengths.npz"), seq_lens=seq, **rec)

    # _collate_fn example (functional.py:9-32)
    specs = [torch.from_numpy(det.unitvar((161, t), 20 + i)) for i, t in enumerate((15, 20, 9))]
    tr = [[4, 5], [1, 2, 3], [6]]
    inputs, targets, pct, sizes = functional._collate_fn(list(zip(specs, tr)))
    np.savez_compressed(os.path.join(out, "collate.npz"), inputs=inputs.numpy(), targets=targets.numpy(),
                        pct=pct.numpy(), sizes=sizes.numpy())


def gen_model(out, name, DeepSpeech, tmp):
    rnn, hidden, layers, classes, t_ins = MODELS[name]
    model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, classes),
                       rnn_type={"gru": "nn.GRU", "lstm": "nn.LSTM"}[rnn], rnn_hidden_size=hidden,
                       rnn_hidden_layers=layers, bidirectional=True)
    ref_sd = model.state_dict()
    shapes = det.state_shapes(rnn, hidden, layers, classes)
    assert list(shapes.keys()) == list(ref_sd.keys()), "state_dict key order/name mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    weights = det.model_state(shapes, base_seed=0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    model.train()

    # data seed: first one whose BatchNorm2d outputs (reference modules, fp32) keep >= 4e-6 distance (10x fp32 round-off of z) to the
    # Hardtanh kinks at every un-masked position (ill-conditioned otherwise: oracle.hardtanh_kink_margin)
    seed = 1
    while True:
        x, targets, pct, tsz = det.batch(len(t_ins), t_ins, classes, seed=seed)
        inputs = torch.from_numpy(x)
        zs = []
        hk = [model.conv.seq_module[i].register_forward_hook(lambda m, i, o: zs.append(o.detach().clone())) for i in (1, 4)]
        with torch.no_grad():
            sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int()
            ol = model.get_seq_lens(sizes)
            model.conv(inputs, ol)
        for h in hk:
            h.remove()
        margin = 1e30
        for zt in zs:
            msk = (torch.arange(zt.size(3)).view(1, 1, 1, -1) < ol.view(-1, 1, 1, 1)).expand_as(zt)
            v = zt[msk].double()
            margin = min(margin, float(torch.minimum(v.abs(), (v - 20).abs()).min()))
        if margin >= 4e-6 or seed > 400:
            break
        seed += 2
    print(name, "data seed", seed, "kink margin", margin)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})  # undo BN running-stat updates
    targets = torch.from_numpy(targets)
    tsz = torch.from_numpy(tsz)
    criterion = torch.nn.CTCLoss(reduction="sum")
    opt = torch.optim.AdamW(model.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)

    taps = {}
    hooks = [model.conv.register_forward_hook(lambda m, i, o: taps.__setitem__("act2", o[0].detach().clone()))]
    for l, r in enumerate(model.rnns):
        hooks.append(r.register_forward_hook(lambda m, i, o, l=l: taps.__setitem__(f"rnn{l}", o.detach().clone())))

    rec = {}
    losses = []
    for step in range(3):
        input_percentages = torch.from_numpy(pct.copy())
        # ---- reference statement sequence: deepspeech_trainer.py:102-117 (fit) ----
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        o, output_sizes = model.forward(inputs, input_sizes)
        o = o.transpose(0, 1)
        float_out = o.float().log_softmax(2)
        loss = criterion(float_out, targets, output_sizes, tsz)
        loss = loss / inputs.size(0)
        loss_value = loss.item()
        # ---- deepspeech_trainer.py:86-95 ----
        opt.zero_grad()
        loss.backward()
        if step == 0:
            rec["input_sizes"] = input_sizes.numpy()
            rec["output_sizes"] = output_sizes.numpy()
            rec["logits"] = o.detach().transpose(0, 1).contiguous().numpy()   # (B,T,C)
            for k, v in taps.items():
                rec["tap_" + k] = subsample(v.numpy())
                rec["tapnorm_" + k] = np.array(float(v.double().norm()))
            for k, p in model.named_parameters():
                g = p.grad.detach().numpy()
                rec["grad_" + k] = subsample(g)
                rec["gradnorm_" + k] = np.array(float(np.sqrt((g.astype(np.float64) ** 2).sum())))
                rec["gradsum_" + k] = np.array(float(g.astype(np.float64).sum()))
            for k, v in model.state_dict().items():
                if "running_" in k:
                    rec["buf_" + k] = v.numpy().copy()
        opt.step()
        losses.append(loss_value)
    rec["losses"] = np.array(losses, dtype=np.float64)
    for k, p in model.named_parameters():
        w = p.detach().numpy()
        rec["final_" + k] = subsample(w)
    for h in hooks:
        h.remove()
    # eval-mode forward (softmax) after 3 steps, same inputs
    model.eval()
    with torch.no_grad():
        input_percentages = torch.from_numpy(pct.copy())
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        o, _ = model.forward(inputs, input_sizes)
    rec["eval_probs"] = o.numpy()
    rec["cfg"] = np.array(json.dumps(dict(rnn=rnn, hidden=hidden, layers=layers, classes=classes, t_ins=t_ins, seed=seed)))
    np.savez_compressed(os.path.join(out, f"model_{name}.npz"), **rec)
    print(name, "losses", losses)


def _full_t_ins(seed, b, lo):
    rng = np.random.default_rng(seed)
    t = sorted((int(v) for v in rng.integers(lo, 1002, size=b)), reverse=True)
    t[0] = 1001
    return t


FULL = {
    # BASELINE.json configs at their FULL size, ragged batch (one 10 s utterance, the rest lo/100 .. 10 s).
    # name: (rnn, hidden, layers, classes, t_ins, data seed)
    "c3_full": ("gru", 1024, 5, 29, _full_t_ins(5, 64, 701), 1),    # configs[2]: the metric configuration, per-GPU batch
    "c2_full": ("gru", 768, 5, 29, _full_t_ins(6, 32, 801), 1),     # configs[1]
    "dbg_full": ("gru", 32, 2, 7, [40, 33, 21], 1),                 # seconds: checks this generator itself (not committed)
}
FULL_MAX = 2048            # elements kept per tensor
LOGIT_STRIDE = 97


def sub_full(a: np.ndarray) -> np.ndarray:
    f = np.asarray(a).reshape(-1)
    return f if f.size <= FULL_MAX else f[::-(-f.size // FULL_MAX)]


def _logit_record(rec, tag, logits, out_lens):
    """per-utterance checksums over the valid frames + a strided sub-sample of them (logits beyond a length are garbage by contract, A.3)"""
    lg = np.asarray(logits, dtype=np.float64)
    rec[f"logitsum_{tag}"] = np.array([lg[b, :int(n)].sum() for b, n in enumerate(out_lens)])
    rec[f"logitnorm_{tag}"] = np.array([np.sqrt((lg[b, :int(n)] ** 2).sum()) for b, n in enumerate(out_lens)])
    rec[f"logits_{tag}"] = np.concatenate([lg[b, :int(n)].reshape(-1)[::LOGIT_STRIDE] for b, n in enumerate(out_lens)])


def gen_model_full(out, name, DeepSpeech, tmp):
    """ONE train step (fit + backward, deepspeech_trainer.py:102-117, :86-87) of the imported reference at a BASELINE configuration's full
    size, and the same step by the fp64 oracle (oracle/ds2_oracle.py with dtype=float64 - the padded + masked restatement that the small
    fixtures pin): loss, per-utterance logit checksums + sub-sample, every parameter gradient sub-sampled (<= FULL_MAX elements each) with
    its full-tensor norm and sum, and the full-tensor distance between the reference's fp32 gradient and the fp64 one - the round-off the
    reference itself carries, which is what a tolerance for ill-conditioned gradients (the conv biases in front of a BatchNorm) has to be
    measured against.  Tens of minutes on 8 cores."""
    import time
    from torch.utils.checkpoint import checkpoint
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ds2_oracle as O
    rnn, hidden, layers, classes, t_ins, seed = FULL[name]
    torch.set_num_threads(os.cpu_count() or 1)
    model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, classes),
                       rnn_type={"gru": "nn.GRU", "lstm": "nn.LSTM"}[rnn], rnn_hidden_size=hidden,
                       rnn_hidden_layers=layers, bidirectional=True)
    shapes = det.state_shapes(rnn, hidden, layers, classes)
    weights = det.model_state(shapes, base_seed=0)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()}
    model.load_state_dict(sd)
    model.train()
    x, targets, pct, tsz = det.batch(len(t_ins), t_ins, classes, seed=seed)
    inputs, targets, tsz = torch.from_numpy(x), torch.from_numpy(targets), torch.from_numpy(tsz)
    criterion = torch.nn.CTCLoss(reduction="sum")
    rec = {}
    t0 = time.time()
    cache = os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "gold", f"ref_part_{name}.npz")   # (scratch: the reference half alone is minutes)
    if os.path.exists(cache):
        zc = np.load(cache)
        rec = {k: zc[k] for k in zc.files if not k.startswith("fullgrad_")}
        gref = {k[len("fullgrad_"):]: zc[k] for k in zc.files if k.startswith("fullgrad_")}
        print(name, "reference step: loaded from", cache, flush=True)
    else:
        # ---- reference statement sequence: deepspeech_trainer.py:102-117 (fit), :86-87 (zero_grad, backward) ----
        input_percentages = torch.from_numpy(pct.copy())
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        o, output_sizes = model.forward(inputs, input_sizes)
        o = o.transpose(0, 1)
        float_out = o.float().log_softmax(2)
        loss = criterion(float_out, targets, output_sizes, tsz)
        loss = loss / inputs.size(0)
        rec["loss_ref"] = np.array(loss.item(), dtype=np.float64)
        model.zero_grad()
        loss.backward()
        print(name, "reference step", f"{time.time() - t0:.0f} s", "loss", float(rec["loss_ref"]), flush=True)
        rec["input_sizes"] = input_sizes.numpy()
        rec["output_sizes"] = output_sizes.numpy()
        _logit_record(rec, "ref", o.detach().transpose(0, 1).numpy(), output_sizes.numpy())
        gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
        for k, v in model.state_dict().items():
            if "running_" in k:
                rec["buf_" + k] = v.numpy().copy()
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        np.savez(cache, **rec, **{"fullgrad_" + k: v for k, v in gref.items()})
        del o, float_out, loss
    del model

    # ---- the same step by the fp64 oracle; the recurrences are re-run in backward (checkpoint) so that the graph of one direction at a
    # time is alive (40 GB otherwise)
    t0 = time.time()
    saved = (O.gru_direction, O.lstm_direction)
    O.gru_direction = lambda *a, _f=saved[0]: checkpoint(_f, *a, use_reentrant=False)
    O.lstm_direction = lambda *a, _f=saved[1]: checkpoint(_f, *a, use_reentrant=False)
    # (ATen has no blocked fp64 convolution on the CPU: it unfolds the whole batch, 78 GB for conv2 at B = 64 - a few utterances at a time instead,
    # the same sums)
    conv2d = torch.nn.functional.conv2d
    torch.nn.functional.conv2d = lambda a, w, b=None, **kw: (conv2d(a, w, b, **kw) if a.size(0) <= 2 else
                                                             torch.cat([conv2d(a[i:i + 2], w, b, **kw) for i in range(0, a.size(0), 2)]))
    try:
        r64 = O.fit_and_grads(sd, inputs, targets, torch.from_numpy(pct.copy()), tsz, dtype=torch.float64)
    finally:
        O.gru_direction, O.lstm_direction = saved
        torch.nn.functional.conv2d = conv2d
    print(name, "fp64 oracle step", f"{time.time() - t0:.0f} s", "loss", r64["loss"], flush=True)
    rec["loss_f64"] = np.array(r64["loss"], dtype=np.float64)
    assert np.array_equal(r64["out_lens"].numpy(), rec["output_sizes"])
    _logit_record(rec, "f64", r64["logits"].numpy(), rec["output_sizes"])
    for k, g in gref.items():
        g64 = r64["grads"][k].numpy()
        rec["grad_ref_" + k] = sub_full(g)
        rec["grad_f64_" + k] = sub_full(g64)
        rec["gradnorm_ref_" + k] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
        rec["gradnorm_f64_" + k] = np.array(np.sqrt((g64 ** 2).sum()))
        rec["gradsum_ref_" + k] = np.array(g.astype(np.float64).sum())
        rec["gradsum_f64_" + k] = np.array(g64.sum())
        rec["graddist_ref_f64_" + k] = np.array(np.sqrt(((g.astype(np.float64) - g64) ** 2).sum()))
    worst = max(gref, key=lambda k: float(rec["graddist_ref_f64_" + k] / max(float(rec["gradnorm_f64_" + k]), 1e-30)))
    print(name, "reference fp32 vs fp64: loss", abs(float(rec["loss_ref"]) - float(rec["loss_f64"])) / float(rec["loss_f64"]),
          "worst gradient", worst, float(rec["graddist_ref_f64_" + worst] / rec["gradnorm_f64_" + worst]), flush=True)
    rec["cfg"] = np.array(json.dumps(dict(rnn=rnn, hidden=hidden, layers=layers, classes=classes, t_ins=t_ins, seed=seed,
                                          full_max=FULL_MAX, logit_stride=LOGIT_STRIDE)))
    np.savez_compressed(os.path.join(out, f"model_{name}.npz"), **rec)
    print(name, os.path.getsize(os.path.join(out, f"model_{name}.npz")), "bytes", flush=True)


def gen_model_uni(out, DeepSpeech, tmp):
    """The unidirectional variant (bidirectional=False + Lookahead, deepspeech.py:83-101, blocks.py:96-132): one step of the reference's
    statement sequence -> logits, loss, every gradient, eval-mode probabilities.  Small (it pins asr_amd's torch-op branch for this variant)."""
    rnn, hidden, layers, classes, context, t_ins, seed = "gru", 16, 2, 7, 5, [40, 33, 21], 3
    model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, classes), rnn_type="nn.GRU", rnn_hidden_size=hidden,
                       rnn_hidden_layers=layers, bidirectional=False, context=context)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    weights = det.model_state(shapes, base_seed=0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    model.train()
    x, targets, pct, tsz = det.batch(len(t_ins), t_ins, classes, seed=seed)
    inputs = torch.from_numpy(x)
    input_sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int()
    o, output_sizes = model.forward(inputs, input_sizes)
    loss = torch.nn.CTCLoss(reduction="sum")(o.transpose(0, 1).float().log_softmax(2), torch.from_numpy(targets), output_sizes, torch.from_numpy(tsz))
    loss = loss / inputs.size(0)
    model.zero_grad()
    loss.backward()
    rec = {"logits": o.detach().numpy(), "loss": np.array(loss.item()), "output_sizes": output_sizes.numpy()}
    for k, p in model.named_parameters():
        rec["grad_" + k] = subsample(p.grad.detach().numpy())
        rec["gradnorm_" + k] = np.array(float(p.grad.detach().double().norm()))
    model.eval()
    with torch.no_grad():
        rec["eval_probs"] = model.forward(inputs, torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int())[0].numpy()
    rec["cfg"] = np.array(json.dumps(dict(rnn=rnn, hidden=hidden, layers=layers, classes=classes, context=context, t_ins=t_ins, seed=seed,
                                          shapes={k: list(v) for k, v in shapes.items()})))
    np.savez_compressed(os.path.join(out, "model_uni_gru_h16_l2.npz"), **rec)
    print("uni", float(rec["loss"]), os.path.getsize(os.path.join(out, "model_uni_gru_h16_l2.npz")), "bytes")


def gen_manifest(out, DeepSpeech, tmp):
    man = {}
    for rnn, hidden, layers, classes in (("gru", 32, 2, 7), ("lstm", 24, 2, 7), ("gru", 768, 5, 29)):
        model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, classes),
                           rnn_type={"gru": "nn.GRU", "lstm": "nn.LSTM"}[rnn], rnn_hidden_size=hidden,
                           rnn_hidden_layers=layers, bidirectional=True)
        sd = model.state_dict()
        man[f"{rnn}_{hidden}x{layers}_c{classes}"] = {
            "keys": {k: list(v.shape) for k, v in sd.items()},
            "param_count": int(sum(p.numel() for p in model.parameters())),
            "param_order": [k for k, _ in model.named_parameters()],
        }
    with open(os.path.join(out, "state_manifest.json"), "w") as f:
        json.dump(man, f, indent=1)


def gen_decode(out):
    from asr_deepspeech.decoders import GreedyDecoder
    rec = {}
    for name, labels, b, t, levels, sizes in det.DECODE_CASES:
        probs = torch.from_numpy(det.decode_probs(name, b, t, len(labels), levels))
        dec = GreedyDecoder(labels, blank_index=0)
        strings, offsets = dec.decode(probs, None if sizes is None else torch.tensor(sizes))
        rec[name] = {"labels": labels, "B": b, "T": t, "levels": levels, "sizes": sizes,
                     "strings": [s[0] for s in strings], "offsets": [o[0].tolist() for o in offsets]}
    with open(os.path.join(out, "decode.json"), "w") as f:
        json.dump(rec, f, indent=0)


def gen_checkpoint(out, DeepSpeech, tmp):
    """A checkpoint in the reference's own wire format (trainers/deepspeech_trainer.py:176-188: torch.save of {epoch, metrics, optimizer,
    scheduler, state_dict}) written from the REFERENCE's model class with the optimizer / scheduler wiring of trainers/__main__.py:41-52
    after one real optimizer step - plus the reference model's eval-mode output on a fixed batch, so that a loader can be checked end to
    end.  (The trainer class itself derives from `sakura`, which is not installed: the five-key dict is assembled here as its save()
    does; `metrics` is a plain dict.)"""
    rnn, hidden, layers, classes, t_ins = "gru", 16, 2, 7, [41, 33, 25]
    torch.manual_seed(11)
    model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_csv(tmp, classes), rnn_type="nn.GRU", rnn_hidden_size=hidden,
                       rnn_hidden_layers=layers, bidirectional=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.99)
    x, targets, pct, tsz = det.batch(len(t_ins), t_ins, classes, seed=5)
    inputs = torch.from_numpy(x)
    sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.size(3))).int()
    model.train()
    o, ol = model.forward(inputs, sizes)                      # (the reference overrides __call__ with its evaluation loop)
    loss = torch.nn.CTCLoss(reduction="sum")(o.transpose(0, 1).float().log_softmax(2), torch.from_numpy(targets), ol, torch.from_numpy(tsz)) / len(t_ins)
    opt.zero_grad()
    loss.backward()
    opt.step()
    sched.step()
    ckpt = {"epoch": 3, "metrics": {"test": {"best": {"cer": 12.5, "wer": 40.0}}}, "optimizer": opt.state_dict(), "scheduler": sched,
            "state_dict": model.state_dict()}
    torch.save(ckpt, os.path.join(out, "ref_checkpoint_gru_16x2_c7.pth"))
    model.eval()
    with torch.no_grad():
        probs, olens = model.forward(inputs, sizes)
    model.train()                                            # (after the eval pass: this forward moves the BatchNorm running statistics)
    o2, ol2 = model.forward(inputs, sizes)
    loss2 = torch.nn.CTCLoss(reduction="sum")(o2.transpose(0, 1).float().log_softmax(2), torch.from_numpy(targets), ol2, torch.from_numpy(tsz)) / len(t_ins)
    np.savez_compressed(os.path.join(out, "ref_checkpoint_gru_16x2_c7_eval.npz"), t_ins=np.asarray(t_ins), data_seed=np.asarray(5),
                        classes=np.asarray(classes), eval_probs=probs.numpy(), output_sizes=olens.numpy(), train_loss=np.asarray(float(loss)),
                        resume_loss=np.asarray(float(loss2)))
    print("checkpoint fixture:", os.path.getsize(os.path.join(out, "ref_checkpoint_gru_16x2_c7.pth")), "bytes")


def gen_data_formats(out, tmp):
    """SURVEY §8(f) rank 4 pin: the data-side formats and partition rules, produced by the reference's OWN classes (imported unmodified
    through _ref_shim; the only stand-ins are for file-system / audio I-O: gnutools.fs.listfiles / name and the ffmpeg-backed WAVConverter,
    which get test doubles that list the files this function creates and return fixed durations).
      * labels.csv -> SpectrogramDataset.labels_map (pandas skips the whitespace-only row: the space-row quirk) and parse_transcript outputs
        (spectrogram_dataset.py:36, :70-73);
      * BucketingSampler bins / iteration / shuffle under a seeded numpy RNG (bucketing_sampler.py:5-25);
      * DistributedBucketingSampler per-rank bins for world 1, 2, 4, 8 and the epoch-seeded shuffle (distributed_bucketing_sampler.py:8-44);
      * the JSUT ETL's manifest and labels layout: JSUTDataset.run -> filter_duration -> to_csv(index=False), export_labels
        (etl/jsut_dataset.py:28-60, etl/__main__.py:46-56)."""
    import pandas as pd
    from _ref_shim import _install
    _install()
    from asr_deepspeech.data.dataset.spectrogram_dataset import SpectrogramDataset as RefDataset
    from asr_deepspeech.data.samplers.bucketing_sampler import BucketingSampler as RefBS
    from asr_deepspeech.data.samplers.distributed_bucketing_sampler import DistributedBucketingSampler as RefDBS
    import asr_deepspeech.etl.jsut_dataset as ref_jsut

    rec = {}
    # ---- labels.csv with the space row in the middle, an apostrophe and kana; manifest with the reference's columns
    labels = ["_", "a", "b", " ", "c", "'", "\u3042", "\u3044", "d"]
    labels_csv = os.path.join(tmp, "labels_q.csv")
    pd.DataFrame({"label": labels}).to_csv(labels_csv, index=False)
    transcripts = ["ab c", "cab\n", "b?a", "_a_b", "d'\u3042\u3044c", "", " ", "xyz", "a\u3042 \u3044b"]
    manifest_csv = os.path.join(tmp, "manifest_q.csv")
    pd.DataFrame({"audio_filepath": [f"u{i}.npy" for i in range(len(transcripts))], "duration": [1.0 + 0.25 * i for i in range(len(transcripts))],
                  "fq": [16000] * len(transcripts), "text": transcripts, "text_size": [len(t) for t in transcripts]}).to_csv(manifest_csv, index=False)
    ds = RefDataset(audio_conf=audio_conf(), manifest_filepath=manifest_csv, labels=labels_csv, normalize=True)
    rec["labels_csv_text"] = open(labels_csv, encoding="utf-8").read()
    rec["manifest_csv_text"] = open(manifest_csv, encoding="utf-8").read()
    rec["labels_map"] = {k: int(v) for k, v in ds.labels_map.items()}
    rec["dataset_len"] = len(ds)
    # the manifest's text column as the reference's DataFrame holds it (pandas reads "" and " " back as NaN): parse what it can
    rec["manifest_text_isnull"] = [bool(x) for x in ds.df["text"].isnull().tolist()]
    rec["parse_transcript"] = [[t, [int(x) for x in ds.parse_transcript(t)]] for t in transcripts]

    # ---- BucketingSampler
    class _N:                                                   # a data source is only asked for its length
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
    bs_cases = []
    for n, b in ((23, 5), (16, 4), (7, 10), (1, 1)):
        smp = RefBS(_N(n), batch_size=b)
        bins0 = [list(map(int, x)) for x in smp.bins]
        np.random.seed(1234)
        it = [list(map(int, x)) for x in smp]                   # shuffles inside each bin, in place
        np.random.seed(99)
        smp.shuffle()
        bs_cases.append({"n": n, "batch_size": b, "bins": bins0, "len": len(smp), "iter_seed1234": it,
                         "bins_after_iter_then_shuffle_seed99": [list(map(int, x)) for x in smp.bins]})
    rec["bucketing_sampler"] = bs_cases

    # ---- DistributedBucketingSampler.  Its ctor calls Sampler.__init__(data_source): accepted (and ignored) by the torch the reference pins
    # (uv.lock: 2.8.0), a TypeError on this image's 2.10 — the base ctor gets the old signature back while the fixture is made.
    from torch.utils.data.sampler import Sampler as _Sampler
    _saved_init = _Sampler.__init__
    _Sampler.__init__ = lambda self, data_source=None: None
    dbs_cases = []
    for n, b in ((23, 3), (64, 8), (5, 2)):
        for world in (1, 2, 4, 8):
            per_rank, lens = [], []
            try:
                for rank in range(world):
                    smp = RefDBS(_N(n), batch_size=b, num_replicas=world, rank=rank)
                    per_rank.append([list(map(int, x)) for x in smp])
                    lens.append(len(smp))
            except AssertionError:                              # fewer bins than the wrap-around padding needs (its own assert, line 32)
                dbs_cases.append({"n": n, "batch_size": b, "world": world, "raises": "AssertionError"})
                continue
            shuf = {}
            for epoch in (0, 1, 7):
                smp = RefDBS(_N(n), batch_size=b, num_replicas=world, rank=0)
                smp.shuffle(epoch)
                shuf[str(epoch)] = {"bins": [list(map(int, x)) for x in smp.bins],
                                    "per_rank": [[list(map(int, x)) for x in _rank_iter(RefDBS, _N(n), b, world, r, epoch)] for r in range(world)]}
            dbs_cases.append({"n": n, "batch_size": b, "world": world, "per_rank": per_rank, "len": lens, "shuffle": shuf})
    rec["distributed_bucketing_sampler"] = dbs_cases
    _Sampler.__init__ = _saved_init

    # ---- JSUT ETL layout (test doubles for the file listing and the ffmpeg converter only)
    landing = os.path.join(tmp, "landing"); os.makedirs(os.path.join(landing, "basic5000"), exist_ok=True)
    lines = ["BASIC5000_0001:\u6c34\u3092 \u30de\u30ec\u30fc\u30b7\u30a2\u304b\u3089\u3001\u8cb7\u308f\u306a\u304f\u3066\u306f\u3002",
             "BASIC5000_0002:\u6728\u66dc\u65e5\u3001\u505c\u6226\u4f1a\u8ac7\u306f\u3002",
             "BASIC5000_0003:\u4e0a\u9662\u8b70\u54e1\u306f ab"]
    tpath = os.path.join(landing, "basic5000", "transcript_utf8.txt")
    open(tpath, "w", encoding="utf-8").write("\n".join(lines) + "\n")
    durs = {"BASIC5000_0001": 3.19, "BASIC5000_0002": 0.9, "BASIC5000_0003": 4.5}
    saved = (ref_jsut.listfiles, ref_jsut.name, ref_jsut.WAVConverter)

    class _Conv:
        def __init__(self, landing, bronze, fq): self.bronze = bronze
        def run(self): return [(os.path.join(self.bronze, "basic5000", "wav", k + ".wav"), d) for k, d in durs.items()]
    ref_jsut.listfiles = lambda root, pats=None: [tpath]
    ref_jsut.name = lambda f: os.path.splitext(os.path.basename(f))[0]
    ref_jsut.WAVConverter = _Conv
    try:
        etl = ref_jsut.JSUTDataset(16000).run(landing, "/bronze")
        kept = etl.filter_duration(1, 5)
        mpath, lpath = os.path.join(tmp, "etl_manifest.csv"), os.path.join(tmp, "etl_labels.csv")
        kept.to_csv(mpath, index=False)                         # etl/__main__.py:55
        etl.export_labels(lpath)
        rec["etl"] = {"transcript_lines": lines, "durations": durs, "manifest_csv_text": open(mpath, encoding="utf-8").read(),
                      "manifest_columns": list(kept.columns), "labels_csv_header": open(lpath, encoding="utf-8").read().splitlines()[0],
                      "labels_set": sorted(pd.read_csv(lpath)["label"].tolist()),
                      "clean_text": [[l, list(ref_jsut.JSUTDataset.clean_text(l.split(":")))] for l in lines]}
    finally:
        ref_jsut.listfiles, ref_jsut.name, ref_jsut.WAVConverter = saved
    with open(os.path.join(out, "data_formats.json"), "w", encoding="utf-8") as f:
        json.dump(rec, f, ensure_ascii=True, indent=0, sort_keys=True)
    print("data_formats.json:", os.path.getsize(os.path.join(out, "data_formats.json")), "bytes")


def _rank_iter(cls, src, b, world, rank, epoch):
    smp = cls(src, batch_size=b, num_replicas=world, rank=rank)
    smp.shuffle(epoch)
    return list(smp)


def main():
    DeepSpeech, blocks, functional = import_reference()
    torch.set_num_threads(4)
    out = HERE
    if "--only-decode" in sys.argv:
        gen_decode(out)
        return
    if "--only-data-formats" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            gen_data_formats(out, tmp)
        return
    if "--only-checkpoint" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            gen_checkpoint(out, DeepSpeech, tmp)
        return
    if "--only-uni" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            gen_model_uni(out, DeepSpeech, tmp)
        return
    if "--only-full" in sys.argv:                      # model_c3_full.npz / model_c2_full.npz: tens of minutes each, not part of the default run
        with tempfile.TemporaryDirectory() as tmp:
            gen_model_full(out, sys.argv[sys.argv.index("--only-full") + 1], DeepSpeech, tmp)
        return
    if "--only-model" in sys.argv:                     # regenerate one model fixture without touching the others
        with tempfile.TemporaryDirectory() as tmp:
            gen_model(out, sys.argv[sys.argv.index("--only-model") + 1], DeepSpeech, tmp)
        return
    gen_decode(out)
    with tempfile.TemporaryDirectory() as tmp:
        gen_ctc(out)
        gen_lengths(out, DeepSpeech, functional, tmp)
        gen_manifest(out, DeepSpeech, tmp)
        for name in MODELS:
            gen_model(out, name, DeepSpeech, tmp)
        gen_checkpoint(out, DeepSpeech, tmp)
        gen_data_formats(out, tmp)
        gen_model_uni(out, DeepSpeech, tmp)
    for f in sorted(os.listdir(out)):
        if f.endswith((".npz", ".json")):
            print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
