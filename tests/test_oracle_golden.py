"""CPU: pin the oracle (oracle/ds2_oracle.py) against golden vectors produced by the unmodified
reference (tests/golden/make_golden.py).  Tolerances are fp32 round-off class."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, MODEL_FIXTURES, load_model_fixture, model_inputs, noise_only_grads, rel_l2, subsample
from oracle import ds2_oracle as O
import det


def test_seq_lens_table():
    z = np.load(f"{GOLDEN}/lengths.npz")
    L = torch.arange(1, 2002, dtype=torch.int32)
    got = O.seq_lens_after_conv(L).numpy()
    assert np.array_equal(got, z["seq_lens"])
    # closed form quoted in SURVEY A.7
    assert np.array_equal(got, (np.arange(1, 2002) + 1) // 2)


@pytest.mark.parametrize("tmax", [201, 501, 1001, 1501, 2001])
def test_length_recovery_quirk(tmax):
    z = np.load(f"{GOLDEN}/lengths.npz")
    tb = np.arange(1, tmax + 1)
    pct = torch.tensor([t / float(tmax) for t in tb], dtype=torch.float64).to(torch.float32)
    got = O.lengths_from_percentages(pct, tmax).numpy()
    assert np.array_equal(got, z[f"rec_{tmax}"])
    if tmax in (201, 1001):  # the quirk exists there (7->6 @201, 127->126 @1001)
        assert (got != tb).sum() > 0 and (got <= tb).all()


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_ctc_numpy_restatement(case):
    z = np.load(f"{GOLDEN}/ctc.npz")
    T, B, C = int(z[f"{case}_T"]), int(z[f"{case}_B"]), int(z[f"{case}_C"])
    scale = float(z[f"{case}_scale"]) if f"{case}_scale" in z else 1.0
    logits = det.unitvar((T, B, C), int(z[f"{case}_seed"])) * np.float32(scale)
    lp = torch.from_numpy(logits).double().log_softmax(2).numpy()
    nll, grad = O.ctc_nll_and_grad_np(lp, z[f"{case}_targets"], z[f"{case}_il"], z[f"{case}_tl"])
    ref = z[f"{case}_nll"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isinf(nll), ~fin)
    assert np.allclose(nll[fin], ref[fin], rtol=2e-6, atol=1e-5)
    if f"{case}_grad" in z:
        assert rel_l2(grad, z[f"{case}_grad"]) < 5e-5  # golden is fp32 log-space, oracle fp64


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_whole_model_matches_reference(name):
    z, cfg = load_model_fixture(name)
    sd, x, targets, pct, tsz = model_inputs(cfg)
    res = O.fit_and_grads(sd, x, targets, pct, tsz, dtype=torch.float32)
    assert np.array_equal(res["input_sizes"].numpy(), z["input_sizes"])
    assert np.array_equal(res["out_lens"].numpy(), z["output_sizes"])
    # logits: compare valid frames only (t >= out_len rows are "garbage" the CTC ignores, but the
    # reference still computes them deterministically from BN of zero rows -> compare all)
    assert rel_l2(res["logits"].numpy(), z["logits"]) < 2e-5
    assert abs(res["loss"] - z["losses"][0]) / z["losses"][0] < 1e-5
    for k, g in res["grads"].items():
        if k in noise_only_grads(cfg):                      # analytically zero: both sides hold round-off only
            assert float(g.double().norm()) <= 1e-5 * float(z["gradnorm_" + k.replace(".bias", ".weight")]), k
            continue
        ref = z["grad_" + k]
        got = subsample(g.numpy())
        nrm = float(z["gradnorm_" + k])
        err = np.linalg.norm(got.astype(np.float64) - ref.astype(np.float64))
        ref_n = np.linalg.norm(ref.astype(np.float64))
        assert err <= 2e-4 * max(ref_n, 1e-6 * max(nrm, 1e-30)) + 1e-9, (k, err, ref_n)
        gn = float((g.double() ** 2).sum().sqrt())
        assert abs(gn - nrm) <= 2e-4 * nrm + 1e-9, (k, gn, nrm)


@pytest.mark.parametrize("name", ["gru_h32_l2", "lstm_h24_l2"])
def test_running_stats_and_eval(name):
    """BN running statistics after one training step, and eval-mode forward after 3 AdamW steps
    (weights advanced with the oracle's AdamW restatement)."""
    z, cfg = load_model_fixture(name)
    sd, x, targets, pct, tsz = model_inputs(cfg)
    m = {k: np.zeros(v.shape, np.float64) for k, v in sd.items() if v.is_floating_point()}
    v2 = {k: np.zeros(v.shape, np.float64) for k, v in sd.items() if v.is_floating_point()}
    for step in range(1, 4):
        res = O.fit_and_grads(sd, x, targets, pct, tsz, dtype=torch.float32)
        assert abs(res["loss"] - z["losses"][step - 1]) / z["losses"][step - 1] < 5e-5
        # running stats (momentum 0.1, unbiased variance) — checked after the first step
        for key in list(sd.keys()):
            if key.endswith("running_mean"):
                base = key[: -len(".running_mean")]
                mu, var, n = res["stats"][base + ".batch_mean"], res["stats"][base + ".batch_var"], res["stats"][base + ".count"]
                sd[key] = 0.9 * sd[key] + 0.1 * mu
                sd[base + ".running_var"] = 0.9 * sd[base + ".running_var"] + 0.1 * var * (n / (n - 1))
                if step == 1:
                    assert np.allclose(sd[key].numpy(), z["buf_" + key], rtol=1e-4, atol=1e-6)
                    assert np.allclose(sd[base + ".running_var"].numpy(), z["buf_" + base + ".running_var"], rtol=1e-4, atol=1e-6)
        for k, g in res["grads"].items():
            p, m[k], v2[k] = O.adamw_step_np(sd[k].double().numpy(), g.double().numpy(), m[k], v2[k], step)
            sd[k] = torch.from_numpy(p).float()
    for k in res["grads"]:
        assert rel_l2(subsample(sd[k].numpy()), z["final_" + k]) < 1e-5, k
    lens = O.lengths_from_percentages(pct, x.size(3))
    with torch.no_grad():
        probs, _ = O.forward(sd, x, lens, training=False)
    assert rel_l2(probs.numpy(), z["eval_probs"]) < 1e-4


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_packed_form_baseline_matches_reference_golden(name):
    """oracle/ds2_packed.py — the reference's OWN formulation (pack_padded_sequence -> fused gru/lstm -> pad_packed_sequence, torch AdamW;
    what bench.py's cpu_baseline times) — against the vectors generated from the imported reference: logits, loss,
    sub-sampled gradients at step 0, and the 3-step AdamW loss curve + final weights + BN running statistics."""
    from oracle import ds2_packed as P
    z, cfg = load_model_fixture(name)
    sd, x, targets, pct, tsz = model_inputs(cfg)
    params = P.leaf_params(sd)
    out, out_lens, loss = P.fit(params, x, targets, pct, tsz)
    assert np.array_equal(out_lens.numpy(), z["output_sizes"])
    assert rel_l2(out.detach().numpy(), z["logits"]) < 2e-5
    assert abs(float(loss.detach()) - float(z["losses"][0])) / float(z["losses"][0]) < 2e-5
    keys = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in keys])
    for k, g in zip(keys, grads):
        if k in noise_only_grads(cfg):
            assert float(g.double().norm()) <= 1e-5 * float(z["gradnorm_" + k.replace(".bias", ".weight")]), k
            continue
        ref = z["grad_" + k]
        err = np.linalg.norm(subsample(g.numpy()).astype(np.float64) - ref.astype(np.float64))
        assert err <= 2e-4 * max(np.linalg.norm(ref.astype(np.float64)), 1e-6 * float(z["gradnorm_" + k])) + 1e-9, k
    if "losses" in z.files:
        params = P.leaf_params(sd)                                  # fresh buffers: the probe above already moved the running statistics
        opt = P.make_optimizer(params)
        nthreads = torch.get_num_threads()
        torch.set_num_threads(min(4, nthreads))                     # the fixtures' thread count (summation order of the conv gradients)
        try:
            for step in range(3):
                value = P.train_step(params, opt, (x, targets, pct.clone(), tsz))
                assert abs(value - z["losses"][step]) / z["losses"][step] < 5e-5
        finally:
            torch.set_num_threads(nthreads)
        for k in keys:
            if "final_" + k in z.files and k not in noise_only_grads(cfg):
                # (AdamW's lr * m / sqrt(v) update turns round-off in a tiny gradient — the fixtures were generated with 4 threads,
                #  the test runs with however many the host has — into a full-size step difference: 1e-4, not 1e-5)
                assert rel_l2(subsample(params[k].detach().numpy()), z["final_" + k]) < 2e-4, k


# ---- spectrogram front-end oracle: librosa itself is absent, so the restatement is cross-checked against
# ---- two independent implementations that are present in the image, and the reference's own test properties.
def _wave(seed, n):
    t = np.arange(n) / 16000.0
    return (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * det.unitvar((n,), seed)).astype(np.float32)


@pytest.mark.parametrize("n,pad_mode", [(16000, "constant"), (16000, "reflect"), (4321, "constant"), (161, "constant")])
def test_stft_oracle_vs_torch_and_scipy(n, pad_mode):
    from scipy.signal import stft as sp_stft, get_window
    from oracle import stft_oracle as S
    y = _wave(3, n)
    ref = S.stft_log_spectrogram(y, 320, 160, "hamming", pad_mode)
    assert ref.shape == (161, 1 + n // 160)                                      # tests/test_spectrogram_dataset.py:37-46
    win = torch.from_numpy(get_window("hamming", 320, fftbins=True))
    D = torch.stft(torch.from_numpy(y).double(), 320, hop_length=160, win_length=320, window=win, center=True, pad_mode=pad_mode,
                   normalized=False, onesided=True, return_complex=True)
    assert np.allclose(np.log1p(D.abs().numpy()), ref, rtol=1e-9, atol=1e-9)
    if pad_mode == "constant" and n % 160 == 0:
        # scipy: boundary="zeros" is the same centring; it scales by 1/sum(window) and appends whole padded segments
        _, _, Z = sp_stft(y.astype(np.float64), window="hamming", nperseg=320, noverlap=160, boundary="zeros", padded=False)
        Z = np.abs(Z) * get_window("hamming", 320, fftbins=True).sum()
        assert np.allclose(np.log1p(Z[:, :ref.shape[1]]), ref, rtol=1e-9, atol=1e-9)
    norm = S.stft_log_spectrogram(y, 320, 160, "hamming", pad_mode, normalize=True)
    t = torch.from_numpy(ref).float()
    assert np.allclose(norm, ((t - t.mean()) / t.std()).numpy(), atol=2e-5)      # spectrogram_parser.py:56-60 (torch: unbiased std)
    assert abs(norm.mean()) < 1e-3 and np.isfinite(norm).all()                    # tests/test_spectrogram_dataset.py:49-58


def test_host_dataset_stft_matches_oracle():
    from asr_amd.data import _stft_spectrogram
    from oracle import stft_oracle as S
    y = _wave(5, 12345)
    for pad_mode in ("constant", "reflect"):
        got = _stft_spectrogram(y, 16000, 0.02, 0.01, "hamming", pad_mode)
        assert got.dtype == np.float32 and np.allclose(got, S.stft_log_spectrogram(y, 320, 160, "hamming", pad_mode), atol=2e-5)
