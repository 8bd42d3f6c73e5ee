"""-m gpu, round 4: the co-resident grouped weight-gradient kernel (csrc/gemm_tn_group.h) and the side-stream schedule that runs it beside
the backward recurrence of the layer below (asr_amd/engine.py, DS2_WGRAD_SIDE)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref_tn(A, B):
    """fp64 reference of A^T B on the bf16-rounded operands (CPU)."""
    return (A.double().cpu().t() @ B.double().cpu()).float()


@pytest.mark.parametrize("K,M,N,lda,ldb", [
    (64, 128, 128, None, None),          # one tile, one k-tile
    (130, 8, 8, None, None),             # smaller than a tile in every direction, K tail of 2 rows
    (200, 264, 328, None, None),         # ragged edges in M and N, K tail
    (1000, 520, 1312, None, None),       # layer 0's I = 1312 (not a multiple of 128)
    (4096, 512, 256, 600, 304),          # pitched operands (column slices of wider buffers)
    (5031, 768, 1024, 1024, 2048),       # odd K, pitched
])
def test_tn_group_single_problem_vs_fp64(K, M, N, lda, ldb):
    from asr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(K + M + N)
    A = torch.randn(K, lda or M, device="cuda", generator=g).bfloat16()[:, :M]
    B = torch.randn(K, ldb or N, device="cuda", generator=g).bfloat16()[:, :N]
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_bf16_tn_group([(A, B, out)])
    ref = _ref_tn(A, B)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 * 4 + 1e-4, (err, ref.abs().max().item())       # fp32 accumulation of exact bf16 products
    # against the 256 x 256 split-K kernel (other summation order, same operands)
    other = ops.gemm_bf16_tn(A, B) if (M % 8 == 0 and N % 8 == 0) else None
    if other is not None:
        assert (out - other).abs().max().item() <= 2e-5 * K ** 0.5 * 4 + 1e-4


def test_tn_group_layer_problem_list_matches_separate_launches_and_is_deterministic():
    """The five products of one GRU layer's weight gradients (dW_ih; dW_hh r,z rows and n rows of both directions: row-shifted, column-sliced
    views of three buffers, two different K) in ONE launch: every output equals the fp64 reference and the single-problem launch bit for
    bit (a tile's result does not depend on which workgroup computes it or on what else is in the list), repeated launches are identical,
    and nothing outside the outputs is written."""
    from asr_amd import ops
    T, Bt, H, G, I = 23, 16, 256, 3, 328
    Mr = T * Bt
    g = torch.Generator(device="cuda").manual_seed(7)
    dgx = torch.randn(Mr, 2 * G * H, device="cuda", generator=g).bfloat16()
    dhn = torch.randn(Mr, 2 * H, device="cuda", generator=g).bfloat16()
    h = torch.randn(Mr, 2 * H, device="cuda", generator=g).bfloat16()
    xn = torch.randn(Mr, I, device="cuda", generator=g).bfloat16()
    guard = 64
    flat = torch.full((2 * G * H * I + 2 * G * H * H + 2 * guard,), 7.0, device="cuda")
    dwih = flat[guard:guard + 2 * G * H * I].view(2 * G * H, I)
    dwhh = flat[guard + 2 * G * H * I:guard + 2 * G * H * I + 2 * G * H * H].view(2, G * H, H)
    rows = 2 * H

    def problems(dwih, dwhh):
        return [(dgx, xn, dwih),
                (dgx[Bt:, 0:rows], h[:Mr - Bt, 0:H], dwhh[0, :rows]), (dgx[:Mr - Bt, G * H:G * H + rows], h[Bt:, H:2 * H], dwhh[1, :rows]),
                (dhn[Bt:, 0:H], h[:Mr - Bt, 0:H], dwhh[0, rows:]), (dhn[:Mr - Bt, H:2 * H], h[Bt:, H:2 * H], dwhh[1, rows:])]

    ops.gemm_bf16_tn_group(problems(dwih, dwhh))
    torch.cuda.synchronize()
    assert float(flat[:guard].min()) == 7.0 and float(flat[-guard:].max()) == 7.0
    first = flat.clone()
    for A, Bm, out in problems(dwih, dwhh):
        ref = _ref_tn(A, Bm)
        assert (out.cpu() - ref).abs().max().item() <= 2e-3, (A.shape, Bm.shape)
        single = torch.empty(out.shape, device="cuda")
        ops.gemm_bf16_tn_group([(A, Bm, single)])
        assert torch.equal(single, out)
    for wg in (0, 8, 24, 64):                                   # any grid walks the same tiles: bit-identical
        flat.fill_(3.0)
        ops.gemm_bf16_tn_group(problems(dwih, dwhh), max_workgroups=wg)
        assert torch.equal(flat[guard:-guard], first[guard:-guard]), wg


def _one_backward(mode, cfg, B, tin, seed=0, steps=1):
    """sha256 of every gradient after one forward + backward of a bf16 model on a fixed batch under engine.WGRAD_SIDE = mode"""
    sys.path.insert(0, ROOT)
    import bench
    from test_gpu_model import make_model
    from asr_amd import engine, ops
    from asr_amd.trainers.deepspeech_trainer import _prep_targets_host
    old = engine.WGRAD_SIDE
    engine.WGRAD_SIDE = mode
    try:
        torch.manual_seed(seed)
        model = make_model(cfg)
        model.precision = "bf16"
        x, targets, pct, tsz = bench.synthetic_batch(B, tin, cfg["classes"], 1)
        x = x.cuda()
        model._ensure_flat(x.device)
        out = None
        for _ in range(steps):
            with torch.no_grad():
                W = model._flat.tensors(model)
                Gr = model._flat.tensors(model, grads=True)
                out_sizes = model.get_seq_lens((pct * tin).int())
                t_h, off_h, tl_h, max_u = _prep_targets_host(targets, tsz)
                lens_dev, tg, off, tl = out_sizes.to(torch.int32).cuda(), t_h.cuda(), off_h.cuda(), tl_h.cuda()
                logits, ctx = engine.forward(W, model._cfg, x, lens_dev, training=True, save=True)
                nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B, want_grad=True)
                engine.backward(W, Gr, model._cfg, ctx, dlogits)
                torch.cuda.synchronize()
                path = ops.rnn_last_path()
                _, grad = model.flat_parameters()
                out = (hashlib.sha256(grad.detach().cpu().numpy().tobytes()).hexdigest(), grad.detach().clone(), path, float(nll.sum()))
        ops.rnn_persistent_check()
        return out
    finally:
        engine.WGRAD_SIDE = old


def test_side_stream_weight_gradients_bit_identical_to_one_stream_schedule():
    """c3's layer shape (3 x 1024 BiGRU, B = 64: 256-workgroup persistent recurrences, the K-split backward kernel): the gradients of the
    side-stream schedule (grouped weight-gradient launches running BESIDE the backward recurrence of the layer below) equal, bit for bit, the
    one-stream schedule of the same kernels — twice, so that a race would have two chances; both are within the K-split / split-K summation
    tolerance of round 3's kernels; no persistent launch starved."""
    from asr_amd import ops
    cfg = dict(rnn="gru", hidden=1024, layers=3, classes=29)
    if not ops.wgrad_fits_beside_bwd_recurrence(3, 1024):
        pytest.skip("the loaded K-split kernel leaves no room for the co-resident kernel")
    main = _one_backward("main", cfg, 64, 301)
    side = _one_backward("1", cfg, 64, 301)
    side2 = _one_backward("1", cfg, 64, 301)
    old = _one_backward("0", cfg, 64, 301)
    assert main[2] & 6 == 6 and side[2] & 6 == 6, (main[2], side[2])           # persistent K-split backward in both
    assert main[0] == side[0] == side2[0], "side-stream gradients differ from the one-stream schedule"
    assert main[3] == side[3] == old[3]
    d = (main[1] - old[1]).norm().item() / old[1].norm().item()
    assert d < 1e-4, d                                                          # same operands, other fp32 summation order


def test_grouped_splitk_weight_gradients_match_separate_launches():
    """DS2_WGRAD_SIDE=sk (the default): the five products of a layer in ONE launch of the 256 x 256 TN kernel with a common split-K factor +
    one reduce launch.  Against round 3's three launches: same operands, same MFMA kernel, another split of K -> other fp32 summation order
    (<= 1e-5 relative on every tensor, loss identical); reruns bit-identical; the stand-alone call equals ds2_gemm_bf16_tn with the same
    split factor to the bit (same slices, same order of the slab sum)."""
    from asr_amd import ops
    cfg = dict(rnn="gru", hidden=1024, layers=3, classes=29)
    sk = _one_backward("sk", cfg, 64, 301)
    sk2 = _one_backward("sk", cfg, 64, 301)
    old = _one_backward("0", cfg, 64, 301)
    assert sk[0] == sk2[0] and sk[3] == old[3]
    assert (sk[1] - old[1]).norm().item() / old[1].norm().item() < 1e-5
    # LSTM (three products) and a shape with a K tail
    for c, B, tin in ((dict(rnn="lstm", hidden=1280, layers=2, classes=29), 32, 161), (dict(rnn="gru", hidden=768, layers=2, classes=29), 24, 203)):
        a, b = _one_backward("sk", c, B, tin), _one_backward("0", c, B, tin)
        assert (a[1] - b[1]).norm().item() / b[1].norm().item() < 1e-5, c
    g = torch.Generator(device="cuda").manual_seed(3)
    K, M, N = 5000, 768, 512
    A = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    Bm = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    one, ref = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    used = ops.gemm_bf16_tn_splitk_group([(A, Bm, one)], splitk=3)
    ops.gemm_bf16_tn(A, Bm, out=ref, splitk=used)
    assert torch.equal(one, ref)
    # K a multiple of the 64-deep k-tile: the grouped launch takes the FOUR-wave kernel (gemm_tn_w4.h: 128 x 128 per wave, 16x16x32 MFMA,
    # operands by two ds_read_b64_tr_b16 each); ds2_gemm_bf16_tn stays on the 8-wave kernel — same slices, same order: equal to the bit,
    # ragged M / N edges and column-sliced operands included
    for K, M, N, sk in ((4992, 768, 512, 3), (8192, 1048, 2056, 2), (640, 264, 8, 1)):
        A = torch.randn(K, M + 16, device="cuda", generator=g).bfloat16()[:, 16:]
        Bm = torch.randn(K, N + 8, device="cuda", generator=g).bfloat16()[:, :N]
        one, ref = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        used = ops.gemm_bf16_tn_splitk_group([(A, Bm, one)], splitk=sk)
        ops.gemm_bf16_tn(A, Bm, out=ref, splitk=used)
        assert torch.equal(one, ref), (K, M, N, float((one - ref).abs().max()))
        again = torch.empty_like(one)
        ops.gemm_bf16_tn_splitk_group([(A, Bm, again)], splitk=sk)
        assert torch.equal(one, again)


def test_side_stream_schedule_lstm_and_narrow_shapes_fall_back_cleanly():
    """Shapes whose backward recurrence leaves no room (LSTM H = 1024: 2 x 224 registers) or that have no K-split kernel keep a one-stream
    schedule; the result is the same in every mode."""
    from asr_amd import ops
    for cfg, B, tin in ((dict(rnn="lstm", hidden=1024, layers=2, classes=29), 32, 161), (dict(rnn="gru", hidden=256, layers=2, classes=29), 16, 121)):
        a = _one_backward("1", cfg, B, tin)
        b = _one_backward("main", cfg, B, tin)
        assert a[0] == b[0], cfg


def test_bench_two_ranks_end_to_end_on_one_gpu():
    """The N > 1 branch of bench.py, executed before multi-GPU hardware sees it: `python bench.py --gpus 2 --ranks-on-one-gpu` self-spawns two
    ranks under torch.distributed.run (both on cuda:0, gloo instead of RCCL), they run barrier -> K steps -> barrier, the MAX over ranks of
    the wall clock is reported, and EXACTLY ONE JSON line comes out (rank 0) with n_gpus 2, global_batch 2 x B, parallelism dp2 and the
    `dist` diagnostics (world size, backend, per-rank ms per step, the big collective's event-timed duration and how much of it outlasted the
    conv-stack backward)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "3", "--warmup", "1",
                        "--ranks-on-one-gpu", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * 4 and out["config"]["parallelism"] == "dp2"
    d = out["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["schedule"] == "conv" and "ranks_on_one_gpu" in d
    assert 0 < d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"]
    assert abs(d["rank_ms_per_step"]["max"] - out["ms_per_step"]) < 1e-6 * max(1.0, out["ms_per_step"])        # the line reports the MAX
    assert abs(out["value"] - 2 * 4 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]                  # whole-job utterances / s
    assert d["collectives_per_step"] == 2 and d["big_collective_bytes"] > 0 and d["big_collective_ms"] > 0    # fc + rnns bucket, conv bucket
    # first-SCALE-run readiness (round 5): the line says which devices the ranks really ran on and what starved launches cost them
    assert d["device_count"] >= 1 and isinstance(d["rccl_version"], str) and len(d["ranks"]) == 2
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["device_index"] == 0 and r["pci_bus_id"] for r in d["ranks"])
    assert d["distinct_devices"] == 1                                           # test mode: both ranks on cuda:0 — a real run reports N
    # (summed over the ranks; NOT necessarily 0 here — see below: two processes time-share the one GPU in this test mode)
    assert 0 <= d["persistent_starved_steps_all_ranks"] <= 2 * 4 and d["persistent_starved_steps_all_ranks"] >= out["persistent_starved_steps"]
    assert d["conv_backward_ms"] > 0 and d["big_collective_outlasts_conv_backward_ms"] >= 0
    # two PROCESSES time-share the one GPU here, so a persistent recurrence launch of one rank can find its CUs held by the other rank's kernels
    # for longer than the spin limit: that is the starvation path doing its job (every rank skips the step, restores the BatchNorm statistics,
    # a cooldown runs the step kernels) — it cannot happen with one process per GPU, and it must leave the run consistent
    starved = out["persistent_starved_steps"]
    assert np.isfinite(out["loss"]) and 0 <= starved <= 4
    assert out["valid_last_step"] or starved > 0


FORWARD_ONLY_WORKER = r'''
import os, sys, json
import torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import bench
from test_gpu_model import make_model
from asr_amd import ops, _lib
torch.manual_seed(0)
model = make_model(dict(rnn="gru", hidden=256, layers=2, classes=29))
model.precision = "bf16"
model.eval()
x, targets, pct, tsz = bench.synthetic_batch(16, 101, 29, 1)
x = x.cuda()
lens = (pct * x.size(3)).int()
res = {"seen_before": ops.rnn_poison_seen()}
with torch.no_grad():
    out, _ = model.forward(x, lens)                         # persistent launch, starves at its first failed poll (DS2_RNN_SPIN_LIMIT=0)
res["persistent"] = bool(ops.rnn_last_path() & 1)
res["nan"] = bool(torch.isnan(out).all())                   # (this read synchronises: the poison kernel has run)
res["seen_after_poison"] = ops.rnn_poison_seen()
try:
    with torch.no_grad():
        model.forward(x, lens)                              # a caller that only ever calls forward(): the NEXT call settles and raises
    res["second_forward_raised"] = False
except _lib.DS2LibraryError as e:
    res["second_forward_raised"] = True
    res["message_names_step_kernels"] = "one-launch-per-step" in str(e)
res["seen_after_settle"] = ops.rnn_poison_seen()
res["cooldown"] = ops.rnn_persistent_counters()[1]
with torch.no_grad():
    out3, _ = model.forward(x, lens)                        # cooldown: step kernels, valid results, no exception
res["third_finite"] = bool(torch.isfinite(out3).all())
res["third_on_step_kernels"] = not bool(ops.rnn_last_path() & 1)
res["seen_end"] = ops.rnn_poison_seen()
print("FWD_JSON " + json.dumps(res))
'''


def test_forward_only_inference_notices_and_heals_a_starved_launch(tmp_path):
    """a caller that uses `model(x)` directly in eval mode and never calls evaluate() / the trainer / ops.rnn_persistent_check.
    A starved persistent launch (forced) poisons that forward's logits in stream order (no synchronisation); the poison kernel also raises a
    pinned host flag, so the NEXT forward notices with a plain host read, settles the record (DS2LibraryError naming the launch, cooldown
    onto the step kernels) and every forward after that returns valid results again — instead of NaN for the rest of the process."""
    import json
    import subprocess
    script = str(tmp_path / "fwd_only.py")
    open(script, "w").write(FORWARD_ONLY_WORKER)
    env = dict(os.environ, DS2_RNN_SPIN_LIMIT="0", DS2_RNN_REARM_CALLS="8")
    r = subprocess.run([sys.executable, script, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("FWD_JSON ")][-1][len("FWD_JSON "):])
    assert not res["seen_before"] and res["persistent"] and res["nan"] and res["seen_after_poison"], res
    assert res["second_forward_raised"] and res["message_names_step_kernels"] and not res["seen_after_settle"] and res["cooldown"] > 0, res
    assert res["third_finite"] and res["third_on_step_kernels"] and not res["seen_end"], res


@pytest.mark.parametrize("M,N,K", [(1024, 768, 1312), (2048, 4608, 768), (520, 264, 328)])
def test_split_bf16_three_term_products_vs_fp64(M, N, K):
    """The fp32 mode's large GEMMs (DS2_F32_GEMM=split): operands split as x = hi + lo (two bf16 terms), product = hi.hi + hi.lo + lo.hi with fp32
    accumulation.  Against the fp64 product of the fp32 operands the relative L2 error stays ~1e-5 (the dropped lo.lo term and the two
    representation errors are 2^-18 each) — two orders inside the 1e-3 the fp32 mode is held to — for the NT form (one GEMM over 3K) and for
    the TN form (three accumulating launches on row-pitched views); the plain bf16 product of the same operands is ~3e-3."""
    from asr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g) * 3.0
    Bm = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    ref = (A.double() @ Bm.double().t()).cpu()
    # the split itself: hi + lo reproduces x to 2^-17 relative
    s2 = ops.split_bf16(A, 2)
    Kp = (K + 7) // 8 * 8
    assert s2.shape == (M, 2 * Kp)
    rec = s2[:, :K].float() + s2[:, Kp:Kp + K].float()
    assert float(((rec - A).abs() / A.abs().clamp_min(1e-30)).max()) < 2.0 ** -16
    out = ops.gemm_bf16_nt(ops.split_bf16(A, 0), ops.split_bf16(Bm, 1))
    e_split = float((out.double().cpu() - ref).norm() / ref.norm())
    e_f32 = float((ops.gemm(A, Bm, transB=True).double().cpu() - ref).norm() / ref.norm())
    e_bf16 = float((ops.gemm_bf16_nt(ops.cast_bf16(A), ops.cast_bf16(Bm)).double().cpu() - ref).norm() / ref.norm())
    print(f"NT M={M} N={N} K={K}: split-bf16 {e_split:.2e}  fp32 kernel {e_f32:.2e}  plain bf16 {e_bf16:.2e}")
    assert e_split < 2e-5 and e_bf16 > 20 * e_split
    # TN form: C[N2, K] = X^T Y with X (M, N2), Y (M, K) sharing the M rows (the weight-gradient products), K and N2 multiples of 8 here
    if K % 8 == 0 and N % 8 == 0:
        X = torch.randn(M, N, device="cuda", generator=g)
        refT = (X.double().t() @ A.double()).cpu()
        xs, ys = ops.split_bf16(X, 0), ops.split_bf16(A, 2)
        Np = N
        outT = torch.empty(N, K, device="cuda")
        for k, (a, b) in enumerate(((xs[:, :N], ys[:, :K]), (xs[:, :N], ys[:, Kp:Kp + K]), (xs[:, 2 * Np:], ys[:, :K]))):
            ops.gemm_bf16_tn(a, b, out=outT, accumulate=k > 0)
        e_tn = float((outT.double().cpu() - refT).norm() / refT.norm())
        print(f"TN: split-bf16 {e_tn:.2e}")
        assert e_tn < 2e-5


def test_grouped_tn_kernels_fuzz():
    """Random problem lists through both grouped TN kernels (the 256 x 256 split-K one and the co-resident 128 x 128 one): 1-6 products per launch,
    M / N / K anywhere (multiples of 8 for M and N; K with tails), operands as column slices of wider buffers with row offsets (the shapes
    the weight gradients hand over), outputs as slices of a guarded flat buffer.  Every product against the fp32 torch product of the same
    bf16 operands; nothing outside the outputs is written; a second launch is bit-identical."""
    from asr_amd import ops
    rng = np.random.default_rng(4)
    g = torch.Generator(device="cuda").manual_seed(4)
    for trial in range(12):
        nprob = int(rng.integers(1, 7))
        probs, refs = [], []
        total = sum(1 for _ in range(0))
        sizes = []
        for _ in range(nprob):
            M, N = 8 * int(rng.integers(1, 90)), 8 * int(rng.integers(1, 90))
            K = int(rng.integers(1, 1400)) if trial % 3 else 64 * int(rng.integers(1, 20))
            sizes.append((M, N, K))
        flat = torch.full((sum(M * N for M, N, _ in sizes) + 2 * 64,), 5.0, device="cuda")
        off = 64
        for M, N, K in sizes:
            pa, pb = M + 8 * int(rng.integers(0, 4)), N + 8 * int(rng.integers(0, 4))
            r0 = int(rng.integers(0, 3)) * 8
            A = torch.randn(K + r0, pa, device="cuda", generator=g).bfloat16()[r0:, pa - M:]
            Bm = torch.randn(K + r0, pb, device="cuda", generator=g).bfloat16()[r0:, :N]
            out = flat[off:off + M * N].view(M, N)
            off += M * N
            probs.append((A, Bm, out))
            refs.append(A.float().t() @ Bm.float())
        kmin = min(k for _, _, k in sizes)
        for name, run in (("splitk_group", lambda: ops.gemm_bf16_tn_splitk_group(probs, splitk=int(rng.integers(1, 4)) if kmin >= 256 else 1)),
                          ("coresident_group", lambda: ops.gemm_bf16_tn_group(probs, max_workgroups=int(rng.choice([0, 8, 40]))))):
            flat[64:-64].fill_(float("nan"))
            run()
            torch.cuda.synchronize()
            assert float(flat[:64].min()) == 5.0 == float(flat[-64:].max()), (name, trial)
            for (A, Bm, out), ref, (M, N, K) in zip(probs, refs, sizes):
                err = float((out - ref).abs().max())
                assert err <= 3e-5 * K ** 0.5 * 8 + 1e-4, (name, trial, (M, N, K), err)
        first = flat.clone()
        ops.gemm_bf16_tn_group(probs)
        assert torch.equal(flat, first), trial


@pytest.mark.parametrize("kind,H,B,T", [("gru", 768, 32, 9), ("gru", 1024, 64, 7), ("lstm", 512, 32, 8), ("gru", 256, 16, 12), ("lstm", 1280, 32, 4)])
def test_split_forward_recurrence_vs_fp64_and_fp32_kernels(kind, H, B, T):
    """fp32 mode, DS2_F32_RNN=split: the persistent forward recurrence with h_t and W_hh as hi + lo bf16 planes (three bf16 MFMAs per product)
    against the fp64 recurrence (oracle.gru_direction / lstm_direction, blocks.py:87-89) and against the fp32-MFMA kernels on ragged lengths:
    h, the saved gates and aux within 2e-5 of fp64 (the fp32 kernels: ~1e-6; the plain bf16 kernel: 3e-3), zeros beyond every length,
    reruns bit-identical.  LSTM H = 1280 at B = 32 does not fit the 16-unit split kernel (registers): the library takes the 10-unit-slice
    split kernel (csrc/rnn_fwd_u10.h, rnn_last_path bit 8) by itself."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import det
    from oracle import ds2_oracle as O
    from asr_amd import ops
    G = 3 if kind == "gru" else 4
    lens = sorted([int(v) for v in det.randint((B,), 51, max(1, T // 3), T + 1)], reverse=True)
    lens[0] = T
    lens_t = torch.tensor(lens, dtype=torch.int32)
    k = 1.0 / H ** 0.5
    gx = torch.from_numpy(det.uniform((T, B, 2, G * H), 52, -1.5, 1.5)).double()
    whh = torch.from_numpy(det.uniform((2, G * H, H), 53, -k, k)).double()
    bhh = torch.from_numpy(det.uniform((2, G * H), 54, -k, k)).double()
    step = O.gru_direction if kind == "gru" else O.lstm_direction
    ref = torch.stack([step(gx[:, :, 0], whh[0], bhh[0], lens_t, False), step(gx[:, :, 1], whh[1], bhh[1], lens_t, True)], 2)   # (T,B,2,H)
    dev = torch.device("cuda:0")
    ld = lens_t.to(dev)
    outs = {}
    for mode in (2, 0, 1):
        wpf, _ = ops.rnn_pack(G, whh.float().to(dev), bf16=mode)
        gxd = gx.float().reshape(T * B, 2 * G * H).to(dev).clone()
        hb, aux = ops.rnn_fwd(G, gxd, wpf, bhh.float().to(dev), ld, T, B, H, bf16=mode)
        outs[mode] = (hb.view(T, B, 2, H).double().cpu(), gxd.cpu(), ops.rnn_last_path())
        if mode == 2:
            gx2 = gx.float().reshape(T * B, 2 * G * H).to(dev).clone()
            hb2, _ = ops.rnn_fwd(G, gx2, wpf, bhh.float().to(dev), ld, T, B, H, bf16=2)
            assert torch.equal(hb2, hb) and torch.equal(gx2, gxd), "reruns differ"
    ops.rnn_persistent_check()
    fits = True
    assert outs[2][2] & 32 and bool(outs[2][2] & 256) == (kind == "lstm" and H == 1280), outs[2][2]
    err = {m: float((outs[m][0] - ref).norm() / ref.norm()) for m in outs}
    print(f"{kind} H={H} B={B} T={T}: h vs fp64: split {err[2]:.2e}  fp32 kernels {err[0]:.2e}  bf16 kernels {err[1]:.2e}  (split kernel took the call: {fits})")
    assert err[2] < 2e-5 and err[0] < 2e-5 and err[1] > 10 * err[2]
    assert float((outs[2][1] - outs[0][1]).abs().max()) < 5e-5                      # saved gates against the fp32 kernels'
    tmask = torch.arange(T).view(T, 1) >= lens_t.view(1, B)
    assert float(outs[2][0][tmask].abs().max()) == 0.0


@pytest.mark.parametrize("kind,H,B,T", [("gru", 768, 32, 9), ("lstm", 512, 32, 8), ("gru", 256, 16, 12), ("gru", 1024, 64, 6), ("lstm", 1280, 32, 5),
                                        ("lstm", 1280, 20, 4)])
def test_split_backward_recurrence_vs_fp64_and_fp32_kernels(kind, H, B, T):
    """fp32 mode, DS2_F32_RNN=split, backward: the persistent (all-gather) backward recurrence with dGh_t and W_hh^T as hi + lo bf16 planes
    against autograd through the fp64 recurrence (oracle.gru_direction / lstm_direction) on ragged lengths: dGx within 2e-5 (fp32 kernels
    ~1e-6, bf16 kernels 3e-3); GRU H = 1024 and LSTM H = 1280 fit no persistent split kernel (registers): the one-launch-per-step kernels take
    the call IN SPLIT FORM (rnn_bwd_step_kernel<.., SP>: dGh as a hi and a lo plane of the ping-pong buffers, rnn_last_path bit 6 without
    bit 1); reruns bit-identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import det
    from oracle import ds2_oracle as O
    from asr_amd import ops
    G = 3 if kind == "gru" else 4
    lens = sorted([int(v) for v in det.randint((B,), 61, max(1, T // 3), T + 1)], reverse=True)
    lens[0] = T
    lens_t = torch.tensor(lens, dtype=torch.int32)
    k = 1.0 / H ** 0.5
    gx = torch.from_numpy(det.uniform((T, B, 2, G * H), 62, -1.5, 1.5)).double().requires_grad_(True)
    whh = torch.from_numpy(det.uniform((2, G * H, H), 63, -k, k)).double()
    bhh = torch.from_numpy(det.uniform((2, G * H), 64, -k, k)).double()
    step = O.gru_direction if kind == "gru" else O.lstm_direction
    y = step(gx[:, :, 0], whh[0], bhh[0], lens_t, False) + step(gx[:, :, 1], whh[1], bhh[1], lens_t, True)
    dy = torch.from_numpy(det.uniform((T, B, H), 65, -1.0, 1.0)).double()
    (y * dy).sum().backward()
    ref = gx.grad.reshape(T * B, 2 * G * H)
    dev = torch.device("cuda:0")
    ld, dyd = lens_t.to(dev), dy.float().reshape(T * B, H).to(dev)
    err, paths, first = {}, {}, None
    for mode in (2, 2, 0):
        wpf, wpb = ops.rnn_pack(G, whh.float().to(dev), bf16=mode)
        gxd = gx.detach().float().reshape(T * B, 2 * G * H).to(dev).clone()
        hb, aux = ops.rnn_fwd(G, gxd, wpf, bhh.float().to(dev), ld, T, B, H, bf16=mode)
        ops.rnn_bwd(G, dyd, gxd, aux, hb, wpb, ld, T, B, H, bf16=mode)
        paths[mode] = ops.rnn_last_path()
        got = gxd.double().cpu()
        if mode == 2 and first is None:
            first = gxd.clone()
        elif mode == 2:
            assert torch.equal(first, gxd), "reruns differ"
        err[mode] = float((got - ref).norm() / ref.norm())
    ops.rnn_persistent_check()
    fits = not ((kind == "gru" and H == 1024) or (kind == "lstm" and H == 1280))      # a PERSISTENT split kernel fits
    assert paths[2] & 64 and bool(paths[2] & 2) == fits, paths
    ksplit = bool(paths[2] & 4)                                   # the K-split split kernel where H is a multiple of 256 and the registers allow
    assert ksplit == (fits and H % 256 == 0), paths
    print(f"{kind} H={H} B={B} T={T}: dGx vs fp64: split {err[2]:.2e}  fp32 kernels {err[0]:.2e}  (split backward kernel took the call: {fits}, K-split form: {ksplit})")
    assert err[2] < 2e-5 and err[0] < 2e-5


def test_grouped_splitk_terms_of_one_product_are_summed():
    """Consecutive problems of ds2_gemm_bf16_tn_splitk_group that name the SAME output are terms of one product (the fp32 mode's
    hi.hi + hi.lo + lo.hi): the launch writes their sum, for any split factor, next to ordinary single-term products, bit-identically
    from run to run; a three-term split product reproduces the fp64 product of the fp32 operands to ~5e-6."""
    from asr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    K, M, N = 3000, 520, 264
    X = torch.randn(K, M, device="cuda", generator=g)
    Y = torch.randn(K, N, device="cuda", generator=g)
    xs, ys = ops.split_bf16(X, 2), ops.split_bf16(Y, 2)                      # [hi | lo], pad8(M) = 520, pad8(N) = 264
    x_hi, x_lo, y_hi, y_lo = xs[:, :M], xs[:, M:], ys[:, :N], ys[:, N:]
    A2 = torch.randn(K - 64, 256, device="cuda", generator=g).bfloat16()
    B2 = torch.randn(K - 64, 512, device="cuda", generator=g).bfloat16()
    ref = (X.double().t() @ Y.double()).cpu()
    ref2 = A2.float().t() @ B2.float()
    prev = None
    for sk in (1, 2, 3):
        out, out2 = torch.full((M, N), float("nan"), device="cuda"), torch.full((256, 512), float("nan"), device="cuda")
        used = ops.gemm_bf16_tn_splitk_group([(x_hi, y_hi, out), (x_hi, y_lo, out), (x_lo, y_hi, out), (A2, B2, out2)], splitk=sk)
        assert used == sk
        e = float((out.double().cpu() - ref).norm() / ref.norm())
        assert e < 2e-5, (sk, e)
        assert float((out2 - ref2).abs().max()) < 3e-2
        again = torch.empty_like(out)
        ops.gemm_bf16_tn_splitk_group([(x_hi, y_hi, again), (x_hi, y_lo, again), (x_lo, y_hi, again)], splitk=sk)
        assert torch.equal(again, out), sk
        prev = out


def _conv2_fp64(a, w, bias, lens):
    """MaskConv's conv2 (modules/blocks.py:36-53 semantic): Conv2d(32,32,(21,11),stride (2,1),pad (10,5)) + mask beyond each length, fp64"""
    y = torch.nn.functional.conv2d(a.double(), w.double(), bias.double(), stride=(2, 1), padding=(10, 5))
    T = y.size(3)
    keep = (torch.arange(T, device=y.device)[None, :] < lens[:, None].to(y.device)).view(-1, 1, 1, T)
    return y * keep


@pytest.mark.parametrize("B,D1,T", [(3, 81, 57), (2, 81, 200)])
def test_fp32_mode_conv2_split_products_vs_fp64(B, D1, T):
    """engine.F32_CONV = split: conv2 forward / input gradient / weight gradient as three bf16-operand launches each on (hi, lo) operands made by
    ops.bf16_residual + the bf16 mode's cast / pack kernels, summed by ops.sum3_ — against fp64 on the same fp32 tensors: <= 2e-5 of the result's
    norm (the fp32-MFMA kernels: ~1e-6; one bf16 product: ~3e-3), ragged lengths masked identically."""
    from asr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    a = (torch.rand(B, 32, D1, T, device="cuda", generator=g) * 20).contiguous()
    w = (torch.randn(32, 32, 21, 11, device="cuda", generator=g) * 0.02).contiguous()
    bias = torch.randn(32, device="cuda", generator=g)
    lens = torch.tensor([T, max(1, T // 2), max(1, T - 3)][:B], dtype=torch.int32, device="cuda")
    keep = (torch.arange(T, device="cuda")[None, :] < lens[:, None]).view(B, 1, 1, T)
    a = (a * keep).contiguous()
    r = ops.bf16_residual(a)
    assert torch.equal(r, a - a.bfloat16().float())
    ph, pl = ops.conv2_pack_bf16(w), ops.conv2_pack_bf16(ops.bf16_residual(w))
    an, al = ops.nhwc_bf16(a), ops.nhwc_bf16(r)
    zb = torch.zeros_like(bias)
    y = ops.conv2_fwd_bf16(an, ph[0], bias, lens)
    one = y.clone()
    ops.sum3_(y, ops.conv2_fwd_bf16(al, ph[0], zb, lens), ops.conv2_fwd_bf16(an, pl[0], zb, lens))
    ref = _conv2_fp64(a, w, bias, lens)
    rel = lambda u, v: ((u.double() - v).norm() / v.norm()).item()
    assert rel(y, ref) < 2e-5, rel(y, ref)
    assert rel(one, ref) > 10 * rel(y, ref)                 # the single bf16 product is what the split improves on
    f32 = ops.conv2_fwd(a, ops.conv_pack(torch.zeros(32, 1, 41, 11, device="cuda"), w)[1], bias, lens)
    assert rel(f32, ref) < 2e-5
    # the FORWARD's form (engine.F32_CONV_FWD = split6): three pieces per operand, six products — fp32-grade, like the fp32-input MFMA kernel
    r2 = ops.bf16_residual(r)
    assert torch.equal(a, a.bfloat16().float() + (r.bfloat16().float() + r2))          # h + m + (what l rounds) is exactly a
    wr = ops.bf16_residual(w)
    am, al2 = ops.nhwc_bf16(r), ops.nhwc_bf16(r2)
    wm, wl2 = ops.conv2_pack_bf16(wr)[0], ops.conv2_pack_bf16(ops.bf16_residual(wr))[0]
    y6 = ops.conv2_fwd_bf16(am, wm, zb, lens)
    ops.sum3_(y6, ops.conv2_fwd_bf16(an, wl2, zb, lens), ops.conv2_fwd_bf16(al2, ph[0], zb, lens))
    ops.sum3_(y6, ops.conv2_fwd_bf16(an, wm, zb, lens), ops.conv2_fwd_bf16(am, ph[0], zb, lens))
    y6 = ops.sum3_(ops.conv2_fwd_bf16(an, ph[0], bias, lens), y6)
    print(f"\nconv2 forward vs fp64: six-term split {rel(y6, ref):.2e}, three-term {rel(y, ref):.2e}, fp32-input MFMA {rel(f32, ref):.2e}, one bf16 product {rel(one, ref):.2e}")
    assert rel(y6, ref) < max(3 * rel(f32, ref), 3e-7), (rel(y6, ref), rel(f32, ref))
    # backward: dy masked like the BatchNorm backward leaves it
    dy = (torch.randn_like(ref.float()) * keep).contiguous()
    rd = ops.bf16_residual(dy)
    a64, w64 = a.double().requires_grad_(True), w.double().requires_grad_(True)
    (_conv2_fp64(a64, w64, bias, lens) * dy.double()).sum().backward()
    dn, dl = ops.nhwc_bf16(dy), ops.nhwc_bf16(rd)
    da = ops.conv2_dgrad_bf16(dn, ph[1], ph[2], D1)
    ops.sum3_(da, ops.conv2_dgrad_bf16(dl, ph[1], ph[2], D1), ops.conv2_dgrad_bf16(dn, pl[1], pl[2], D1))
    assert rel(da, a64.grad) < 2e-5, rel(da, a64.grad)
    ap, apl, dp, dpl = ops.padcast_bf16(a), ops.padcast_bf16(r), ops.padcast_bf16(dy), ops.padcast_bf16(rd)
    gw, gb, gc = (torch.empty_like(w) for _ in range(3))
    ops.conv2_wgrad_bf16(ap, dp, lens, gw, T)
    ops.conv2_wgrad_bf16(apl, dp, lens, gb, T)
    ops.conv2_wgrad_bf16(ap, dpl, lens, gc, T)
    ops.sum3_(gw, gb, gc)
    assert rel(gw, w64.grad) < 2e-5, rel(gw, w64.grad)


def test_fp32_mode_split_conv_step_matches_fp32_mfma_conv_step():
    """whole fp32-mode step, conv2's backward through the split products (default) vs through the fp32-input MFMA kernels (DS2_F32_CONV=f32):
    the forward is the same code (logits, loss and every gradient behind the conv stack bit-identical), the conv-stack gradients within 1e-4
    of the other mode's norm; reruns bit-identical.  (The forward stays fp32 on purpose — see engine.F32_CONV: a split forward moved these
    gradients by 2e-3 through flipped Hardtanh branches.)"""
    sys.path.insert(0, ROOT)
    import bench
    from test_gpu_model import make_model
    from asr_amd import engine, ops
    from asr_amd.trainers.deepspeech_trainer import _prep_targets_host
    cfg = dict(rnn="gru", hidden=256, layers=2, classes=29)
    B, tin = 6, 241

    def run(mode):
        old = engine.F32_CONV
        engine.F32_CONV = mode
        try:
            torch.manual_seed(11)
            model = make_model(cfg)
            model.precision = "fp32"
            x, targets, pct, tsz = bench.synthetic_batch(B, tin, cfg["classes"], 1)
            x = x.cuda()
            model._ensure_flat(x.device)
            with torch.no_grad():
                W = model._flat.tensors(model)
                Gr = model._flat.tensors(model, grads=True)
                out_sizes = model.get_seq_lens((pct * tin).int())
                t_h, off_h, tl_h, max_u = _prep_targets_host(targets, tsz)
                lens_dev, tg, off, tl = out_sizes.to(torch.int32).cuda(), t_h.cuda(), off_h.cuda(), tl_h.cuda()
                logits, ctx = engine.forward(W, model._cfg, x, lens_dev, training=True, save=True)
                nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B, want_grad=True)
                engine.backward(W, Gr, model._cfg, ctx, dlogits)
                torch.cuda.synchronize()
                return logits.clone(), float(nll.sum()), {k: v.clone() for k, v in Gr.items()}
        finally:
            engine.F32_CONV = old

    ls, ns, gs = run("split")
    ls2, ns2, gs2 = run("split")
    lf, nf, gf = run("f32")
    assert torch.equal(ls, ls2) and all(torch.equal(gs[k], gs2[k]) for k in gs)
    assert ns == nf and torch.equal(ls, lf)
    ds = {}
    for k in gf:
        if gf[k].dtype != torch.float32:
            continue
        if not k.startswith("conv."):
            assert torch.equal(gs[k], gf[k]), k
        elif gf[k].norm().item() > 0 and k not in ("conv.seq_module.0.bias", "conv.seq_module.3.bias"):
            ds[k] = ((gs[k] - gf[k]).norm() / gf[k].norm()).item()       # (full-length batch: the conv biases' gradients are analytically zero)
    print("\nsplit-conv-backward vs fp32-MFMA-conv step, relative gradient differences:", {k: float("%.2g" % v) for k, v in ds.items()})
    assert len(ds) >= 4
    for k, d in ds.items():
        assert d < 1e-4, (k, d)


def test_weight_gradients_beside_a_recurrence_that_leaves_cus_idle_are_bit_identical():
    """DS2_WGRAD_IDLE: where a persistent backward recurrence leaves >= 64 CUs without a workgroup, the layer above's grouped split-K launch
    runs on the side stream beside it (bf16: behind the recurrence launch, when the recurrence holds at least half the chip; fp32 mode: the
    whole off-critical-path block, from the second step of a shape on).  Same kernels, same operands, same split factors: every gradient is
    bit-identical to the one-stream schedule, twice (a race would have two chances), and no persistent launch starves."""
    from asr_amd import engine, ops
    sys.path.insert(0, ROOT)
    import bench
    from test_gpu_model import make_model
    from asr_amd.trainers.deepspeech_trainer import _prep_targets_host
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def run(precision, cfg, B, tin, idle, steps):
        old = engine.WGRAD_IDLE
        engine.WGRAD_IDLE = idle
        engine._BWD_PERSISTENT.clear()
        try:
            torch.manual_seed(3)
            model = make_model(cfg)
            model.precision = precision
            x, targets, pct, tsz = bench.synthetic_batch(B, tin, cfg["classes"], 1)
            x = x.cuda()
            model._ensure_flat(x.device)
            out = []
            for _ in range(steps):
                with torch.no_grad():
                    W, Gr = model._flat.tensors(model), model._flat.tensors(model, grads=True)
                    out_sizes = model.get_seq_lens((pct * tin).int())
                    t_h, off_h, tl_h, max_u = _prep_targets_host(targets, tsz)
                    lens_dev, tg, off, tl = out_sizes.to(torch.int32).cuda(), t_h.cuda(), off_h.cuda(), tl_h.cuda()
                    logits, ctx = engine.forward(W, model._cfg, x, lens_dev, training=True, save=True)
                    nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B, want_grad=True)
                    engine.backward(W, Gr, model._cfg, ctx, dlogits)
                    torch.cuda.synchronize()
                    out.append((model.flat_parameters()[1].clone(), ops.rnn_last_path(), bool(getattr(ctx, "side_used", False))))
            ops.rnn_persistent_check()
            return out
        finally:
            engine.WGRAD_IDLE = old

    # bf16, 3 x 1024 GRU, B = 48: 2 x 3 x 32 = 192 workgroups in the recurrence, 64 CUs idle
    cfg = dict(rnn="gru", hidden=1024, layers=3, classes=29)
    if 64 <= engine._idle_cus_beside_bwd_recurrence(torch.device("cuda:0"), 48, 1024) <= cus // 2:
        on, on2, off = run("bf16", cfg, 48, 241, True, 1), run("bf16", cfg, 48, 241, True, 1), run("bf16", cfg, 48, 241, False, 1)
        assert on[0][1] & 6 == 6 and on[0][2] and not off[0][2], (on[0][1:], off[0][1:])       # K-split persistent backward; side stream used only when on
        assert torch.equal(on[0][0], off[0][0]) and torch.equal(on2[0][0], off[0][0])
    # fp32 mode, 2 x 768 GRU, B = 32 (c2's layer shape: 96 workgroups): the side stream from the second step on
    cfg = dict(rnn="gru", hidden=768, layers=2, classes=29)
    on, off = run("fp32", cfg, 32, 161, True, 3), run("fp32", cfg, 32, 161, False, 3)
    assert on[1][1] & 2, on[1][1]                                                              # persistent (split) backward recurrence
    for a, b in zip(on, off):
        assert torch.equal(a[0], b[0])


@pytest.mark.parametrize("kind,H,B,T", [("lstm", 160, 32, 9), ("gru", 320, 16, 7), ("lstm", 640, 20, 6), ("lstm", 1280, 32, 5), ("lstm", 1280, 13, 4),
                                        ("gru", 960, 32, 5)])
def test_u10_split_forward_recurrence_vs_fp64_and_fp32_kernels(kind, H, B, T):
    """csrc/rnn_fwd_u10.h: the split forward recurrence with TEN-unit hidden slices (G x 10 gate columns in 16-column MFMA tiles, h published
    dword by dword, operand gathered plane by plane) — what lets BASELINE C4's LSTM (H = 1280, B = 32) run as one launch per layer in the fp32
    mode.  Forced here on smaller shapes too (debug flag 256): h, the saved gates and the cell state / hn within 2e-5 of the fp64 recurrence
    (blocks.py:87-89) and 5e-5 of the fp32 kernels' values, zeros beyond every length, reruns bit-identical, no starved launch."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import det
    from oracle import ds2_oracle as O
    from asr_amd import ops, _lib
    G = 3 if kind == "gru" else 4
    lens = sorted([int(v) for v in det.randint((B,), 61, max(1, T // 3), T + 1)], reverse=True)
    lens[0] = T
    lens_t = torch.tensor(lens, dtype=torch.int32)
    k = 1.0 / H ** 0.5
    gx = torch.from_numpy(det.uniform((T, B, 2, G * H), 62, -1.5, 1.5)).double()
    whh = torch.from_numpy(det.uniform((2, G * H, H), 63, -k, k)).double()
    bhh = torch.from_numpy(det.uniform((2, G * H), 64, -k, k)).double()
    step = O.gru_direction if kind == "gru" else O.lstm_direction
    ref = torch.stack([step(gx[:, :, 0], whh[0], bhh[0], lens_t, False), step(gx[:, :, 1], whh[1], bhh[1], lens_t, True)], 2)   # (T,B,2,H)
    dev = torch.device("cuda:0")
    ld = lens_t.to(dev)
    wpf, _ = ops.rnn_pack(G, whh.float().to(dev), bf16=2)

    def run(flags):
        old = ops.debug_flags(flags)
        try:
            gxd = gx.float().reshape(T * B, 2 * G * H).to(dev).clone()
            hb, aux = ops.rnn_fwd(G, gxd, wpf, bhh.float().to(dev), ld, T, B, H, bf16=2)
            torch.cuda.synchronize()
            return hb.view(T, B, 2, H).clone(), gxd, aux.clone(), ops.rnn_last_path()
        finally:
            ops.debug_flags(old)

    u, u2 = run(256), run(256)
    assert u[3] & 256 and u[3] & 32 and u[3] & 1, u[3]                      # the 10-unit kernel took the call
    assert all(torch.equal(a, b) for a, b in zip(u[:3], u2[:3])), "reruns differ"
    wpf0, _ = ops.rnn_pack(G, whh.float().to(dev), bf16=0)
    g0 = gx.float().reshape(T * B, 2 * G * H).to(dev).clone()
    h0, a0 = ops.rnn_fwd(G, g0, wpf0, bhh.float().to(dev), ld, T, B, H, bf16=0)
    ops.rnn_persistent_check()
    e_u = float((u[0].double().cpu() - ref).norm() / ref.norm())
    e_0 = float((h0.view(T, B, 2, H).double().cpu() - ref).norm() / ref.norm())
    print(f"{kind} H={H} B={B} T={T}: h vs fp64: 10-unit split {e_u:.2e}  fp32 kernels {e_0:.2e}")
    assert e_u < 2e-5 and e_0 < 2e-5
    assert float((u[1] - g0).abs().max()) < 5e-5 and float((u[2] - a0).abs().max()) < 5e-5       # saved gates, cell state / hn
    tmask = torch.arange(T).view(T, 1) >= lens_t.view(1, B)
    assert float(u[0].cpu()[tmask].abs().max()) == 0.0
