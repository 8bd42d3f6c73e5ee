#!/usr/bin/env python
"""bench.py — DeepSpeech2 CTC train-step throughput (utterances/sec) on MI355X, BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = one full train step of the hot path on one synthetic batch already resident in HBM:
forward, CTC, backward, (RCCL gradient all-reduce,) fused AdamW, loss.item() sync.
Workloads (--workload): c3 = BASELINE metric config (5x1024 BiGRU, 10 s / 161-bin, B=64 per GPU, C=29),
c2 = configs[1] (5x768 BiGRU, B=32).  Weak scaling: per-GPU batch fixed.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rnn, hidden, layers, classes, per-GPU batch, input frames)
    "c3": ("gru", 1024, 5, 29, 64, 1001),
    "c2": ("gru", 768, 5, 29, 32, 1001),
    "c1": ("gru", 256, 2, 29, 4, 201),
    "c4": ("lstm", 1280, 7, 29, 32, 1501),
    "c5": ("gru", 1024, 5, 80, 64, 2001),       # ragged 3-20 s (T_b ~ U{301..2001}), length-sorted, ~80 kana classes
}
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input matrix rate
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA (the 5 PF marketing figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0


def train_flops_per_utt(rnn, H, L, C, T):
    """FLOPs(train) = 2*conv1 + 3*(conv2 + rnn + fc)."""
    G = 3 if rnn == "gru" else 4
    conv1 = 2 * 32 * 81 * T * 451
    conv2 = 2 * 32 * 41 * T * 7392
    rnn_f = sum(2 * 2 * T * G * H * ((1312 if l == 0 else H) + H) for l in range(L))
    fc = 2 * T * H * C
    return 2 * conv1 + 3 * (conv2 + rnn_f + fc)


def audio_conf():
    return SimpleNamespace(sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming", speed_volume_perturb=False,
                           spec_augment=False, noise_dir=None, noise_prob=0.4, noise_levels=(0.0, 0.5))


def label_file(tmp, n):
    import pandas as pd
    chars = ["_", "'"] + list("abcdefghijklmnopqrstuvwxyz") + ["|"] + [chr(0x3041 + i) for i in range(200)]
    path = os.path.join(tmp, "labels.csv")
    pd.DataFrame({"label": chars[:n]}).to_csv(path, index=False)
    return path


def synthetic_batch(B, tin, C, seed, ragged=False):
    """Collated batch in the reference's contract (functional.py:9-32).  ragged: T_b ~ U{301..tin}, sorted descending."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, 161, tin, generator=g)
    if ragged:
        tb = torch.randint(301, tin + 1, (B,), generator=g).sort(descending=True).values
        tb[0] = tin
    else:
        tb = torch.full((B,), tin)
    for i in range(B):
        x[i, :, :, int(tb[i]):] = 0.0
    pct = torch.tensor([int(t) / float(tin) for t in tb], dtype=torch.float32)
    tsz = (tb // 20).to(torch.int32)
    targets = torch.randint(1, C, (int(tsz.sum()),), generator=torch.Generator().manual_seed(seed + 1), dtype=torch.int32)
    return x, targets, pct, tsz


def _oracle_state(rnn, H, L, C):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import det
    shapes = det.state_shapes(rnn, H, L, C)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var") or (k.endswith("weight") and len(shp) == 1):
            sd[k] = torch.ones(shp)
        elif k.endswith("running_mean") or (k.endswith("bias") and ("batch_norm" in k or k.startswith("fc.") or ".1." in k or ".4." in k)):
            sd[k] = torch.zeros(shp)
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * (1.0 / (H ** 0.5))
    return sd


def cpu_baseline_worker(rnn, H, L, C, tin):
    """Runs in a child process (hard wall-clock limit enforced by the parent).  Times the CPU oracle
    (oracle/ds2_oracle.py: padded+masked restatement of fit + backward) on B=1 of the same model and
    utterance length.  The thread count is probed first: on many-core hosts the op mix of this model
    (thousands of tiny per-frame ops) is SLOWER with all cores than with 8-32 threads."""
    from oracle import ds2_oracle as O
    cores = os.cpu_count() or 1
    sd = _oracle_state(rnn, H, L, C)

    def run(t_in, reps=1):
        x, targets, pct, tsz = synthetic_batch(1, t_in, C, 1)
        best = 1e30
        for _ in range(reps):
            t0 = time.time()
            O.fit_and_grads(sd, x, targets, pct, tsz)
            best = min(best, time.time() - t0)
        return best

    cands = sorted({min(cores, 8), min(cores, 16)})   # larger teams only get slower on this op mix (measured)
    probe = {}
    for n in cands:
        torch.set_num_threads(n)
        run(41)                       # warm-up (thread pool, allocator)
        probe[n] = run(41)
    nthr = min(probe, key=probe.get)
    torch.set_num_threads(nthr)
    t_small = run(201)
    est_full = t_small * tin / 201.0
    if est_full <= 45.0:
        dt, sample_t = run(tin), tin
    else:
        dt, sample_t = est_full, 201
    out = {"value": 1.0 / dt, "unit": "utterances/sec", "cores": nthr, "kind": "port", "host_cores": cores,
           "sample": (f"CPU oracle fit+backward, B=1 of the same {L}x{H} {rnn} model, T_in={sample_t}"
                      + ("" if sample_t == tin else f" scaled linearly to T_in={tin}")
                      + f"; threads probed {probe} -> {nthr}")}
    print("CPU_BASELINE_JSON " + json.dumps(out), flush=True)


def cpu_baseline(rnn, H, L, C, tin, limit_s=240):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", rnn, str(H), str(L), str(C), str(tin)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="16", MKL_NUM_THREADS="16"))
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                return json.loads(line[len("CPU_BASELINE_JSON "):])
        return {"value": None, "unit": "utterances/sec", "cores": 0, "kind": "port", "sample": "worker failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "utterances/sec", "cores": 0, "kind": "port", "sample": f"worker exceeded {limit_s}s"}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
        rnn, H, L, C, tin = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        return cpu_baseline_worker(rnn, H, L, C, tin)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--dtype", default="", choices=["", "f32", "bf16"], help="default: the config dtype (bf16 for c3 and c5, f32 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-section time breakdown to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = args.gpus > 1 or world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from asr_amd import CTCLoss, DeepSpeech, FusedAdamW, ops
    from asr_amd.trainers import DeepSpeechTrainer

    rnn, H, L, C, B, tin = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        model = DeepSpeech(audio_conf=audio_conf(), decoder=None, label_path=label_file(tmp, C), rnn_type=rnn, rnn_hidden_size=H,
                           rnn_hidden_layers=L, bidirectional=True)
    model.to(dev).train()
    dtype = args.dtype or ("bf16" if args.workload in ("c3", "c5") else "f32")
    model.precision = "bf16" if dtype == "bf16" else "fp32"
    opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
    x, targets, pct, tsz = synthetic_batch(B, tin, C, 1 + rank, ragged=(args.workload == "c5"))
    x = x.to(dev)                                     # inputs resident in HBM before the timed region

    def one_step():
        return tr.step((x, targets, pct.clone(), tsz))

    for _ in range(args.warmup):
        valid, lv = one_step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        valid, lv = one_step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    utts = world * B * args.steps / dt

    # ---- roofline of the dominant kernel: the recurrent step kernel (fwd), timed live with HIP events on
    # torch's current stream, which is the stream libds2hip launches on.
    G = 3 if rnn == "gru" else 4
    T = (tin + 1) // 2
    M = T * B
    gx = torch.randn(M, 2 * G * H, device=dev) * 0.5
    whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
    bhh = torch.zeros(2, G * H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    bf = dtype == "bf16"
    wpf, wpb_probe = ops.rnn_pack(G, whh, bf16=bf)
    pack = bf and B % 8 == 0                              # the train step's own mode (engine.forward)
    ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=bf, packed_gates=pack)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    gx2 = gx.clone()
    dyp = torch.randn(M, H, device=dev)
    side = torch.empty(M, 2 * G * H, dtype=torch.bfloat16, device=dev) if pack else None
    torch.cuda.synchronize()
    e0.record()
    fwd_out = ops.rnn_fwd(G, gx2, wpf, bhh, lens, T, B, H, bf16=bf, packed_gates=pack)
    e1.record()
    # the same layer's backward recurrence on the state just saved, in the train step's own mode
    if pack:
        ops.rnn_bwd(G, dyp, None, fwd_out[1], fwd_out[0], wpb_probe, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=fwd_out[2])
    else:
        ops.rnn_bwd(G, dyp, gx2, fwd_out[1], fwd_out[0], wpb_probe, lens, T, B, H, bf16=bf)
    e2.record()
    torch.cuda.synchronize()
    bwd_layer_us = e1.elapsed_time(e2) * 1e3
    from asr_amd import _lib as _ds2lib
    path_bits = _ds2lib.load().ds2_rnn_last_path()
    layer_us = e0.elapsed_time(e1) * 1e3                 # one layer's whole forward recurrence (T time steps, both directions)
    # bf16 mode runs the recurrence as ONE persistent launch per layer (rnn_fwd_persistent_kernel) when every workgroup can be resident at
    # once (grid <= CU count) — what ds2_rnn_fwd decides; otherwise (and in fp32) it is one rnn_fwd_step_kernel launch per time step.
    persistent = bool(path_bits & 1)                     # what ds2_rnn_fwd actually did (ds2_rnn_last_path)
    launches = 1 if persistent else T
    us_per_launch = layer_us / launches
    flops_per_launch = 2.0 * 2 * B * H * G * H * (T if persistent else 1)      # both directions; all T steps in the persistent launch
    achieved = flops_per_launch / (us_per_launch * 1e-6) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS if bf else FP32_MFMA_PEAK_TFLOPS
    # HBM-side bytes from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), measured for exactly this shape and mode
    # (c3, bf16 operands, packed gate records): 4.48 MB per time step in the persistent kernel (profiles/r01_pmc_persistent/), 20.84 MB
    # per launch of the step kernel (profiles/r01_pmc/); null for other shapes.
    traffic = None
    if args.workload == "c3" and bf and B == 64:
        traffic = 4.48e6 * T if persistent else 20.84e6
    alg_bytes_step = (3 * 4 + 8 + 4 + 2 if pack else 8 * 4 + (2 if bf else 4)) * B * 2 * H
    roofline = {"kernel": "rnn_fwd_persistent_kernel" if persistent else "rnn_fwd_step_kernel", "bound": "mfma", "achieved": achieved,
                "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "algorithmic_hbm_bytes_per_launch": alg_bytes_step * (T if persistent else 1),
                "us_per_launch": us_per_launch, "us_per_time_step": layer_us / T, "launches_per_step": (1 if persistent else T) * L}
    # the backward recurrence (same FLOPs per time step): on a single GPU it is persistent too wherever its W_hh^T slice fits the
    # registers.  Whichever of the two kernels takes longer per train step is THE dominant kernel and goes first.
    bwd_persistent = bool(path_bits & 2)
    bl = 1 if bwd_persistent else T
    bl_steps = T if bwd_persistent else 1
    b_ach = flops_per_launch / (T if persistent else 1) * (T if bwd_persistent else 1) / (bwd_layer_us / bl * 1e-6) / 1e12
    bwd_traffic = None
    if args.workload == "c3" and bf and B == 64 and bwd_persistent:
        bwd_traffic = 6.71e6 * T                           # profiles/r01_pmc_persistent/: FETCH 1650 KB x2 + WRITE 3256 KB per time step
    roofline_bwd = {"kernel": "rnn_bwd_persistent_kernel" if bwd_persistent else "rnn_bwd_step_kernel", "bound": "mfma", "achieved": b_ach,
                    "peak": peak, "unit": "TFLOP/s", "frac": b_ach / peak, "traffic": bwd_traffic,
                    # gate record 8 + previous state 4 + dGx 3 x 2 + d(hn) 4 bytes per hidden unit and direction, dy 4 bytes per unit (packed mode)
                    "algorithmic_hbm_bytes_per_launch": ((8 + 4 + 2 * G + 4) * 2 + 4 if pack else (4 * G + 4 + 4 + 4 * G + 4) * 2 + 4) * B * H * bl_steps,
                    "us_per_launch": bwd_layer_us / bl,
                    "us_per_time_step": bwd_layer_us / T, "launches_per_step": bl * L}
    if bwd_layer_us > layer_us:
        roofline, roofline_bwd = roofline_bwd, roofline
    roofline["second_kernel"] = roofline_bwd

    if args.breakdown and rank == 0:
        breakdown(model, tr, x, targets, pct, tsz)

    if rank == 0:
        step_flops = sum(train_flops_per_utt(rnn, H, L, C, (int(round(float(p) * tin)) + 1) // 2) for p in pct)
        out = {
            "metric": "utterances/sec (10 s, 161-bin) DS2 5x1024 BiGRU CTC train step" if args.workload == "c3"
                      else f"utterances/sec DS2 {L}x{H} bi-{rnn} CTC train step",
            "value": utts, "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic N(0,1) 161-bin spectrograms, random-init weights, random labels U=T_in/20",
            "config": {"workload": f"{args.workload}: DS2 {L}x{H} bi-{rnn.upper()} {dtype}, {tin} input frames ({tin // 100} s), "
                                   f"batch {B}/GPU, {C} classes", "global_batch": B * world, "parallelism": f"dp{world}"},
            "loss": lv, "step_tflops": step_flops * world / (ms * 1e-3) / 1e12,
            "step_frac_of_fp32_mfma_peak": step_flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "step_frac_of_bf16_mfma_peak": step_flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rnn, H, L, C, tin)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def breakdown(model, tr, x, targets, pct, tsz):
    """Section timing with events (one extra step, outside the timed region)."""
    from asr_amd import engine, ops
    from asr_amd.ctc import _prep_targets
    dev = x.device
    B = x.size(0)
    ev = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((name, e))

    input_sizes = pct.clone().mul_(int(x.size(3))).int()
    out_lens = model.get_seq_lens(input_sizes)
    lens_dev = out_lens.to(dev)
    tg, off, tl, max_u = _prep_targets(targets, tsz, dev)
    W = model._flat.tensors(model)
    Gr = model._flat.tensors(model, grads=True)
    orig_rnn_fwd, orig_rnn_bwd, orig_gemm = ops.rnn_fwd, ops.rnn_bwd, ops.gemm_raw
    acc = {"rnn_fwd": 0.0, "rnn_bwd": 0.0, "gemm": 0.0}

    def timed(fn, key):
        def w(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            pend.append((key, s, e))
            return r
        return w
    pend = []
    ops.rnn_fwd, ops.rnn_bwd, ops.gemm_raw = timed(orig_rnn_fwd, "rnn_fwd"), timed(orig_rnn_bwd, "rnn_bwd"), timed(orig_gemm, "gemm")
    try:
        with torch.no_grad():
            torch.cuda.synchronize()
            mark("start")
            logits, ctx = engine.forward(W, model._cfg, x, lens_dev, training=True, save=True)
            mark("forward")
            nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B)
            mark("ctc")
            engine.backward(W, Gr, model._cfg, ctx, dlogits)
            mark("backward")
            tr._optimizer.step()
            mark("adamw")
            torch.cuda.synchronize()
    finally:
        ops.rnn_fwd, ops.rnn_bwd, ops.gemm_raw = orig_rnn_fwd, orig_rnn_bwd, orig_gemm
    for key, s, e in pend:
        acc[key] += s.elapsed_time(e)
    msg = ["breakdown (ms):"]
    for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
        msg.append(f"  {n1}: {e0.elapsed_time(e1):.2f}")
    msg.append("  of which " + ", ".join(f"{k}={v:.2f}" for k, v in acc.items()))
    print("\n".join(msg), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
