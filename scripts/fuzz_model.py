"""Random-shape sweep of the whole train step (fp32 and bf16) against the fp64 oracle: catches shape-dependent bugs the fixed test
cases do not cover (odd batch sizes, H not a multiple of 16/32, single-frame rows, GRU and LSTM).  Not part of pytest (minutes)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import det
from oracle import ds2_oracle as O
from test_gpu_model import make_model
from helpers import model_inputs, rel_l2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    rnn = ["gru", "lstm"][rng.randint(2)]
    H = int(rng.choice([8, 12, 20, 36, 44, 64, 100]))
    L = int(rng.randint(1, 4))
    B = int(rng.randint(1, 19))
    if len(sys.argv) > 3 and sys.argv[3] == "b8":      # exercise the bf16 training path's packed records / bf16 dGx (needs B % 8 == 0)
        B = int(rng.choice([8, 16, 24, 40]))
    tmax = int(rng.randint(2, 70))
    t_ins = sorted([int(v) for v in rng.randint(1, tmax + 1, size=B)], reverse=True)
    t_ins[0] = tmax
    if B * ((tmax + 1) // 2) < 8:
        # BatchNorm1d over fewer than 8 rows is ill-conditioned (two rows: x_hat = +-1 exactly, every upstream difference is amplified by
        # 1/sigma): the comparison then measures conditioning, not the kernels (seen: 1.5e-3 on a 2-row case, 7e-6 typical)
        print(f"[{it}] skip (T*B < 8 rows)"); continue
    cfg = dict(rnn=rnn, hidden=H, layers=L, classes=int(rng.choice([5, 29])), t_ins=t_ins)
    sd, x, targets, pct, tsz = model_inputs(cfg, well_conditioned=False)
    lens = O.lengths_from_percentages(pct, x.size(3))
    out_lens = O.seq_lens_after_conv(lens)
    tsz = torch.minimum(tsz, out_lens.to(tsz.dtype)).clamp(min=1)
    targets = torch.cat([torch.full((int(k),), 1 + (i % (cfg["classes"] - 1)), dtype=targets.dtype) for i, k in enumerate(tsz.tolist())])
    ref = O.fit_and_grads(sd, x, targets, pct, tsz, dtype=torch.float64)
    if not np.isfinite(ref["loss"]):
        print(f"[{it}] skip (infeasible alignment)"); continue
    gmax = max(float(np.linalg.norm(v.numpy())) for v in ref["grads"].values())
    for prec in ("fp32", "bf16"):
        model = make_model(cfg, sd); model.precision = prec
        out, ol = model.forward(x.cuda(), lens)
        from asr_amd import CTCLoss
        loss = CTCLoss(reduction="sum")(out.transpose(0, 1), targets, ol, tsz) / B
        loss.backward()
        tol_o, tol_g = (1e-3, 1e-3) if prec == "fp32" else (3e-2, 2e-1)
        worst = 0.0
        for k, p in model.named_parameters():
            g = ref["grads"][k].numpy()
            e = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - g) / max(np.linalg.norm(g), 1e-3 * gmax)
            worst = max(worst, e)
        el = abs(float(loss.detach()) - ref["loss"]) / abs(ref["loss"])
        eo = rel_l2(out.detach().cpu().numpy(), ref["logits"].numpy())
        ok = el <= tol_o and eo <= tol_o and worst <= tol_g and np.isfinite(worst)
        note = ""
        if not ok and eo <= tol_o and el <= tol_o:
            # forward agrees, gradients do not: is a BatchNorm2d output sitting on a Hardtanh kink (DESIGN.md §2 numerics note)?
            margin = O.hardtanh_kink_margin(sd, x, lens)
            if margin < 4e-6:
                ok, note = True, f" (kink margin {margin:.1e}: gradient is ill-conditioned, not counted)"
        bad += (not ok)
        print(f"[{it}] {rnn} H={H} L={L} B={B} tmax={tmax} C={cfg['classes']} {prec}: loss {el:.1e} logits {eo:.1e} worst grad {worst:.1e} {'ok' if ok else 'FAIL'}{note}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
