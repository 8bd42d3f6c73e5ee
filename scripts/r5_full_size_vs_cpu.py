"""The metric configuration at its FULL size — 5 x 1024 BiGRU, batch 64, T_in = 1001 (ragged: 64 lengths between 7 and 10 s), 29 classes —
directly against the CPU oracle (oracle/ds2_packed.py, the reference's statement sequence in its packed-sequence form, pinned against the
reference goldens): logits, loss and every parameter gradient of the HIP path in fp32 mode and in the config's bf16 mode.  The test suite
holds this comparison at B <= 16 / T_in <= 501 for run time (tests/test_gpu_configs.py); this script is the same code at full size, its output
is committed under profiles/.   python scripts/r5_full_size_vs_cpu.py [B]      (CPU part: a few minutes on 16 threads)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from helpers import model_inputs, rel_l2, hardtanh_flip_fraction
from test_gpu_model import make_model
from oracle import ds2_oracle as O, ds2_packed as P
from asr_amd import CTCLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
t_ins = sorted((int(v) for v in rng.integers(701, 1002, size=B)), reverse=True)
t_ins[0] = 1001
cfg = dict(rnn="gru", hidden=1024, layers=5, classes=29, t_ins=t_ins)
sd, x, targets, pct, tsz = model_inputs(cfg, well_conditioned=False)
torch.set_num_threads(min(16, os.cpu_count() or 1))
t0 = time.time()
params = P.leaf_params(sd)
out_ref, _, loss_ref = P.fit(params, x, targets, pct.clone(), tsz)
loss_ref.backward()
gref = {k: v.grad.detach().numpy().astype(np.float64) for k, v in params.items() if v.requires_grad}
print(f"CPU oracle (packed form, {torch.get_num_threads()} threads): B={B} T_in={x.size(3)} loss {float(loss_ref):.6f}  {time.time() - t0:.0f} s", flush=True)
gmax = max(float(np.linalg.norm(g)) for g in gref.values())
lens = O.lengths_from_percentages(pct, x.size(3))
for precision in ("fp32", "bf16"):
    model = make_model(cfg, sd)
    model.precision = precision
    out, out_lens = model.forward(x.cuda(), lens)
    loss = CTCLoss(reduction="sum")(out.transpose(0, 1), targets, out_lens, tsz) / B
    loss.backward()
    e_logits = max(rel_l2(out[b, :int(out_lens[b])].detach().cpu().numpy(), out_ref[b, :int(out_lens[b])].detach().numpy()) for b in range(B))
    e_loss = abs(float(loss.detach()) - float(loss_ref.detach())) / float(loss_ref.detach())
    grads = {k: p.grad.cpu().numpy().astype(np.float64) for k, p in model.named_parameters()}
    errs = {k: np.linalg.norm(g - gref[k]) / max(np.linalg.norm(gref[k]), 1e-4 * gmax, 1e-12) for k, g in grads.items()}
    worst_rnn = max(((k, e) for k, e in errs.items() if not k.startswith("conv.")), key=lambda kv: kv[1])
    worst_conv = max(((k, e) for k, e in errs.items() if k.startswith("conv.") and not k.endswith(".bias")), key=lambda kv: kv[1])
    # conv biases sit in front of a BatchNorm: their gradient is a near-cancelling sum (exactly zero for an un-padded batch) — reported apart, with
    # its size relative to the largest gradient
    cb = {k: (errs[k], float(np.linalg.norm(gref[k])) / gmax) for k in errs if k.startswith("conv.") and k.endswith(".bias") and "seq_module.0" in k or k == "conv.seq_module.3.bias"}
    line = (f"{precision}: loss {float(loss.detach()):.6f} (rel {e_loss:.2e})  logits worst utterance {e_logits:.2e}  worst rnn/fc gradient {worst_rnn[0]} {worst_rnn[1]:.2e}  "
            f"worst conv / BatchNorm2d weight gradient {worst_conv[0]} {worst_conv[1]:.2e}  conv biases (err, |g| / |g|max): " +
            ", ".join(f"{k.split('.')[2]}: {e:.2e}, {n:.1e}" for k, (e, n) in sorted(cb.items())))
    if precision == "bf16":
        flips = hardtanh_flip_fraction(model, x, pct)
        line += f"  (Hardtanh flips {flips:.2e}: counted bound 3 sqrt(f) = {3 * flips ** 0.5:.2e})"
    print(line, flush=True)
    del model
