"""GPU diagnostic: per-intermediate comparison of the conv stack fwd/bwd against the oracle."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from helpers import load_model_fixture, model_inputs, rel_l2
from test_gpu_model import make_model
from oracle import ds2_oracle as O
from asr_amd import engine, ops
from asr_amd.ctc import _prep_targets

name = sys.argv[1] if len(sys.argv) > 1 else "gru_h48_l3"
z, cfg = load_model_fixture(name)
sd, x, targets, pct, tsz = model_inputs(cfg)
B = x.size(0)
# ---- oracle with taps
params = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
taps = {}
lens_in = O.lengths_from_percentages(pct, x.size(3))
out, out_lens = O.forward(params, x.double(), lens_in, training=True, taps=taps)
lp = out.transpose(0, 1).log_softmax(2)
loss = O.ctc_loss_sum(lp, targets, out_lens, tsz) / B
names = ["conv1", "act1", "conv2", "act2"]
gt = torch.autograd.grad(loss, [taps[n] for n in names] + [params["conv.seq_module.0.weight"], params["conv.seq_module.3.weight"],
                                                          params["conv.seq_module.0.bias"], params["conv.seq_module.3.bias"]], retain_graph=True)
gtap = dict(zip(names, gt[:4]))
print("lens_in", lens_in.tolist(), "out_lens", out_lens.tolist(), "loss", float(loss))
# ---- HIP with captured intermediates
model = make_model(cfg, sd)
model._ensure_flat(torch.device("cuda:0"))
cap = {}
def wrap(fname):
    orig = getattr(ops, fname)
    def w(*a, **k):
        r = orig(*a, **k)
        cap.setdefault(fname, []).append((a, r))
        return r
    setattr(ops, fname, w)
for f in ["conv1_fwd", "conv2_fwd", "bn2d_act_fwd", "bn2d_act_bwd", "conv2_dgrad", "transpose_bft"]:
    wrap(f)
W = model._flat.tensors(model)
Gr = model._flat.tensors(model, grads=True)
lens_dev = out_lens.cuda()
with torch.no_grad():
    logits, ctx = engine.forward(W, model._cfg, x.cuda(), lens_dev, training=True, save=True)
    tg, off, tl, max_u = _prep_targets(targets, tsz, "cuda")
    nll, dlogits = ops.ctc_loss(logits, tg, off, lens_dev, tl, max_u, 1.0 / B)
    engine.backward(W, Gr, model._cfg, ctx, dlogits)
T = logits.shape[0]
mask = (torch.arange(T).view(1, 1, 1, T) < out_lens.view(B, 1, 1, 1))
def cmp(label, got, ref, m=None):
    got, ref = got.detach().cpu().double(), ref.detach().double()
    if m is not None:
        got, ref = got * m, ref * m
    print(f"{label:28s} rel_l2 {rel_l2(got.numpy(), ref.numpy()):.3e}   |ref| {float(ref.norm()):.4e}")
cmp("y1", cap["conv1_fwd"][0][1], taps["conv1"])
cmp("a1", cap["bn2d_act_fwd"][0][1], taps["act1"])
cmp("y2", cap["conv2_fwd"][0][1], taps["conv2"])
cmp("a2", cap["bn2d_act_fwd"][1][1], taps["act2"])
cmp("loss", (nll.sum() / B).reshape(1), loss.reshape(1))
# backward: transpose_bft calls: [0] fwd (to_tbf), [1] bwd -> da2
cmp("da2", cap["transpose_bft"][1][1].view(B, 32, -1, T), gtap["act2"])
cmp("dy2 (masked region)", cap["bn2d_act_bwd"][0][1], gtap["conv2"], mask)
cmp("da1", cap["conv2_dgrad"][0][1], gtap["act1"])
cmp("dy1 (masked region)", cap["bn2d_act_bwd"][1][1], gtap["conv1"], mask)
cmp("dW1", Gr["conv.seq_module.0.weight"], gt[4])
cmp("dW2", Gr["conv.seq_module.3.weight"], gt[5])
cmp("db1", Gr["conv.seq_module.0.bias"], gt[6])
cmp("db2", Gr["conv.seq_module.3.bias"], gt[7])
# isolate conv1 wgrad: feed the ORACLE's dy1 (masked) into the kernel
dy1_ref = (gtap["conv1"] * mask).float().cuda().contiguous()
dW1 = torch.empty(32, 1, 41, 11, device="cuda")
ops.conv1_wgrad(x.cuda(), dy1_ref, lens_dev, dW1)
cmp("dW1 from oracle dy1", dW1, gt[4])
ops.conv1_wgrad(x.cuda(), dy1_ref, torch.full_like(lens_dev, T), dW1)
cmp("dW1 from oracle dy1, no skip", dW1, gt[4])
for k in sorted(Gr):
    if k in params and params[k].requires_grad:
        g = torch.autograd.grad(loss, params[k], retain_graph=True)[0]
        print(f"  grad {k:45s} {rel_l2(Gr[k].cpu().double().numpy(), g.numpy()):.3e}")

# ---- localize the bn2d_act_bwd error per (b, t)
(a_args, dy2_got) = cap["bn2d_act_bwd"][0]
Y2, dA2 = a_args[0], a_args[1]
ref_dy2 = (gtap["conv2"] * mask).float()
err = (dy2_got.cpu() - ref_dy2).double()
per_bt = (err ** 2).sum((1, 2))            # (B, T)
refn = (ref_dy2.double() ** 2).sum((1, 2))
print("per-sample err^2:", per_bt.sum(1).tolist())
for b in range(B):
    bad = [(t, float(per_bt[b, t]), float(refn[b, t])) for t in range(T) if per_bt[b, t] > 1e-6 * max(float(refn[b, t]), 1e-12) and per_bt[b, t] > 1e-10]
    print("b", b, "len", int(out_lens[b]), "bad frames:", bad[:8], "... total", len(bad))
# recompute with torch on the same GPU tensors
m2, v2 = a_args[3], a_args[4]
ga, be = a_args[5], a_args[6]
xh = (Y2 - m2.view(1, -1, 1, 1)) * torch.rsqrt(v2.view(1, -1, 1, 1) + 1e-5)
zz = xh * ga.view(1, -1, 1, 1) + be.view(1, -1, 1, 1)
mk = mask.cuda()
dz = torch.where(mk & (zz > 0) & (zz < 20), dA2, torch.zeros_like(dA2))
print("dbeta: kernel", Gr["conv.seq_module.4.bias"][:4].tolist(), "torch", dz.sum((0, 2, 3))[:4].tolist(), "oracle",
      torch.autograd.grad(loss, params["conv.seq_module.4.bias"], retain_graph=True)[0][:4].tolist())
print("lens_dev", lens_dev.tolist(), lens_dev.dtype)

# ---- frame inspection
bb, tt = 3, 9
yf = Y2[bb, :, :, tt].cpu()
print("frame y2: max|y|", float(yf.abs().max()), "mean", float(yf.mean()), "global std", float(Y2.std()))
zf = zz[bb, :, :, tt].cpu()
print("frame z: >20:", int((zf >= 20).sum()), "<=0:", int((zf <= 0).sum()), "of", zf.numel(), " max z", float(zf.max()))
# oracle z for conv2 (recompute in fp64 from the oracle taps)
y2o = taps["conv2"].detach()
mu_o = y2o.mean((0, 2, 3)); var_o = ((y2o - mu_o.view(1, -1, 1, 1)) ** 2).mean((0, 2, 3))
z_o = (y2o - mu_o.view(1, -1, 1, 1)) * torch.rsqrt(var_o.view(1, -1, 1, 1) + 1e-5) * params["conv.seq_module.4.weight"].detach().view(1, -1, 1, 1) + params["conv.seq_module.4.bias"].detach().view(1, -1, 1, 1)
pass_o = (z_o > 0) & (z_o < 20) & mask
pass_k = ((zz > 0) & (zz < 20)).cpu() & mask
diff = (pass_o != pass_k)
print("pass-mask differences total:", int(diff.sum()), " in frame:", int(diff[bb, :, :, tt].sum()))
idx = diff.nonzero()[:10]
for i in idx:
    b_, c_, d_, t_ = [int(v) for v in i]
    print("  diff at", (b_, c_, d_, t_), "z_oracle", float(z_o[b_, c_, d_, t_]), "z_kernel", float(zz[b_, c_, d_, t_]), "y", float(Y2[b_, c_, d_, t_]), "dA", float(dA2[b_, c_, d_, t_]))
print("var kernel vs oracle (first 4):", v2[:4].tolist(), var_o[:4].tolist())
print("max rel var err:", float(((v2.cpu().double() - var_o) / var_o).abs().max()), " max abs mean err:", float((m2.cpu().double() - mu_o).abs().max()))
e = (dy2_got.cpu() - ref_dy2)[bb, :, :, tt]
print("frame err by channel (abs max):", [round(float(v), 5) for v in e.abs().amax(1)][:32])
print("frame ref by channel (abs max):", [round(float(v), 5) for v in ref_dy2[bb, :, :, tt].abs().amax(1)][:32])
