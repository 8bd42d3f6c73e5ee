"""Run-to-run determinism of engine.forward at the bench configuration (c3 bf16): reports the FIRST saved tensor that differs."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import bench
from asr_amd import DeepSpeech, engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
WL = os.environ.get("WL", "c3")
rnn, H, L, C, B, tin = bench.WORKLOADS[WL]
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    model = DeepSpeech(audio_conf=bench.audio_conf(), decoder=None, label_path=bench.label_file(tmp, C), rnn_type=rnn, rnn_hidden_size=H,
                       rnn_hidden_layers=L, bidirectional=True)
model.to(dev).train()
model.precision = os.environ.get("PREC", "bf16")
x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1, ragged=True)
x = x.to(dev)
model._ensure_flat(dev)
W = model._flat.tensors(model)
T = (tin + 1) // 2
out_lens = torch.tensor([max(1, (int(round(float(p) * tin)) + 1) // 2) for p in pct], dtype=torch.int32, device=dev)
def snap():
    with torch.no_grad():
        logits, ctx = engine.forward(W, model._cfg, x, out_lens, training=True, save=True)
    d = {k: v for k, v in (("y1", ctx.y1), ("a1", ctx.a1), ("a1p", ctx.a1p), ("y2", ctx.y2)) if v is not None}
    for l, lc in enumerate(ctx.layers):
        for nm in ("xin", "xn", "gx", "rec", "hbuf"):
            t = getattr(lc, nm, None)
            if t is not None: d[f"L{l}.{nm}"] = t
    d["logits"] = logits
    return {k: v.clone() for k, v in d.items()}
ref = snap()
bad = 0
for it in range(N):
    cur = snap()
    for k in ref:
        if not torch.equal(ref[k].view(torch.uint8) if ref[k].dtype == torch.bfloat16 else ref[k], cur[k].view(torch.uint8) if cur[k].dtype == torch.bfloat16 else cur[k]):
            a, b = ref[k].float(), cur[k].float()
            nd = int((a != b).sum())
            idx = (a != b).nonzero()[:3].tolist()
            print(f"run {it}: first differing tensor {k} shape {tuple(a.shape)}: {nd} elements differ, max abs {float((a - b).abs().max()):.3e}, at {idx}", flush=True)
            bad += 1
            break
print(f"{bad} of {N} reruns differ", flush=True)
