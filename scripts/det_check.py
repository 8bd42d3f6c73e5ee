"""Run-to-run determinism of one fused train step at the bench configuration (c3 bf16): the same state and batch, N times; reports
which parameter gradients differ between runs (a difference = a race or an uninitialised read somewhere in backward)."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from asr_amd import CTCLoss, DeepSpeech, FusedAdamW
from asr_amd.trainers import DeepSpeechTrainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
ref = None
bad = 0
for it in range(N):
    model.load_state_dict(sd0)
    opt = FusedAdamW(model, lr=3e-4)
    tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
    _, lv = tr.step((x, targets, pct.clone(), tsz))
    Gr = {k: v.clone() for k, v in model._flat.tensors(model, grads=True).items()}
    if ref is None:
        ref = (lv, Gr)
        continue
    diffs = [(k, float((Gr[k] - ref[1][k]).abs().max())) for k in Gr if not torch.equal(Gr[k], ref[1][k])]
    if lv != ref[0] or diffs:
        bad += 1
        print(f"run {it}: loss {lv} vs {ref[0]}; differing grads: {diffs[:12]}", flush=True)
print(f"{bad} of {N - 1} reruns differ", flush=True)
