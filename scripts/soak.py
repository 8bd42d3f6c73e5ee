"""Soak: N fused train steps of the bench configuration on one synthetic batch; loss must fall and stay finite, HBM use must not grow."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from asr_amd import CTCLoss, DeepSpeech, FusedAdamW
from asr_amd.trainers import DeepSpeechTrainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
rnn, H, L, C, B, tin = bench.WORKLOADS[os.environ.get("WL", "c3")]
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    model = DeepSpeech(audio_conf=bench.audio_conf(), decoder=None, label_path=bench.label_file(tmp, C), rnn_type=rnn, rnn_hidden_size=H,
                       rnn_hidden_layers=L, bidirectional=True)
model.to(dev).train()
model.precision = os.environ.get("PREC", "bf16")
opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1, ragged=True)
x = x.to(dev)
t0 = time.time()
for i in range(N):
    valid, lv = tr.step((x, targets, pct.clone(), tsz))
    if i % 25 == 0 or i == N - 1:
        print(f"step {i:4d} loss {lv:10.4f} valid {valid} mem {torch.cuda.memory_allocated() / 2**30:6.2f} GiB (peak {torch.cuda.max_memory_allocated() / 2**30:6.2f})", flush=True)
    assert valid and lv == lv
print(f"{N} steps in {time.time() - t0:.1f} s")
