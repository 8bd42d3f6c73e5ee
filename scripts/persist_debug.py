"""Runs fused train steps with a small polling limit and prints the persistent kernel's starvation record (debug helper)."""
import os, sys, tempfile, ctypes
os.environ.setdefault("DS2_RNN_SPIN_LIMIT", "20000")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from asr_amd import CTCLoss, DeepSpeech, FusedAdamW, _lib, ops
from asr_amd.trainers import DeepSpeechTrainer
lib = _lib.load()
dev = torch.device("cuda:0")
rnn, H, L, C, B, tin = bench.WORKLOADS["c3"]
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as tmp:
    model = DeepSpeech(audio_conf=bench.audio_conf(), decoder=None, label_path=bench.label_file(tmp, C), rnn_type=rnn, rnn_hidden_size=H,
                       rnn_hidden_layers=L, bidirectional=True)
model.to(dev).train(); model.precision = "bf16"
opt = FusedAdamW(model, lr=1.5e-4)
tr = DeepSpeechTrainer(model, CTCLoss(reduction="sum"), 1, None, opt, None, None, dev, dev, False, None)
x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1, ragged=bool(int(os.environ.get("RAGGED", "0"))))
x = x.to(dev)
rec = (ctypes.c_int * 8)()
for i in range(int(os.environ.get("NSTEPS", "30"))):
    valid, lv = tr.step((x, targets, pct.clone(), tsz))
    lib.ds2_rnn_persistent_status(ops._ctxp(), ctypes.cast(rec, ctypes.c_void_p))
    print(f"step {i}: loss {lv} valid {valid} record {list(rec)}", flush=True)
