"""How far is the bf16-mode CTC loss from the fp32-mode loss on the metric config (same weights, same batch)?"""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from asr_amd import DeepSpeech, CTCLoss
dev = torch.device("cuda:0")
for wl in (sys.argv[1:] or ["c3"]):
    rnn, H, L, C, B, tin = bench.WORKLOADS[wl]
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        model = DeepSpeech(audio_conf=bench.audio_conf(), decoder=None, label_path=bench.label_file(tmp, C), rnn_type=rnn,
                           rnn_hidden_size=H, rnn_hidden_layers=L, bidirectional=True)
    model.to(dev).train()
    x, targets, pct, tsz = bench.synthetic_batch(B, tin, C, 1, ragged=(wl == "c5"))
    lens = (pct * x.size(3)).int()
    res = {}
    for prec in ("fp32", "bf16"):
        model.precision = prec
        with torch.no_grad():
            out, out_lens = model.forward(x.to(dev), lens)
            loss = CTCLoss(reduction="sum")(out.transpose(0, 1), targets, out_lens, tsz) / B
        res[prec] = (float(loss), out.float().cpu())
    l32, l16 = res["fp32"][0], res["bf16"][0]
    d = (res["bf16"][1] - res["fp32"][1]).norm() / res["fp32"][1].norm()
    print(f"{wl}: loss fp32 {l32:.6f}  bf16 {l16:.6f}  rel gap {abs(l16 - l32) / abs(l32):.3e}  logits rel_l2 {float(d):.3e}", flush=True)
