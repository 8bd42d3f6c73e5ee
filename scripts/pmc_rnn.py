"""Run ONLY the recurrent forward and backward kernels at the bench shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
G, H, B = 3, 1024, 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 501
bf = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
dev = torch.device("cuda:0")
gx = torch.randn(T * B, 2 * G * H, device=dev) * 0.5
whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
bhh = torch.zeros(2, G * H, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
wpf, wpb = ops.rnn_pack(G, whh, bf16=bf)
# the train step's own mode: packed gate records, bf16 copies of h / d(hn) and bias partial sums for the TN-form weight gradients
h_bf = torch.empty(T * B, 2 * H, dtype=torch.bfloat16, device=dev) if bf else None
out = ops.rnn_fwd(G, gx, wpf, bhh, lens, T, B, H, bf16=bf, packed_gates=bf, h_bf16=h_bf)
if bf:
    hb, aux, rec = out
    dy = torch.randn(T * B, H, device=dev)
    side = torch.empty(T * B, 2 * G * H, dtype=torch.bfloat16, device=dev)
    dhn = torch.empty(T * B, 2 * H, dtype=torch.bfloat16, device=dev)
    bias_part = torch.empty(B, 2, 4, H, device=dev)
    # as the train step calls it: the BatchNorm1d backward of the layer above applied inside the kernel (ds2_rnn_bwd_bn)
    bn_x = torch.randn(T * B, H, device=dev)
    mean, var, gamma = bn_x.mean(0), bn_x.var(0, unbiased=False), torch.ones(H, device=dev)
    sums = ops.bn1d_bwd_sums(dy, bn_x, mean, var, gamma)
    ops.rnn_bwd_bn(G, dy, bn_x, mean, var, gamma, sums, None, aux, hb, wpb, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=rec,
                   dhn_bf16=dhn, bias_part=bias_part)
    assert ops.rnn_last_path() & 16, "the fused K-split launch did not take the call"
torch.cuda.synchronize()
print("done", T, "launches")
