import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ab_ksplit import *
G, H, B, T = 3, 1024, 64, 3
gx, whh, bhh, lens, dy = setup(G, H, B, T, ragged=False)
wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
fw = ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=True, packed_gates=True)
ks = bwd(G, H, B, T, fw, wpb, lens, dy, 0)[0].float().view(T, B, 2, G, H)
st = bwd(G, H, B, T, fw, wpb, lens, dy, 64)[0].float().view(T, B, 2, G, H)
d = (ks - st)
print("rel err per t:", [f"{float(d[t].norm()/st[t].norm()):.2e}" for t in range(T)])
print("rel err per dir:", [f"{float(d[:, :, k].norm()/st[:, :, k].norm()):.2e}" for k in range(2)])
print("rel err per gate:", [f"{float(d[:, :, :, g].norm()/st[:, :, :, g].norm()):.2e}" for g in range(G)])
e = d.pow(2).sum((0, 2, 3))          # (B, H)
print("err energy per row%16:", [f"{float(e.view(B // 16, 16, H).sum((0, 2))[r]):.2e}" for r in range(16)])
print("err energy per unit%32:", [f"{float(e.view(B, H // 32, 32).sum((0, 1))[u]):.2e}" for u in range(32)])
print("err energy per slice (first 8):", [f"{float(e.view(B, H // 32, 32).sum((0, 2))[u]):.2e}" for u in range(8)])
