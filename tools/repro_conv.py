"""Determinism of single operators at bench size: run ucdir_op_conv several times on the same device inputs and compare
outputs and output statistics bit-wise.  python tools/repro_conv.py B H W cin cout ksize mode gn"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
B, H, W, cin, cout, ks, mode, gn = map(int, sys.argv[1:9])
g = C.rng(0)
x = C.bfr(torch.randn(B, cin, H, W, generator=g) * 1.3 + 0.6).cuda()
w = (torch.randn(cout, cin, ks, ks, generator=g) * math.sqrt(1.5 / (cin * ks * ks))).numpy().copy()
b = (torch.randn(cout, generator=g) * 0.1).numpy().copy()
gm = (1 + 0.25 * torch.randn(cin, generator=g)).numpy().copy() if gn else None
bt = (0.2 * torch.randn(cin, generator=g)).numpy().copy() if gn else None
Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
outs, sts = [], []
for r in range(4):
    y = torch.empty(B, cout, Ho, Wo, device="cuda")
    st = np.zeros((B, 2), dtype=np.float64)
    ulib.check(L.ucdir_op_conv(C._p(x), cin, C._p(None), 0, B, H, W, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), cout, ks, mode, 1,
                               C._p(None), C._p(y), C._hp(st), C._st()))
    torch.cuda.synchronize()
    outs.append(y.clone()); sts.append(st.copy())
ref = np.stack([outs[0].double().sum(dim=(1, 2, 3)).cpu().numpy(), outs[0].double().pow(2).sum(dim=(1, 2, 3)).cpu().numpy()], 1)
for r in range(4):
    rel = (sts[r] - ref) / np.abs(ref)
    print("run", r, "stats vs torch sums of the (bf16-rounded) output: max rel", float(np.abs(rel).max()), "S rel per sample:", np.array2string(rel[:, 0], precision=2, max_line_width=250))
for r in range(1, 4):
    d = (outs[0] - outs[r]).abs()
    print("run", r, "out equal:", torch.equal(outs[0], outs[r]), "ndiff", int((d > 0).sum()), "stats equal:", np.array_equal(sts[0], sts[r]),
          "max stat rel diff", float(np.abs(sts[0] - sts[r]).max() / np.abs(sts[0]).max()))
