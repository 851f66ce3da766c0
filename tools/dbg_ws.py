"""Debug helper: where does akgm (C = 64) differ from the oracle?  python tools/dbg_ws.py B H W [grid]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
import hip_checks as C
from oracle import ucdir_oracle as O
from ucdir_amd import lib as ulib
L = ulib.load()
B, H, W = map(int, sys.argv[1:4]); grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
Cc = 64
g = C.rng(5)
h = C.bfr(torch.randn(B, Cc, H, W, generator=g).abs() * 0.8 - 0.2)
att = torch.randn(B, 8, H, W, generator=g) * 0.5
res = C.bfr(torch.randn(B, Cc, H, W, generator=g))
if os.environ.get('DBG_RES0'): res = torch.zeros_like(res)
if os.environ.get('DBG_ATT1'): att = torch.full_like(att, 0.3)
if os.environ.get('DBG_H1'): h = torch.full_like(h, 0.5)
wsp = torch.randn(8 * Cc, Cc // 8, 3, 3, generator=g) * math.sqrt(1.5 / (9 * Cc // 8))
bsp = torch.randn(8 * Cc, generator=g) * 0.1
gamma = 1 + 0.25 * torch.randn(Cc, generator=g); beta = 0.2 * torch.randn(Cc, generator=g)
hn = F.group_norm(h, 1, gamma, beta, eps=1e-5)
hset = F.conv2d(hn, wsp, bsp, padding=1, groups=8).view(B, Cc, 8, H, W)
y = O.swish((hset * att.unsqueeze(1)).sum(2)) + res
ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
first = None
for rep in range(int(os.environ.get('DBG_REPS', '3'))):
    dy = torch.full((B, Cc, H, W), 7.0, device="cuda")
    dh, datt, dres = h.cuda(), att.cuda(), res.cuda()
    st = np.zeros((B, 2))
    ulib.check(L.ucdir_op_akgm(C._p(dh), C._p(datt), C._p(dres), B, Cc, H, W, C._hp(wsp.numpy().copy()), C._hp(bsp.numpy().copy()),
                               C._hp(gamma.numpy().copy()), C._hp(beta.numpy().copy()), C._p(dy), C._hp(st), C._st()))
    torch.cuda.synchronize()
    d = (dy.cpu() - y)
    bad = (~torch.isfinite(d)) | (d.abs() > 0.05)
    if first is None: first = dy.clone()
    if not torch.equal(first, dy): print("rep", rep, "NOT bit-identical to rep 0:", int((first != dy).sum()), "elements")
    if rep < 3 or bad.any(): print("rep", rep, "bad elements", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(dy).sum()))
    if bad.any():
        idx = bad.nonzero()
        bs, cs, ys, xs = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        tiles = set(zip(bs.tolist(), (ys // 16).tolist(), (xs // 16).tolist()))
        print("  samples", sorted(set(bs.tolist())), "groups", sorted(set((cs // 8).tolist())), "n tiles", len(tiles))
        tl = sorted(tiles)[:40]
        print("  tiles (b, ty, tx) -> linear", [(t, t[0] * ((H + 15) // 16) * ((W + 15) // 16) + t[1] * ((W + 15) // 16) + t[2]) for t in tl])
        import collections
        gh = collections.Counter()
        for (bb, ty, tx) in tiles:
            sub = bad[bb, :, ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
            gh[tuple(int(sub[8 * gq:8 * gq + 8].sum() > 40) for gq in range(8))] += 1
        print("  bad-group patterns over tiles:", dict(gh))
        t0 = tl[0]
        sub = bad[t0[0], :, t0[1] * 16:(t0[1] + 1) * 16, t0[2] * 16:(t0[2] + 1) * 16]
        print("  first bad tile: per-group bad counts", [int(sub[8 * gq:8 * gq + 8].sum()) for gq in range(8)], "rows with bad", sorted(set(sub.nonzero()[:, 1].tolist())))
        pm = bad[t0[0], :, t0[1] * 16:(t0[1] + 1) * 16, t0[2] * 16:(t0[2] + 1) * 16].sum(0)
        print("  bad channels per pixel (rows 0-5):")
        for rr in range(6): print("   ", pm[rr].tolist())
        dd = d[t0[0], :, t0[1] * 16:(t0[1] + 1) * 16, t0[2] * 16:(t0[2] + 1) * 16]
        print("  |err| of pixel (0,0) per channel:", [float("%.3g" % v) for v in dd[:, 0, 0].abs().tolist()][:24])
        print("  sample values", dy.cpu()[t0[0], 0, t0[1] * 16:t0[1] * 16 + 2, t0[2] * 16:t0[2] * 16 + 4], y[t0[0], 0, t0[1] * 16:t0[1] * 16 + 2, t0[2] * 16:t0[2] * 16 + 4])
