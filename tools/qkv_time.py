"""qkv_ws / flash launch times of the attention operator through the library's profiler:  python tools/qkv_time.py B C H W"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
B, Cc, H, W = map(int, sys.argv[1:5])
g = C.rng(0)
x = (torch.randn(B, Cc, H, W, generator=g) * 1.2 + 0.3).cuda()
n = lambda t: t.numpy().copy()
gm, bt = n(1 + 0.25 * torch.randn(Cc, generator=g)), n(0.2 * torch.randn(Cc, generator=g))
wq = n(torch.randn(3 * Cc, Cc, 1, 1, generator=g) * math.sqrt(3.0 / Cc))
wo, bo = n(torch.randn(Cc, Cc, 1, 1, generator=g) * math.sqrt(1.5 / Cc)), n(torch.randn(Cc, generator=g) * 0.1)
y = torch.empty_like(x)
best = {}
for r in range(5):
    ulib.check(L.ucdir_profile_enable(1 if r else 0))
    ulib.check(L.ucdir_op_attention(C._p(x), B, Cc, H, W, C._hp(gm), C._hp(bt), C._hp(wq), C._hp(wo), C._hp(bo), 0, C._p(y), C._st()))
    ulib.check(L.ucdir_profile_enable(0))
    if r:
        cap = 64
        keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
        ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        nr = ctypes.c_int32(0)
        ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
        for i in range(nr.value):
            k = int(keys[i]); t = ms[i] / max(ln[i], 1)
            best[k] = min(best.get(k, 1e9), t)
print(" ".join("key %d %.1f us" % (k, t * 1e3) for k, t in sorted(best.items())), "|y| %.5f" % float(y.abs().mean()))
