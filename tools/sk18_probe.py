"""18^2-level 3x3 convs at B = 16: default (K cut in two + finish pass) vs whole-K launches of SHORT units (skmix forced, UCDIR_SK_KSPLIT=0).
    UCDIR_SK_KSPLIT=0 python tools/sk18_probe.py"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
B = 16


def prof_read():
    cap = 64
    keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    nr = ctypes.c_int32(0)
    ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
    return [(int(keys[i]), int(ln[i]), float(ms[i]), float(fl[i])) for i in range(nr.value)]


for (H, c0, c1, cout) in [(18, 512, 512, 512), (18, 512, 0, 512), (36, 512, 512, 512), (36, 256, 0, 512)]:
    g = C.rng(0)
    cin = c0 + c1
    x0 = torch.randn(B, c0, H, H, generator=g).cuda()
    x1 = torch.randn(B, c1, H, H, generator=g).cuda() if c1 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (9 * cin))).numpy().copy()
    b = np.zeros(cout, np.float32)
    gm, bt = np.ones(cin, np.float32), np.zeros(cin, np.float32)
    y = torch.empty(B, cout, H, H, device="cuda")
    res = []
    for mix in (-1, 0, 1):
        ulib.check(L.ucdir_debug_flag(b"convsk", 2))
        ulib.check(L.ucdir_debug_flag(b"skmix", mix))
        best, key = None, -1
        for r in range(4):
            ulib.check(L.ucdir_profile_enable(1 if r else 0))
            ulib.check(L.ucdir_op_conv(C._p(x0), c0, C._p(x1), c1, B, H, H, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), cout, 3, 0, 1,
                                       C._p(None), C._p(y), C._hp(None), C._st()))
            ulib.check(L.ucdir_profile_enable(0))
            if r:
                rows = prof_read()
                t = sum(x[2] for x in rows)
                key = rows[0][0] if rows else -1
                best = t if best is None or t < best else best
        res.append((mix, key, best * 1e3, float(y.abs().mean())))
    ulib.check(L.ucdir_debug_flag(b"skmix", -1)); ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    print(f"{H}^2 {cin}->{cout}: " + " | ".join("skmix %2d key %d %7.1f us (|y| %.4f)" % r for r in res))
