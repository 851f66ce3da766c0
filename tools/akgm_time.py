"""AKGM-tail launch times at the bench configuration's levels (B = 16) through the library's profiler:  python tools/akgm_time.py [levels]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
B = 16
levels = [int(v) for v in sys.argv[1:]] or [72, 36, 18, 144, 288]
CH = {288: 64, 144: 128, 72: 256, 36: 512, 18: 512}


def prof_read():
    cap = 64
    keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    nr = ctypes.c_int32(0)
    ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
    return [(int(keys[i]), int(ln[i]), float(ms[i]), float(fl[i])) for i in range(nr.value)]


for H in levels:
    Cc = CH[H]
    g = C.rng(0)
    h = torch.randn(B, Cc, H, H, generator=g).cuda(); att = torch.randn(B, 8, H, H, generator=g).cuda(); res = torch.randn(B, Cc, H, H, generator=g).cuda()
    wsp = (torch.randn(8 * Cc, Cc // 8, 3, 3, generator=g) * 0.1).numpy().copy(); bsp = np.zeros(8 * Cc, np.float32)
    gm = np.ones(Cc, np.float32); bt = np.zeros(Cc, np.float32)
    y = torch.empty(B, Cc, H, H, device="cuda")
    best, keys = None, []
    for r in range(5):
        ulib.check(L.ucdir_profile_enable(1 if r else 0))
        ulib.check(L.ucdir_op_akgm(C._p(h), C._p(att), C._p(res), B, Cc, H, H, C._hp(wsp), C._hp(bsp), C._hp(gm), C._hp(bt), C._p(y), C._hp(None), C._st()))
        ulib.check(L.ucdir_profile_enable(0))
        if r:
            rows = prof_read()
            t = sum(x[2] for x in rows); keys = [x[0] for x in rows]
            best = t if best is None or t < best else best
    print(f"akgm {H}^2 C={Cc}: {best * 1e3:7.1f} us keys {keys} |y| {float(y.abs().mean()):.4f}")
