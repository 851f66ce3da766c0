"""conv3x3_halo launch times (ups.16 = 192 -> 64 + res_conv at 288^2; 64 -> 128 + res at 144^2; the 144^2-grid Upsample) through the profiler."""
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


def run(fn):
    best = None
    for r in range(4):
        ulib.check(L.ucdir_profile_enable(1 if r else 0))
        fn()
        ulib.check(L.ucdir_profile_enable(0))
        if r:
            rows = prof_read()
            t = sum(x[2] for x in rows)
            best = t if best is None or t < best else best
            keys = [x[0] for x in rows]
    return best * 1e3, keys


g = C.rng(0)
for (H, c0, c1, cout, kind) in [(288, 128, 64, 64, "res"), (144, 64, 0, 128, "res"), (144, 128, 0, 128, "up"), (288, 64, 64, 64, "res")]:
    cin = c0 + c1
    x0 = torch.randn(B, c0, H, H, generator=g).cuda()
    x1 = torch.randn(B, c1, H, H, generator=g).cuda() if c1 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (9 * cin))).numpy().copy()
    wr = (torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(1.5 / cin)).numpy().copy()
    b = np.zeros(cout, np.float32); gm = np.ones(cin, np.float32); bt = np.zeros(cin, np.float32)
    Ho = 2 * H if kind == "up" else H
    y = torch.empty(B, cout, Ho, Ho, device="cuda"); yr = torch.empty(B, cout, H, H, device="cuda")
    if kind == "res":
        fn = lambda: ulib.check(L.ucdir_op_conv_res(C._p(x0), c0, C._p(x1), c1, B, H, H, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), C._hp(wr), C._hp(b),
                                                    cout, 1, C._p(y), C._p(yr), C._hp(None), C._st()))
    else:
        fn = lambda: ulib.check(L.ucdir_op_conv(C._p(x0), c0, C._p(None), 0, B, H, H, C._hp(w), C._hp(b), C._hp(None), C._hp(None), cout, 3, 2, 0,
                                                C._p(None), C._p(y), C._hp(None), C._st()))
    t, keys = run(fn)
    print(f"{H}^2 {cin}->{cout} {kind}: {t:7.1f} us keys {keys} |y| {float(y.abs().mean()):.4f}")
