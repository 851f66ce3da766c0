"""Per-layer A/B of the 3x3 conv kernels at the bench configuration's shapes (B = 16), through the C ABI with the library's own
HIP-event profiler:  python tools/conv_layers.py [reps] [B] [filter]
Prints, per (H, cin -> cout, mode), the launch time and TFLOP/s with the stream-K kernel off (one-shot conv3x3_halo) and on."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
flt = sys.argv[3] if len(sys.argv) > 3 else ""
SHAPES = [  # H, c0, c1, cout, mode (0 conv1, 2 Upsample conv on the H x H input grid)
    (72, 512, 256, 256, 0), (72, 256, 256, 256, 0), (72, 256, 128, 256, 0), (72, 256, 0, 256, 0), (72, 128, 0, 256, 0),
    (36, 512, 512, 512, 0), (36, 512, 256, 512, 0), (36, 512, 0, 512, 0), (36, 256, 0, 512, 0),
    (18, 512, 512, 512, 0), (18, 512, 0, 512, 0),
    (72, 256, 0, 256, 2), (36, 512, 0, 512, 2), (18, 512, 0, 512, 2),
    (144, 256, 128, 128, 0), (144, 128, 128, 128, 0), (144, 128, 64, 128, 0), (144, 128, 0, 128, 0), (144, 64, 0, 128, 0), (144, 128, 0, 128, 2),
]


def prof_read():
    cap = 64
    keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    nr = ctypes.c_int32(0)
    ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
    return [(int(keys[i]), int(ln[i]), float(ms[i]), float(fl[i])) for i in range(nr.value)]


for (H, c0, c1, cout, mode) in SHAPES:
    tag = f"{H}^2 {c0}+{c1}->{cout} {'up' if mode == 2 else 'conv'}"
    if flt and flt not in tag:
        continue
    g = C.rng(0)
    cin = c0 + c1
    x0 = (torch.randn(B, c0, H, H, generator=g)).cuda()
    x1 = (torch.randn(B, c1, H, H, generator=g)).cuda() if c1 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (9 * cin))).numpy().copy()
    b = np.zeros(cout, np.float32)
    gn = mode == 0
    gm = np.ones(cin, np.float32) if gn else None
    bt = np.zeros(cin, np.float32) if gn else None
    Ho = 2 * H if mode == 2 else H
    y = torch.empty(B, cout, Ho, Ho, device="cuda")
    out = []
    for sk in (0, 1, 2):
        ulib.check(L.ucdir_debug_flag(b"convsk", sk))
        best = None
        for r in range(reps + 1):
            ulib.check(L.ucdir_profile_enable(1 if r else 0))
            ulib.check(L.ucdir_op_conv(C._p(x0), c0, C._p(x1), c1, B, H, H, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), cout, 3, mode, 1 if gn else 0,
                                       C._p(None), C._p(y), C._hp(None), C._st()))
            ulib.check(L.ucdir_profile_enable(0))
            if r:
                rows = [x for x in prof_read() if x[0] in (20, 21, 22, 120, 121, 125, 126, 127, 128, 129)]
                t = sum(x[2] for x in rows)
                best = t if best is None or t < best else best
                key = rows[0][0] if rows else -1
        flops = 2.0 * 9 * cin * cout * Ho * Ho * B
        out.append((key, best, flops / (best * 1e-3) / 1e12 if best else 0.0, float(y.abs().mean())))
    ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    print("%-26s old %3d %7.1f us %6.1f TF | persistent8 %3d %7.1f us %6.1f TF x%.2f | oneshot4 %3d %7.1f us %6.1f TF x%.2f  (|y| %.4f %.4f %.4f)" % (
        tag, out[0][0], out[0][1] * 1e3, out[0][2], out[1][0], out[1][1] * 1e3, out[1][2], out[0][1] / max(out[1][1], 1e-9),
        out[2][0], out[2][1] * 1e3, out[2][2], out[0][1] / max(out[2][1], 1e-9), out[0][3], out[1][3], out[2][3]))
    sys.stdout.flush()

