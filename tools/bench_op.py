"""Micro-benchmark of single operators through the C ABI (for rocprofv3 --pmc runs).
  python tools/bench_op.py conv B H W cin cout [reps]      (3x3 stride-1 conv with GN fold + swish)
  python tools/bench_op.py akgm B H W C [reps]
  python tools/bench_op.py attn B C H W [reps]
"""
import os, ctypes, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
kind = sys.argv[1]
if os.environ.get("BENCH_CONVSK"):      # force a conv_sk kind (ucdir_debug_flag("convsk", n)) regardless of the size thresholds
    ulib.check(L.ucdir_debug_flag(b"convsk", int(os.environ["BENCH_CONVSK"])))
if kind == "conv":
    B, H, W, cin, cout = map(int, sys.argv[2:7]); reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    g = C.rng(0)
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (9 * cin))).numpy().copy()
    b = np.zeros(cout, np.float32); gm = np.ones(cin, np.float32); bt = np.zeros(cin, np.float32)
    y = torch.empty(B, cout, H, W, device="cuda")
    for _ in range(reps):
        ulib.check(L.ucdir_op_conv(C._p(x), cin, C._p(None), 0, B, H, W, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), cout, 3, 0, 1,
                                   C._p(None), C._p(y), C._hp(None), C._st()))
    torch.cuda.synchronize()
    print("done", float(y.abs().mean()))
elif kind == "convres":     # python tools/bench_op.py convres B H W c0 c1 cout [reps]
    B, H, W, c0, c1, cout = map(int, sys.argv[2:8]); reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
    g = C.rng(0)
    cin = c0 + c1
    x0 = torch.randn(B, c0, H, W, generator=g).cuda(); x1 = torch.randn(B, c1, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (9 * cin))).numpy().copy()
    wr = (torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(1.5 / cin)).numpy().copy()
    b = np.zeros(cout, np.float32); gm = np.ones(cin, np.float32); bt = np.zeros(cin, np.float32)
    y = torch.empty(B, cout, H, W, device="cuda"); yr = torch.empty(B, cout, H, W, device="cuda")
    for _ in range(reps):
        ulib.check(L.ucdir_op_conv_res(C._p(x0), c0, C._p(x1), c1, B, H, W, C._hp(w), C._hp(b), C._hp(gm), C._hp(bt), C._hp(wr), C._hp(b),
                                       cout, 1, C._p(y), C._p(yr), C._hp(None), C._st()))
    torch.cuda.synchronize()
    print("done", float(y.abs().mean()), float(yr.abs().mean()))
elif kind == "akgm":
    B, H, W, Cc = map(int, sys.argv[2:6]); reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    g = C.rng(0)
    h = torch.randn(B, Cc, H, W, generator=g).cuda(); att = torch.randn(B, 8, H, W, generator=g).cuda(); res = torch.randn(B, Cc, H, W, generator=g).cuda()
    wsp = (torch.randn(8 * Cc, Cc // 8, 3, 3, generator=g) * 0.1).numpy().copy(); bsp = np.zeros(8 * Cc, np.float32)
    gm = np.ones(Cc, np.float32); bt = np.zeros(Cc, np.float32)
    y = torch.empty(B, Cc, H, W, device="cuda")
    for _ in range(reps):
        ulib.check(L.ucdir_op_akgm(C._p(h), C._p(att), C._p(res), B, Cc, H, W, C._hp(wsp), C._hp(bsp), C._hp(gm), C._hp(bt), C._p(y), C._hp(None), C._st()))
    torch.cuda.synchronize()
    print("done", float(y.abs().mean()))
elif kind == "attn":        # python tools/bench_op.py attn B C H W [reps]   (flash kernel forced; UCDIR_LIB=timing build prints the phase stamps)
    B, Cc, H, W = map(int, sys.argv[2:6]); reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    g = C.rng(0)
    x = (torch.randn(B, Cc, H, W, generator=g) * 1.2 + 0.3).cuda()
    n = lambda t: t.numpy().copy()
    gm, bt = n(1 + 0.25 * torch.randn(Cc, generator=g)), n(0.2 * torch.randn(Cc, generator=g))
    wq = n(torch.randn(3 * Cc, Cc, 1, 1, generator=g) * math.sqrt(3.0 / Cc))
    wo, bo = n(torch.randn(Cc, Cc, 1, 1, generator=g) * math.sqrt(1.5 / Cc)), n(torch.randn(Cc, generator=g) * 0.1)
    y = torch.empty_like(x)
    ulib.check(L.ucdir_debug_flag(b"flash", 1))
    for _ in range(reps):
        ulib.check(L.ucdir_op_attention(C._p(x), B, Cc, H, W, C._hp(gm), C._hp(bt), C._hp(wq), C._hp(wo), C._hp(bo), 0, C._p(y), C._st()))
    torch.cuda.synchronize()
    print("done", float(y.abs().mean()))
