"""One-shot GPU diagnostic: run every parity check and print a compact table (for gpurun)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import hip_checks as C  # noqa: E402
from ucdir_amd.spec import UNetConfig  # noqa: E402

SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)


def run(name, fn, *a, **k):
    t0 = time.time()
    try:
        r = fn(*a, **k)
        if isinstance(r, tuple):
            r = r[0]
        print(f"[{name}] {time.time() - t0:.1f}s", json.dumps(r, default=float))
    except Exception as e:  # keep going: one call should report everything
        print(f"[{name}] FAILED {type(e).__name__}: {e}")
        traceback.print_exc()
    sys.stdout.flush()


def sid_b4():
    import ctypes
    import numpy as np
    from oracle import ucdir_oracle as O
    from ucdir_amd.weights import synth_inputs
    net, sd = C.build_net(SID)
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(4, 256, 256, seed=31))
    lvl = torch.tensor([[0.0029], [0.3], [0.6], [0.95]])
    x6 = torch.cat([cond, x_t], 1)
    L = C.ulib.load()
    C.ulib.check(L.ucdir_profile_enable(1))
    with torch.no_grad():
        eps = net.denoise_fn(x6.cuda(), lvl.cuda(), guide.cuda()).cpu()
    C.ulib.check(L.ucdir_profile_enable(0))
    cap = 64
    keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    nr = ctypes.c_int32(0)
    C.ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
    out = {"keys": sorted(int(keys[i]) for i in range(nr.value))}
    for b in (0, 3):
        out[f"s{b}"] = C.metrics(eps[b:b + 1], O.dy3h_forward(sd, x6[b:b + 1], lvl[b:b + 1], guide[b:b + 1]))
    return out


def main():
    which = sys.argv[1:] or ["ops", "small", "sid", "sampler"]
    print(torch.cuda.get_device_name(0))
    if "ops" in which:
        run("conv3x3 64->64 plain", C.conv_case, 2, 20, 20, 64, 0, 64, 3, 0, False, False, False)
        run("conv3x3 64->128 gn+silu", C.conv_case, 2, 24, 40, 64, 0, 128, 3, 0, True, True, False)
        run("conv3x3 cat(128+64)->128 gn+silu", C.conv_case, 2, 24, 40, 128, 64, 128, 3, 0, True, True, False)
        run("conv3x3 cat(128+64)->64 gn+silu", C.conv_case, 1, 33, 17, 128, 64, 64, 3, 0, True, True, False)
        run("conv down 128", C.conv_case, 2, 32, 32, 128, 0, 128, 3, 1, False, False, False)
        run("conv up 128", C.conv_case, 2, 16, 16, 128, 0, 128, 3, 2, False, False, False)
        run("conv up 64", C.conv_case, 1, 16, 24, 64, 0, 64, 3, 2, False, False, False)
        run("conv1x1 cat(128+64)->64 +res", C.conv_case, 2, 24, 40, 128, 64, 64, 1, 0, False, False, True)
        run("conv1x1 512->512 gn", C.conv_case, 1, 12, 12, 512, 0, 512, 1, 0, True, False, False)
        for Cc in (64, 128, 256, 512):
            run(f"akgm C={Cc}", C.akgm_case, 2, Cc, 20, 24)
        run("attention C=128 12x10", C.attention_case, 2, 128, 12, 10)
        run("attention C=512 36x36", C.attention_case, 1, 512, 36, 36)
        run("sampler_step", C.sampler_step_case)
    if "small" in which:
        run("forward SMALL 64x48 B=2 (taps)", C.forward_case, SMALL, 2, 64, 48, [0.0029, 0.6], taps=True)
    if "sidb4" in which:        # B = 4 at 256^2: the smallest batch at which the persistent kernels engage (4 tiles per CU at 288^2)
        run("forward SID B=4 256x256", sid_b4)
    if "sid" in which:
        ns = C.build_net(SID)
        run("forward SID 256x256 B=1 (taps)", C.forward_case, SID, 1, 256, 256, [0.2394], seed=21, taps=True, net_sd=ns)
        if "sampler" in which:
            run("sampler SID 64x64 T=8", C.sampler_case, SID, 64, 64, 8, net_sd=ns)


if __name__ == "__main__":
    main()
