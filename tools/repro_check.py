"""Find the first layer whose output differs between two identical forwards (determinism check)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hip_checks as C
from ucdir_amd.spec import UNetConfig, unet_layers
from ucdir_amd.weights import synth_inputs
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
net, sd = C.build_net(SID)
cond, guide, x_t = map(torch.from_numpy, synth_inputs(B, S, S, seed=4))
lvl = torch.linspace(0.01, 0.99, B).reshape(B, 1).cuda()
x6 = torch.cat([cond, x_t], 1).cuda(); g = guide.cuda()
runs = []
for r in range(3):
    with torch.no_grad():
        eps = net.denoise_fn(x6, lvl, g).clone()
    torch.cuda.synchronize()
    taps = {}
    for Ld in unet_layers(SID):
        taps[Ld.name] = net.denoise_fn.debug_read(Ld.name, "out").clone()
        if Ld.kind == "block":
            taps[Ld.name + ":h1"] = net.denoise_fn.debug_read(Ld.name, "h1").clone()
    torch.cuda.synchronize()
    runs.append((eps, taps))
for r in (1, 2):
    print("run", r, "eps equal:", torch.equal(runs[0][0], runs[r][0]))
    shown = 0
    for k in runs[0][1]:
        a, b = runs[0][1][k], runs[r][1][k]
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            nz = (d > 0).nonzero()
            print("  first differing tap:", k, "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel(), "first idx", nz[0].tolist(), "last idx", nz[-1].tolist())
            shown += 1
            if shown >= 4: break
