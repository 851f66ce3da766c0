import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hip_checks as C
from ucdir_amd.spec import UNetConfig
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)
SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
for cfg, name, H in ((SMALL, "small", 64), (SID, "sid", 256)):
    net_sd = C.build_net(cfg)
    for seed, lv in ((31, 0.4), (21, 0.0029)):
        out, eps, ref = C.forward_case(cfg, 1, H, H, [lv], seed=seed, taps=True, net_sd=net_sd, emu=True)
        print(name, seed, lv, "eps vs oracle %.3e  eps vs emu %.3e  emu vs oracle %.3e" % (out["eps"]["rel_rms"], out["eps_emu"]["rel_rms"], out["emu_vs_oracle"]["rel_rms"]))
        for k, v in out.items():
            if k.endswith("@emu"):
                base = k[:-4]
                print("   %-28s vs oracle %.3e   vs emu %.3e" % (base, out[base]["rel_rms"], v["rel_rms"]))
