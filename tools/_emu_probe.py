import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hip_checks as C
from ucdir_amd.spec import UNetConfig
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)
SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
for cfg, name, H, B in ((SMALL, "small", 64, 2), (SID, "sid", 256, 1), (SID, "sid", 256, 4)):
    net_sd = C.build_net(cfg)
    out = C.layerwise_emu_case(cfg, B, H, H, [0.4, 0.003, 0.8, 0.95][:B], seed=31, net_sd=net_sd)
    print(name, B, "worst layer %.3e" % max(v["rel_rms"] for v in out.values()))
    for k, v in out.items():
        print("   %-28s rel_rms %.3e  max_abs %.3e" % (k, v["rel_rms"], v["max_abs"]))
