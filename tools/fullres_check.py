"""One denoising step of the full-resolution inter-step patch split (BASELINE.json configs[2]):
1424x2128 SID frame -> +64 reflect pad (DDPM.test) -> six 1024^2 windows (utils/util.py:108-146)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hip_checks as C
from ucdir_amd.spec import UNetConfig
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)
net, sd = C.build_net(SID)
H, W = 1424 + 128, 2128 + 128
g = torch.Generator().manual_seed(0)
cond = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).cuda()
x_t = torch.randn(1, 3, H, W, generator=g).cuda()
guide = (cond * 0.5).contiguous()
lvl = torch.full((1, 1), 0.3, device="cuda")
with torch.no_grad():
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        eps = net.denoise_fn(torch.cat([cond, x_t], 1), lvl, guide)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"step {i}: {dt*1e3:.1f} ms  eps mean {eps.mean().item():.4f} std {eps.std().item():.4f} finite {bool(torch.isfinite(eps).all())} shape {tuple(eps.shape)}")
print("workspace GB", net.denoise_fn._handle() and __import__('ucdir_amd.lib', fromlist=['x']).load().ucdir_workspace_bytes(net.denoise_fn._handle()) / 1e9)
print("max mem GB", torch.cuda.max_memory_allocated() / 1e9)
