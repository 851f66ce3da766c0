"""Side measurements for DESIGN.md (not the headline metric): the test()-equivalent 416^2 path, the T = 100 config
and the 20-step DPM-Solver++ sampler, same synthetic weights / inputs as bench.py.  python tools/extra_configs.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ucdir_amd import networks
from ucdir_amd.weights import synth_inputs, synth_state_dict

dev = torch.device("cuda:0")
net = networks.define_G(bench.sid_opt())
sd = synth_state_dict(net.denoise_fn.cfg, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
net = net.to(dev).eval()


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def sched(T):
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=0.4), dev)


for name, B, S, T in (("headline 256^2 B=16 T=50", 16, 256, 50), ("test()-equivalent: 384^2 input -> 416^2 compute, B=16 T=50", 16, 384, 50),
                      ("B=32 T=100 (BASELINE configs[3])", 32, 256, 100)):
    sched(T)
    x = torch.from_numpy(synth_inputs(B, S, S, seed=0)[0]).to(dev)
    dt = timed(lambda: net.super_resolution(x, False))
    print(f"{name}: {B / dt:.2f} img/s ({dt * 1e3 / T:.2f} ms per UNet forward)")
sched(50)
x = torch.from_numpy(synth_inputs(16, 256, 256, seed=0)[0]).to(dev)
with torch.no_grad():
    initx = net.predictor(x)
dt = timed(lambda: net.dpm_solver_sample(x, steps=20, order=2, kwargs={"guide": initx}) + initx)
print(f"DPM-Solver++ 20 steps, 256^2 B=16: {16 / dt:.2f} img/s")

# BASELINE configs[4]: 512^2 JPEG-restoration geometry: inter-step patch split with skip 256 / padding 32 (nine 256^2 windows
# per step, one engine batch), fp16 attention operands, T = 50
opt = bench.sid_opt()
opt["model"]["unet"]["attn_dtype"] = "fp16"
net4 = networks.define_G(opt)
net4.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
net4 = net4.to(dev).eval()
net4.set_new_noise_schedule(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4), dev)
dn = net4.denoise_fn
dn.patch_threshold, dn.patch_skip, dn.patch_padding, dn.patch_max_batch = 0, 256, 32, 9
x = torch.from_numpy(synth_inputs(1, 512, 512, seed=0)[0]).to(dev)
dt = timed(lambda: net4.super_resolution(x, False))
print(f"configs[4]: 512^2, 9 windows of 256^2 per step (skip 256 / pad 32), fp16 attention, T=50: {1 / dt:.3f} img/s "
      f"({dt * 1e3 / 50:.2f} ms per step)")
