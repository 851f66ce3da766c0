"""Generate tests/golden/*.npz by running the REAL reference (build container only).

Run:  python -O oracle/gen_golden.py        (from the repo root; -O strips the reference's
                                             ``assert noisy.is_cuda`` in utils/util.py:113)

The reference is imported from /root/reference (never copied); only input/output *data*
is written.  Weights come from ``ucdir_amd.weights.synth_state_dict`` so the GPU box can
regenerate them bit-identically; they are therefore NOT stored (except the tiny config's,
as a cross-check of the generator itself).
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.modules["lpips"] = types.ModuleType("lpips")  # model/diffusion.py:12 imports it at top level
    from model import networks  # noqa
    return networks


def ref_model(networks, unet_opt, seed=0):
    from ucdir_amd.spec import UNetConfig
    from ucdir_amd.weights import synth_state_dict
    opt = yaml.safe_load(open(os.path.join(REF, "config", "sid.yaml")))
    opt["model"]["unet"].update(unet_opt)
    net = networks.define_G(opt).eval()
    cfg = UNetConfig.from_opt(opt["model"]["unet"])
    sd = synth_state_dict(cfg, seed)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return net, cfg, sd


TINY = dict(inner_channel=8, channel_mults=[1, 2], res_blocks=1, attn_res=[64], image_size=128)
SMALL = dict(inner_channel=64, channel_mults=[1, 2, 4], res_blocks=1, attn_res=[32], image_size=128)
SCHED50 = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)
SCHED100 = dict(schedule="linear", n_timestep=100, linear_start=1e-6, linear_end=0.4)
SCHED8 = dict(schedule="linear", n_timestep=8, linear_start=1e-6, linear_end=0.4)


def extra_full_config(net):
    """SURVEY.md 8c fixtures (ii, B = 2) and (v, a real image) on the full SID configuration (round 3)."""
    from ucdir_amd.weights import synth_inputs
    # (ii) B = 2: two samples with different levels in ONE call (per-sample GroupNorm statistics, per-sample time embedding)
    cond, guide, x_t = synth_inputs(2, 256, 256, seed=22)
    x6 = torch.from_numpy(np.concatenate([cond, x_t], 1))
    lv = torch.tensor([[0.05], [0.7]], dtype=torch.float32)
    e = net.denoise_fn(x6, lv, guide=torch.from_numpy(guide))
    rec = {"levels": lv.numpy()}
    for b in range(2):
        rec[f"b{b}_stats"] = np.array([e[b].mean(), e[b].std(), e[b].min(), e[b].max()], dtype=np.float64)
        rec[f"b{b}_crop"] = e[b, :, 100:132, 60:92].numpy()
        rec[f"b{b}_ds"] = e[b, :, ::8, ::8].numpy()
    np.savez_compressed(os.path.join(OUT, "sid_forward_b2.npz"), **rec)
    # (v) a real image: 256^2 crop of dataset/celebahq_64_512/sr_64_512/00030.png as the condition, the reference's own
    # predictor output as the guide (ResiGaussianGuideDY.super_resolution, model/diffusion.py:473-478), one denoiser call
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(REF, "dataset", "celebahq_64_512", "sr_64_512", "00030.png")).convert("RGB"))
    crop = np.ascontiguousarray(img[128:384, 128:384])                                  # (256, 256, 3) uint8
    c = torch.from_numpy(crop).permute(2, 0, 1)[None].float() / 255.0 * 2.0 - 1.0       # [-1, 1] like data/LRHR_dataset.py
    gpred = net.predictor(c)
    xt = torch.from_numpy(synth_inputs(1, 256, 256, seed=23)[2])
    lv1 = torch.tensor([[0.239415851]], dtype=torch.float32)
    e = net.denoise_fn(torch.cat([c, xt], 1), lv1, guide=gpred)
    np.savez_compressed(os.path.join(OUT, "sid_real_image.npz"), cond_u8=crop, level=lv1.numpy(),
                        pred_stats=np.array([gpred.mean(), gpred.std(), gpred.min(), gpred.max()], dtype=np.float64),
                        pred_crop=gpred[0, :, 100:132, 60:92].numpy(),
                        eps_stats=np.array([e.mean(), e.std(), e.min(), e.max()], dtype=np.float64),
                        eps_crop=e[0, :, 100:132, 60:92].numpy(), eps_ds=e[0, :, ::8, ::8].numpy())


def main():
    from ucdir_amd.weights import synth_inputs
    os.makedirs(OUT, exist_ok=True)
    networks = import_reference()
    torch.set_grad_enabled(False)
    if "--extra-only" in sys.argv:           # only the round-3 fixtures (the others are unchanged)
        net, cfg, sd = ref_model(networks, {})
        extra_full_config(net)
        return

    # (iv) schedule tables ----------------------------------------------------------------------
    net, cfg, sd = ref_model(networks, TINY)
    for tag, sch in (("T50", SCHED50), ("T100", SCHED100), ("T8", SCHED8)):
        net.set_new_noise_schedule(sch, torch.device("cpu"))
        tabs = {k: v.numpy() for k, v in net.state_dict().items() if "." not in k}
        tabs["sqrt_alphas_cumprod_prev"] = net.sqrt_alphas_cumprod_prev
        np.savez_compressed(os.path.join(OUT, f"schedule_{tag}.npz"), **tabs)

    # (i) tiny config: forward, intermediates, predictor, 8-step sampler ---------------------------
    cond, guide, x_t = synth_inputs(2, 48, 40, seed=1)
    x6 = torch.from_numpy(np.concatenate([cond, x_t], 1))
    lvl = torch.tensor([[0.2394], [0.9]], dtype=torch.float32)
    g = torch.from_numpy(guide)
    eps = net.denoise_fn(x6, lvl, guide=g)
    pred = net.predictor(torch.from_numpy(cond))
    # one block + attention in isolation (downs.1 has attn at image_size 128/attn_res 64? -> level 1)
    temb = net.denoise_fn.noise_level_mlp(lvl)
    xb = torch.from_numpy(synth_inputs(2, 32, 32, seed=2)[2])[:, :1].repeat(1, 8, 1, 1) * \
        torch.linspace(0.5, 1.5, 8).view(1, 8, 1, 1)
    gb = torch.from_numpy(synth_inputs(2, 64, 64, seed=3)[1])
    blk = net.denoise_fn.downs[1]
    yb = blk.res_block(xb, temb, gb)
    attn_mod = net.denoise_fn.mid[0].attn
    xa = torch.from_numpy(synth_inputs(2, 12, 10, seed=4)[2])[:, :1].repeat(1, 16, 1, 1) * \
        torch.linspace(-1.0, 1.0, 16).view(1, 16, 1, 1)
    ya = attn_mod(xa)
    # 8-step sampler with recorded noise draws
    net.set_new_noise_schedule(SCHED8, torch.device("cpu"))
    torch.manual_seed(7)
    c1 = torch.from_numpy(cond[:1])
    st = torch.get_rng_state()
    out_c = net.super_resolution(c1, True)
    torch.set_rng_state(st)
    draws = [torch.randn(c1.shape)] + [torch.randn(c1.shape) for _ in range(7)]
    np.savez_compressed(
        os.path.join(OUT, "tiny_forward.npz"),
        cond=cond, guide=guide, x_t=x_t, level=lvl.numpy(), eps=eps.numpy(), predictor=pred.numpy(),
        temb=temb.numpy(), block_x=xb.numpy(), block_guide=gb.numpy(), block_y=yb.numpy(),
        attn_x=xa.numpy(), attn_y=ya.numpy(),
        sampler_out=out_c.numpy(), sampler_noise=np.stack([d.numpy() for d in draws]),
        **{"w::" + k: v for k, v in sd.items() if k.startswith("denoise_fn.")})

    # strided sampler (ddim_sample, 5 of 50 steps) with recorded noise draws
    net.set_new_noise_schedule(SCHED50, torch.device("cpu"))
    torch.manual_seed(9)
    st = torch.get_rng_state()
    gd = torch.from_numpy(guide[:1])
    out_d = net.ddim_sample(c1, False, kwargs={"guide": gd})
    torch.set_rng_state(st)
    ddraws = np.stack([torch.randn(c1.shape).numpy() for _ in range(6)])
    np.savez_compressed(os.path.join(OUT, "tiny_ddim.npz"), cond=cond[:1], guide=guide[:1], noise=ddraws, out=out_d.numpy())

    # (iii) patch_forward_guide at skip=128/padding=32 on 160x200 (tiny net) -----------------------
    from utils.util import patch_forward_guide
    cond, guide, x_t = synth_inputs(1, 160, 200, seed=5)
    x6 = torch.from_numpy(np.concatenate([cond, x_t], 1))
    lvl1 = torch.tensor([[0.5]], dtype=torch.float32)
    outp = patch_forward_guide(x6, net.denoise_fn.naiveforward,
                               params={"time": lvl1, "guide": torch.from_numpy(guide)}, skip=128, padding=32)
    np.savez_compressed(os.path.join(OUT, "tiny_patch.npz"), cond=cond, guide=guide, x_t=x_t,
                        level=lvl1.numpy(), out=outp.numpy())

    # small config (kernel-supported channel counts): forward goldens for the GPU parity tests ----
    net, cfg, sd = ref_model(networks, SMALL)
    cond, guide, x_t = synth_inputs(2, 64, 48, seed=11)
    x6 = torch.from_numpy(np.concatenate([cond, x_t], 1))
    lvl = torch.tensor([[0.0029], [0.6]], dtype=torch.float32)
    eps = net.denoise_fn(x6, lvl, guide=torch.from_numpy(guide))
    np.savez_compressed(os.path.join(OUT, "small_forward.npz"), cond=cond, guide=guide, x_t=x_t,
                        level=lvl.numpy(), eps=eps.numpy().astype(np.float16))

    # (ii) full SID config @1x6x256x256: stats + crop for three noise levels ----------------------
    net, cfg, sd = ref_model(networks, {})
    cond, guide, x_t = synth_inputs(1, 256, 256, seed=21)
    x6 = torch.from_numpy(np.concatenate([cond, x_t], 1))
    rec = {}
    for i, lv in enumerate((0.002865232, 0.239415851, 0.9999995)):
        e = net.denoise_fn(x6, torch.tensor([[lv]], dtype=torch.float32), guide=torch.from_numpy(guide))
        rec[f"eps{i}_stats"] = np.array([e.mean(), e.std(), e.min(), e.max()], dtype=np.float64)
        rec[f"eps{i}_crop"] = e[0, :, 100:132, 60:92].numpy()
        rec[f"eps{i}_ds"] = e[0, :, ::8, ::8].numpy()
    rec["levels"] = np.array((0.002865232, 0.239415851, 0.9999995))
    pred = net.predictor(torch.from_numpy(cond))
    rec["pred_crop"] = pred[0, :, 100:132, 60:92].numpy()
    rec["pred_stats"] = np.array([pred.mean(), pred.std(), pred.min(), pred.max()], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "sid_forward.npz"), **rec)
    extra_full_config(net)
    print("golden written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
