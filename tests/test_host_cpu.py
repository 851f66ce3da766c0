"""CPU tests of the host side: no compute call into the HIP library happens here."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ucdir_oracle as O
from ucdir_amd import config, lib, patch
from ucdir_amd.spec import (SCHEDULE_BUFFERS, UNetConfig, netg_param_shapes, padded_size, unet_layers,
                            unet_param_shapes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)


def test_library_loads_and_exports_every_declared_symbol():
    L = lib.load()
    assert L.ucdir_abi_version() == 1
    hdr = open(os.path.join(ROOT, "include", "ucdir_hip.h")).read()
    declared = set(re.findall(r"\b(ucdir_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ucdir_ctx", "ucdir_config"}
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/ucdir_hip.h but not exported"
    assert declared == set(lib.EXPORTED), declared ^ set(lib.EXPORTED)


def test_library_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    cfg = lib.UcdirConfig()
    cfg.in_channel, cfg.out_channel, cfg.inner_channel, cfg.n_mults = 6, 3, 64, 2
    cfg.channel_mults[0], cfg.channel_mults[1] = 1, 2
    cfg.res_blocks, cfg.image_size = 1, 128
    h = ctypes.c_void_p()
    rc = L.ucdir_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in L.ucdir_last_error()
    from ucdir_amd.ucdir import DY3h
    net = DY3h(inner_channel=64, channel_mults=(1, 2), res_blocks=1, attn_res=(64,))
    with pytest.raises(lib.UcdirError):
        net(torch.zeros(1, 6, 64, 64), torch.zeros(1, 1), torch.zeros(1, 3, 64, 64))


def test_layer_plan_matches_survey():
    L = unet_layers(SID)
    names = [l.name for l in L]
    assert names[:15] == [f"downs.{i}" for i in range(15)] and names[15:17] == ["mid.0", "mid.1"]
    assert [l.name for l in L if l.attn] == ["downs.10", "downs.11", "mid.0", "ups.4", "ups.5", "ups.6"]
    assert [(l.cin, l.cout) for l in L if l.name in ("ups.4", "ups.5", "ups.6")] == [(1024, 512), (1024, 512), (768, 512)]
    shapes = unet_param_shapes(SID)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 97_352_387 or abs(
        sum(int(np.prod(s)) for s in shapes.values()) - 97.35e6) < 0.01e6
    assert len(netg_param_shapes(SID)) + len(SCHEDULE_BUFFERS) == 582
    assert padded_size(256) == 288 and padded_size(288) == 320 and padded_size(384) == 416


def test_product_schedule_equals_golden(golden_dir):
    from ucdir_amd.diffusion import GaussianDiffusion
    g = np.load(os.path.join(golden_dir, "schedule_T50.npz"))
    gd = GaussianDiffusion(torch.nn.Identity(), 128)
    gd.set_new_noise_schedule(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4), torch.device("cpu"))
    for k in SCHEDULE_BUFFERS:
        np.testing.assert_array_equal(getattr(gd, k).numpy(), g[k], err_msg=k)
    np.testing.assert_array_equal(gd.sqrt_alphas_cumprod_prev, g["sqrt_alphas_cumprod_prev"])
    lv, cr, crm1, c1, c2, sg = gd.step_coefficients(49)
    assert abs(lv - 0.002865232) < 1e-9 and abs(cr - 349.0098) < 1e-3 and abs(sg - np.exp(0.5 * -0.9162962)) < 1e-6
    assert gd.step_coefficients(0)[5] == 0.0


def test_config_overrides():
    a = argparse.Namespace(config=os.path.join(ROOT, "config", "sid.yaml"), phase="val", debug=False,
                           checkpoint="ck/I_Elatest", enable_wandb=False)
    o = config.parse(a, make_dirs=False)
    assert o["name"] == "val_sid-ema"
    assert o["model"]["beta_schedule"]["val"]["n_timestep"] == 50
    assert o["model"]["beta_schedule"]["val"]["linear_end"] == 0.4
    assert o["path"]["resume_state"] == "ck/I_Elatest"
    assert o["path"]["experiments_root"].endswith("val_sid-ema_s50")
    assert o["no_such_key"] is None
    assert o["model"]["unet"]["channel_mults"] == [1, 2, 4, 8, 8]


def _toy_net(x, time, guide):
    """Deterministic stand-in denoiser with a 5x5 receptive field (CPU)."""
    w = torch.linspace(-1, 1, 3 * 6 * 25).view(3, 6, 5, 5) / 25
    return F.conv2d(x, w, padding=2) + 0.1 * F.avg_pool2d(guide, 5, stride=1, padding=2) + time.view(-1, 1, 1, 1)


def test_patch_scheduler_matches_reference_semantics():
    torch.manual_seed(0)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    ref = O.patch_forward_guide(x, _toy_net, t, g, skip=96, padding=16)
    got = patch.patch_forward_guide(x, _toy_net, {"time": t, "guide": g}, skip=96, padding=16, max_batch=3)
    assert torch.allclose(got, ref, atol=1e-6)
    assert patch.patch_windows(1680, 2384, 1024, 64) == O.patch_windows(1680, 2384, 1024, 64)
    # image smaller than the window: pd = skip - min + padding (utils/util.py:114-115)
    x2 = torch.randn(2, 6, 40, 70); g2 = torch.randn(2, 3, 40, 70); t2 = torch.tensor([[0.3], [0.7]])
    ref2 = O.patch_forward_guide(x2, _toy_net, t2, g2, skip=64, padding=8)
    got2 = patch.patch_forward_guide(x2, _toy_net, {"time": t2, "guide": g2}, skip=64, padding=8)
    assert torch.allclose(got2, ref2, atol=1e-6)


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    got = patch.patch_forward_guide(x, _toy_net, {"time": t, "guide": g}, skip=96, padding=16, group=dist.group.WORLD)
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_patch_split_sharded_over_ranks_gloo(world):
    """Windows of a step sharded over ranks + one all-gather == single-process result on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + world
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    torch.manual_seed(0)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    ref = O.patch_forward_guide(x, _toy_net, t, g, skip=96, padding=16)
    for r in range(world):
        assert torch.allclose(outs[r], ref, atol=1e-6), r


def test_metrics_restatement():
    from ucdir_amd import metrics as M
    t = torch.tensor([[[-1.2, 0.0], [0.5, 1.0]]]).repeat(3, 1, 1)
    img = M.tensor2img(t)
    assert img.shape == (2, 2, 3) and img.dtype == np.uint8
    assert img[0, 0, 0] == 0 and img[0, 1, 0] == 128 and img[1, 1, 0] == 255 and img[1, 0, 0] == 191
    np.testing.assert_array_equal(img, O.tensor2img(t))
    a = np.random.RandomState(0).randint(0, 255, (32, 32, 3)).astype(np.uint8)
    assert M.calculate_psnr(a, a) == float("inf") and abs(M.calculate_ssim(a, a) - 1.0) < 1e-12
    b = a.copy(); b[0, 0, 0] ^= 8
    assert abs(M.calculate_psnr(a, b) - O.psnr(a, b)) < 1e-12
