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
    assert L.ucdir_abi_version() == lib.ABI_VERSION == 5
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
    q.put((rank, got.numpy()))      # by value: a tensor travels as a shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_patch_split_sharded_over_ranks_gloo(world):
    """Windows of a step sharded over ranks + one all-gather == single-process result on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(world))}
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


# ------------------------------------------------------------------------------------------------------------------
# round 2: sharded patch split wired into the product entry points, rank-identical noise, checkpoint strictness
# ------------------------------------------------------------------------------------------------------------------
def _dy3h_stub(group=None):
    """A product DY3h whose engine call (naiveforward) is replaced by the CPU toy denoiser: DY3h.forward's own dispatch
    (threshold, skip / padding constants, patch_group) is what runs."""
    from ucdir_amd.ucdir import DY3h
    net = DY3h(inner_channel=64, channel_mults=(1, 2), res_blocks=1, attn_res=(64,))
    net.naiveforward = _toy_net
    net.patch_threshold, net.patch_skip, net.patch_padding = 100 * 100, 96, 16
    net.patch_group, net.patch_max_batch = group, 2
    return net


def _dy3h_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ucdir_amd import model as M
    net = _dy3h_stub()
    # DDPM.setup_distributed is what sr.py / create_model run: it must hand the world group to the denoiser and seed the noise
    holder = M.DDPM.__new__(M.DDPM)
    holder.netG = type("G", (), {})()
    holder.netG.denoise_fn = net
    holder.setup_distributed()
    # (the seed itself is installed per image by DDPM.test, only for images the ranks restore together)
    assert net.patch_group is dist.group.WORLD and holder._shared_noise_seed is not None
    torch.manual_seed(0)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    q.put((rank, net(x, t, g).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_dy3h_forward_shards_windows_over_ranks_gloo():
    """model/ucdir.py:298-300 through the PRODUCT dispatch: DY3h.forward on 2 gloo ranks (group set by
    DDPM.setup_distributed) equals the reference's sequential window loop on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_dy3h_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=180) for _ in range(2))}
    for p in procs:
        p.join(60)
    torch.manual_seed(0)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    ref = O.patch_forward_guide(x, _toy_net, t, g, skip=96, padding=16)
    single = _dy3h_stub()(x, t, g)
    assert torch.allclose(single, ref, atol=1e-6)
    for r in range(2):
        assert torch.allclose(outs[r], ref, atol=1e-6), r


def test_patch_geometry_of_the_full_resolution_config():
    """BASELINE configs[2]: 1424x2128 -> DDPM.test +128 -> patch pad +128 -> 1680x2384 -> six 1024^2 windows
    (SURVEY.md §8 a10), two per rank on 3 GPUs, one on 6, at most one on 8."""
    H, W = 1424 + 128, 2128 + 128
    pd = patch.patch_pad(H, W, 1024, 64)
    assert pd == 64
    wins = patch.patch_windows(H + 2 * pd, W + 2 * pd, 1024, 64)
    assert wins == [(0, 1024, 0, 1024), (0, 1024, 896, 1920), (0, 1024, 1360, 2384),
                    (656, 1680, 0, 1024), (656, 1680, 896, 1920), (656, 1680, 1360, 2384)]
    # GoPro 720x1280: min < skip -> pd = 1024 - 720 + 64 (utils/util.py:114-115)
    assert patch.patch_pad(720, 1280, 1024, 64) == 368


def test_constant_batch_chunks():
    """Windows are cut into equal chunks (last one padded): the engine sees one batch size per step."""
    seen = []

    def net(x, time, guide):
        seen.append(x.shape[0])
        return _toy_net(x, time, guide)
    torch.manual_seed(1)
    x = torch.randn(1, 6, 150, 210); g = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    ref = O.patch_forward_guide(x, _toy_net, t, g, skip=96, padding=16)       # 12 windows
    got = patch.patch_forward_guide(x, net, {"time": t, "guide": g}, skip=96, padding=16, max_batch=5)
    assert torch.allclose(got, ref, atol=1e-6)
    assert len(set(seen)) == 1 and seen[0] <= 5, seen


def test_seeded_noise_is_identical_across_instances():
    from ucdir_amd.diffusion import GaussianDiffusion
    a, b = GaussianDiffusion(torch.nn.Identity(), 128), GaussianDiffusion(torch.nn.Identity(), 128)
    like = torch.zeros(1, 3, 8, 8)
    draws = []
    for gd in (a, b):
        gd.noise_seed = 77
        gd._start_noise(torch.device("cpu"))
        draws.append([gd._noise(like, k) for k in range(3)])
    for u, v in zip(*draws):
        assert torch.equal(u, v)
    assert not torch.equal(draws[0][0], draws[0][1])
    a._start_noise(torch.device("cpu"))                      # re-seeded at the start of every loop
    assert torch.equal(a._noise(like, 0), draws[0][0])


def test_sampler_releases_the_weight_check_hold_when_its_setup_raises():
    """Round-4 advice: _begin() sets the denoiser's hold flag; every sampler enters its try / finally at once, so a failure in
    the setup behind it (noise source, buffer allocation) still runs _end(): the hold is released and the patch cache cleared."""
    from ucdir_amd.diffusion import GaussianDiffusion

    class Den(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.hold, self.cleared, self.patch_threshold = False, 0, 1 << 30
        def hold_weight_check(self, on): self.hold = bool(on)
        def clear_patch_cache(self): self.cleared += 1

    gd = GaussianDiffusion(Den(), 128)
    gd.set_new_noise_schedule(dict(schedule="linear", n_timestep=8, linear_start=1e-6, linear_end=0.4), torch.device("cpu"))

    def boom(*a, **k):
        raise RuntimeError("noise source failed")
    gd._start_noise = boom
    x = torch.zeros(1, 3, 8, 8)
    for call in (lambda: gd.p_sample_loop(x), lambda: gd.ddim_sample(x), lambda: gd.dpm_solver_sample(x)):
        n = gd.denoise_fn.cleared
        with pytest.raises(RuntimeError, match="noise source failed"):
            call()
        assert gd.denoise_fn.hold is False and gd.denoise_fn.cleared == n + 1


def test_patch_window_list_is_validated_before_upload():
    from ucdir_amd import patch as P
    wins = P.patch_windows(1680, 2384, 1024, 64)
    P._check_windows(wins, 1680, 2384, 1024)                               # the scheduler's own list passes
    for bad in [(0, 1024, 1400, 2424), (-8, 1016, 0, 1024), (0, 1000, 0, 1024), (700, 1724, 0, 1024)]:
        with pytest.raises(ValueError, match="patch window"):
            P._check_windows([bad], 1680, 2384, 1024)


def test_config_overrides_gopro_and_jpeg(tmp_path):
    import yaml
    base = yaml.safe_load(open(os.path.join(ROOT, "config", "sid.yaml")))
    for name, suffix, has_factor in (("gop-x", "full", False), ("jpg-q10", "fullimage10", True), ("other", "", False)):
        cfg = dict(base); cfg["name"] = name
        p = tmp_path / f"{name}.yaml"
        yaml.safe_dump(cfg, open(p, "w"))
        a = argparse.Namespace(config=str(p), phase="val", debug=False, checkpoint=None, enable_wandb=False)
        o = config.parse(a, make_dirs=False)
        T = o["model"]["beta_schedule"]["val"]["n_timestep"]
        assert o["path"]["experiments_root"].endswith(f"_s{T}{suffix}"), o["path"]["experiments_root"]
        da = o["datasets"]["val"]["data_args"]
        if name != "other":
            assert T == 50 and o["model"]["beta_schedule"]["val"]["linear_end"] == 0.4
        assert (da["factor"] == [10, 10] and da["crop_size"] == -1) if has_factor else True
        if name == "gop-x":
            assert da["dataroot"]["lq"].endswith("GoPro/input/")


def test_checkpoint_loading_is_strict_about_network_keys():
    from ucdir_amd import model as M
    from ucdir_amd import networks
    opt = {"model": {"which_model_G": "ucdir", "unet_name": "DY3h", "diffusion_name": "ResiGaussianGuideDY",
                     "unet": dict(in_channel=6, out_channel=3, inner_channel=64, channel_mults=[1, 2], attn_res=[64],
                                  res_blocks=1, dropout=0, norm_groups=1),
                     "diffusion": dict(image_size=128, channels=3, conditional=True)}}
    net = networks.define_G(opt)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["betas"] = torch.zeros(2000)                                     # training-length buffer: skipped, not an error
    rep = M.load_checkpoint_state(net, {"module." + k: v for k, v in sd.items()})     # DDP prefix stripped
    assert rep["loaded"] == len(sd) - 1 and rep["skipped_buffers"] == ["betas"]
    k0 = next(k for k in sd if k.startswith("denoise_fn.") and k.endswith("conv1.weight"))
    with pytest.raises(RuntimeError, match="missing"):
        M.load_checkpoint_state(net, {k: v for k, v in sd.items() if k != k0})
    renamed = dict(sd); renamed["denoise_fn.renamed.weight"] = renamed.pop(k0)
    with pytest.raises(RuntimeError):
        M.load_checkpoint_state(net, renamed)
    bad = dict(sd); bad[k0] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="wrong shape"):
        M.load_checkpoint_state(net, bad)
    extra = dict(sd); extra["optimizer_step"] = torch.zeros(1)
    M.load_checkpoint_state(net, extra)                                  # EMA path: unrelated keys are logged and ignored
    with pytest.raises(RuntimeError):
        M.load_checkpoint_state(net, extra, strict=True)                 # non-EMA path (strict = not finetune_norm)
    assert net.denoise_fn._wdirty                                        # a load marks the engine's packed weights stale


def test_device_side_uint8_conversion_matches_tensor2img():
    from ucdir_amd import metrics as M
    t = torch.randn(3, 17, 23, generator=torch.Generator().manual_seed(0)) * 0.8
    t[0, 0, 0], t[1, 0, 0], t[2, 0, 0] = -1.0 + 1 / 255, 0.0, 1.0 - 1 / 255      # exact .5 cases round half to even
    np.testing.assert_array_equal(M.tensor2img_u8_device(t), M.tensor2img(t))
    np.testing.assert_array_equal(M.tensor2img_u8_device(t.unsqueeze(0)), O.tensor2img(t))


def test_launch_plan_split_factors():
    """Host-side launch planning of the engine (no device needed): the K-split of conv3x3_halo follows the cost model fitted to
    the B = 1 traces - deep levels split 4-16 ways, the 144^2 level (10 MB of partial sums per split) and full grids never -
    and the 64-per-group AKGM kernel spreads its units by whole rounds of 512 resident workgroups."""
    from ucdir_amd import lib as ulib
    L = ulib.load()
    plan = L.ucdir_debug_launch_plan
    assert plan(b"ksplit", 16, 32, 3, 324 * 512.0) == 16              # B = 1, 18^2, 1024 -> 512: 16 workgroups x 96 K steps
    assert 2 <= plan(b"ksplit", 100, 16, 3, 72 * 72 * 256.0) <= 5     # B = 1, 72^2, 512 -> 256
    assert plan(b"ksplit", 162, 12, 3, 144 * 144 * 128.0) == 1        # B = 1, 144^2: the finish pass would cost what the split saves
    assert plan(b"ksplit", 256, 16, 3, 16 * 324 * 512.0) == 1         # B = 16, 18^2: one workgroup per CU already
    assert plan(b"ksplit", 1296, 12, 5, 16 * 144 * 144 * 128.0) == 1  # full grids are never split
    for wgs, nch in ((16, 32), (48, 24), (100, 8), (256, 32)):
        ks = plan(b"ksplit", wgs, nch, 3, 1e5)
        assert ks >= 1 and wgs * ks <= 512 and nch // ks >= 2 or ks == 1
    assert [plan(b"usplit", n, 0, 0, 0.0) for n in (768, 256, 48, 16, 3072)] == [2, 2, 4, 4, 1]
    assert plan(b"nonsense", 1, 1, 1, 0.0) == -1


# ------------------------------------------------------------------------------------------------------------------
# round 3: per-module patch cache, per-image rank-identical noise, in-place weight updates are noticed
# ------------------------------------------------------------------------------------------------------------------
def test_patch_cache_is_per_module_and_cleared():
    """Two DY3h modules alternate on different guides without evicting each other's padded windows; the work buffers
    (denoised canvas) persist across steps and clear_patch_cache() releases everything."""
    a, b = _dy3h_stub(), _dy3h_stub()
    torch.manual_seed(2)
    x = torch.randn(1, 6, 150, 210); ga = torch.randn(1, 3, 150, 210); gb = torch.randn(1, 3, 150, 210); t = torch.tensor([[0.3]])
    ra = O.patch_forward_guide(x, _toy_net, t, ga, skip=96, padding=16)
    rb = O.patch_forward_guide(x, _toy_net, t, gb, skip=96, padding=16)
    for _ in range(2):
        assert torch.allclose(a(x, t, ga), ra, atol=1e-6) and torch.allclose(b(x, t, gb), rb, atol=1e-6)
    ka, kb = a._patch_cache["guide"], b._patch_cache["guide"]
    assert ka[2] is ga and kb[2] is gb and a._patch_cache is not b._patch_cache
    den = a._patch_cache["den"][1]
    a(x, t, ga)
    assert a._patch_cache["den"][1] is den and a._patch_cache["guide"] is ka        # reused, not re-cut / re-allocated
    # two forwards of the same shape must not alias (round-3 advice): the result is a fresh tensor, not a view into `den`
    x2 = torch.randn(1, 6, 150, 210)
    e1 = a(x, t, ga); e1_copy = e1.clone()
    e2 = a(x2, t, ga)
    assert torch.equal(e1, e1_copy) and not torch.equal(e1, e2) and e1.data_ptr() != e2.data_ptr()
    a.clear_patch_cache()
    assert a._patch_cache == {} and "guide" in b._patch_cache
    from ucdir_amd import patch as P
    assert not hasattr(P, "_GUIDE_WINDOWS")


def test_noise_seed_is_per_image_and_only_for_sharded_images():
    """DDPM.test installs the rank-identical generator only for images the ranks restore together, offset by the image
    index: small (strided) images keep independent default noise, two sharded images do not share their noise."""
    from ucdir_amd import model as M
    from ucdir_amd.diffusion import GaussianDiffusion
    calls = []

    class G:
        noise_seed = None
        noise_index = 0

        def __init__(self):
            self.denoise_fn = type("D", (), {"patch_group": object(), "patch_threshold": 200 * 200})()

        def eval(self):
            pass

        def super_resolution(self, sr, continous):
            calls.append((self.noise_seed, self.noise_index, tuple(sr.shape[-2:])))
            return sr

    h = M.DDPM.__new__(M.DDPM)
    h.netG = G(); h._shared_noise_seed = 77
    for idx, size in ((3, 100), (4, 66), (5, 100)):
        h.data = {"SR": torch.zeros(1, 3, size, size), "Index": torch.tensor([idx])}
        h.test()
    assert calls == [(77, 3, (228, 228)), (None, 4, (194, 194)), (77, 5, (228, 228))]
    # the generator seed really differs per image and is reproducible
    gd = GaussianDiffusion.__new__(GaussianDiffusion)
    gd.noise_seed, gd.noise_source = 77, None
    seeds = []
    for idx in (3, 5, 3):
        gd.noise_index = idx
        gd._start_noise(torch.device("cpu"))
        seeds.append(gd._gen.initial_seed())
    assert seeds[0] == seeds[2] != seeds[1]


def test_in_place_weight_updates_change_the_signature():
    """optimizer.step / p.data.copy_ / nn.init do not pass through load_state_dict or _apply: the signature DY3h and
    UNetSeeInDark compare once per image (prepare_guide / forward) must change."""
    from ucdir_amd.ucdir import DY3h, UNetSeeInDark
    for net in (DY3h(inner_channel=64, channel_mults=(1, 2), res_blocks=1, attn_res=(64,)), UNetSeeInDark()):
        s0 = net._weights_signature()
        assert s0 == net._weights_signature()
        p = list(net.parameters())[7]
        with torch.no_grad():
            p.mul_(1.0001)
        s1 = net._weights_signature()
        assert s1 != s0
        with torch.no_grad():
            list(net.parameters())[3].data.copy_(torch.zeros_like(list(net.parameters())[3]))
        s2 = net._weights_signature()
        assert s2 != s1
        torch.nn.init.normal_(list(net.parameters())[0])
        assert net._weights_signature() != s2


def test_graft_entry_build():
    """The driver's "does it build" entry point runs here without a GPU (and does not carry a stale ABI number)."""
    import importlib
    import __graft_entry__ as g
    importlib.reload(g)
    g.build()
