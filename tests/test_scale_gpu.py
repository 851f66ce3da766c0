"""GPU parity tests for what ships at scale (``pytest -m gpu``): flash attention at the token counts of the real
levels and of the 1024^2 patch windows, the 50-step sampler at the full SID configuration, ``DDPM.test``, one real
patch window, BASELINE configs[3] (B = 32, T = 100) and configs[4] (JPEG: skip 256 / pad 32, fp16 attention), HIP-graph
replay, and the ``sr.py -p val`` outputs against the oracle's images.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PSNR_50 = 45.57       # 50-step restoration vs oracle, measured on MI355X (see test_sampler_50_steps_full_sid_config)

import hip_checks as C  # noqa: E402
from oracle import ucdir_oracle as O  # noqa: E402
from ucdir_amd.spec import UNetConfig  # noqa: E402
from ucdir_amd.weights import synth_inputs  # noqa: E402

SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)
FWD_TOL = 1.7e-2      # see tests/test_hip_gpu.py
PATCH_TOL = 2.0e-2     # nine separately normalised 256^2 windows: measured 1.51e-2 (single 256^2 forward: 1.1e-2)
ATT_TOL = 1.2e-2
SCHED50 = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)


@pytest.fixture(scope="module")
def sid_net():
    return C.build_net(SID)


# ---- attention ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 128, 12, 10), (2, 256, 20, 13), (1, 384, 9, 30), (1, 512, 18, 18), (2, 512, 36, 36),
                                   (1, 512, 64, 64), (1, 512, 128, 128)],
                         ids=["C128_N120", "C256_N260", "C384_N270", "N324", "N1296", "N4096", "N16384"])
def test_flash_attention_token_counts(shape):
    """model/ucdir.py:165-182 at the real token counts: 18^2 / 36^2 (256^2 crops), 64^2 / 128^2 (1024^2 windows), plus
    ragged N (not a multiple of the 64-key / 128-query tiles) and every channel count the kernel is instantiated for."""
    m = C.attention_case(*shape)
    assert not m["nan"] and m["rel_rms_branch"] < ATT_TOL, m


@pytest.mark.parametrize("shape", [(2, 128, 12, 10), (1, 512, 36, 36), (1, 512, 64, 64)], ids=["C128_N120", "N1296", "N4096"])
def test_flash_attention_fp16_operands(shape):
    """BASELINE configs[4]: q, k, v', P as IEEE half on v_mfma_*_f16 (10 mantissa bits against bf16's 7: the bound is the
    bf16 one, the measured error is lower)."""
    m = C.attention_case(*shape, fp16=True)
    assert not m["nan"] and m["rel_rms_branch"] < ATT_TOL, m
    mb = C.attention_case(*shape, fp16=False)
    assert m["rel_rms_branch"] < 1.05 * mb["rel_rms_branch"] + 1e-4, (m, mb)


def test_flash_attention_online_softmax_rescale_is_exercised():
    """The running-max rescale branch is data dependent (cdna guide §5.4 rule 26): a few tokens with a large norm make
    the row maxima jump at late KV tiles (tokens 700, 1100, 1290 sit in tiles 10, 17 and 20 of 21), for their own rows and
    for every row correlated with them.  The flash kernel is compared on the FULL tensor with the materialised-score path
    (QK^T / softmax / PV launches), which consumes the very same bf16 q, k, v' tensor: the two differ only in the
    summation order and in where P is rounded, so they must agree far more tightly than either agrees with the fp32 oracle
    (with logits of several hundred, the bf16 rounding of q and k alone moves near-tied softmax weights)."""
    L = C.ulib.load()
    g = C.rng(5)
    B, Cc, H, W = 1, 512, 36, 36
    x = torch.randn(B, Cc, H, W, generator=g) * 0.7
    flat = x.view(B, Cc, -1)
    for tok, s in ((700, 4.0), (1100, 6.0), (1290, 8.0), (3, 3.0)):
        flat[:, :, tok] *= s
    x = C.bfr(x)
    sd = {"a.norm.weight": 1 + 0.25 * torch.randn(Cc, generator=g), "a.norm.bias": 0.2 * torch.randn(Cc, generator=g),
          "a.qkv.weight": torch.randn(3 * Cc, Cc, 1, 1, generator=g) * (3.0 / Cc) ** 0.5,
          "a.out.weight": torch.randn(Cc, Cc, 1, 1, generator=g) * (1.5 / Cc) ** 0.5,
          "a.out.bias": torch.randn(Cc, generator=g) * 0.1}
    y = O.self_attention(sd, "a.", x)
    n = lambda k: sd[k].numpy().copy()
    dx = x.cuda()
    outs = {}
    try:
        for flash in (1, 0):
            C.ulib.check(L.ucdir_debug_flag(b"flash", flash))
            dy = torch.empty(B, Cc, H, W, device="cuda")
            C.ulib.check(L.ucdir_op_attention(C._p(dx), B, Cc, H, W, C._hp(n("a.norm.weight")), C._hp(n("a.norm.bias")),
                                              C._hp(n("a.qkv.weight")), C._hp(n("a.out.weight")), C._hp(n("a.out.bias")), 0,
                                              C._p(dy), C._st()))
            torch.cuda.synchronize()
            outs[flash] = dy.cpu()
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"flash", -1))
    branch = (y - x).pow(2).mean().sqrt()
    assert bool(torch.isfinite(outs[1]).all())
    rel_paths = float((outs[1] - outs[0]).pow(2).mean().sqrt() / branch)
    assert rel_paths < 4e-3, rel_paths                    # same operands: only P rounding position / summation order differ
    for tok in (700, 1100, 1290):                         # the spiked rows themselves, not hidden in the average
        e = (outs[1] - outs[0]).view(B, Cc, -1)[:, :, tok]
        r = (y - x).view(B, Cc, -1)[:, :, tok]
        assert float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()) < 8e-3, tok
    rel_oracle = float((outs[1] - y).pow(2).mean().sqrt() / branch)
    assert rel_oracle < 6e-2, rel_oracle                  # sanity against fp32 (bf16 q, k at |logit| >> 1)


# ---- the sampler at the configuration that ships -------------------------------------------------------------------------
def test_sampler_50_steps_full_sid_config(sid_net):
    """north_star: 'the 50-step sampler reproduces ...': full SID configuration, B = 1, 256^2, T = 50, injected noise,
    HIP path vs the CPU oracle (model/diffusion.py:185-211, 473-478).  uint8 PSNR bound for bf16 from SURVEY.md §8c."""
    m = C.sampler_case(SID, 256, 256, 50, seed=6, net_sd=sid_net)
    print("50-step restoration, full SID configuration, B = 1, 256^2: uint8 PSNR vs oracle = %.2f dB, rel-RMS %.3e" % (m["psnr_u8"], m["rel_rms"]))
    assert not m["nan"] and m["psnr_u8"] > 35.0, m
    # measured on MI355X: PSNR_50 dB; a drift of more than 3 dB below it is a regression even above the 35 dB bound
    assert m["psnr_u8"] > PSNR_50 - 3.0, m


def test_ddpm_test_matches_oracle():
    """DDPM.test caller semantics (model/model.py:124-138): reflect-pad 64, super_resolution(continous=True), crop."""
    from ucdir_amd import model as M
    net, sd = C.build_net(SMALL)
    T = 8
    sched = dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=0.4)
    tab = O.schedule_tables(sched)
    net.set_new_noise_schedule(sched, torch.device("cuda"))
    cond = torch.from_numpy(synth_inputs(1, 72, 88, seed=12)[0])
    g = C.rng(112)
    noises = [torch.randn(1, 3, 72 + 128, 88 + 128, generator=g) for _ in range(T)]
    ref = O.ddpm_test(sd, tab, cond, noises, continous=True)
    ddpm = M.DDPM.__new__(M.DDPM)
    ddpm.netG, ddpm.device = net, torch.device("cuda")
    ddpm.feed_data({"SR": cond, "HR": cond, "Index": 0})
    net.noise_source = lambda shape, device, k: noises[k].to(device)
    try:
        ddpm.test(continous=True)
    finally:
        net.noise_source = None
    got = ddpm.SR.cpu()
    assert got.shape == ref.shape == (1 + T, 3, 72, 88), (got.shape, ref.shape)
    assert torch.allclose(got[0], ref[0], atol=0.1)                       # ret_img[0] = the input + initx (diffusion.py:478)
    assert O.psnr(O.tensor2img(got[-1]), O.tensor2img(ref[-1])) > 35.0
    vis = ddpm.visuals_u8()
    np.testing.assert_array_equal(vis["SR"], O.tensor2img(got[-1]))        # device-side uint8 == host-side conversion


def test_one_real_patch_window_vs_oracle(sid_net):
    """One window of the reference's inter-step split at its real geometry (skip 1024, padding 64: the windows are
    1024^2 naiveforward calls, utils/util.py:124-139): attention at N = 16384 / 4096, conv tiles at 1024 .. 64."""
    net, sd = sid_net
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(1, 1024, 1024, seed=31))
    lvl = torch.tensor([[0.35]])
    x6 = torch.cat([cond, x_t], 1)
    ref = O.dy3h_naive_forward(sd, x6, lvl, guide)
    with torch.no_grad():
        got = net.denoise_fn.naiveforward(x6.cuda(), lvl.cuda(), guide.cuda())
    torch.cuda.synchronize()
    m = C.metrics(got, ref)
    print("1024^2 window forward:", m)
    assert not m["nan"] and m["rel_rms"] < FWD_TOL, m
    ws = C.ulib.load().ucdir_workspace_bytes(net.denoise_fn._handle())
    print("workspace bytes for one 1024^2 window:", ws)
    assert ws < 5.5e9, ws                      # one window incl. 195 MB weights (no S / P score tensors: they alone were 1.5 GB)


def test_gopro_config_b32_t100(sid_net):
    """BASELINE configs[3]: B = 32, 256^2, T = 100 (second restoration task).  Full size: finite, reproducible with the
    rank-identical seeded generator, samples differ; the T = 100 schedule drives the first three steps exactly like the
    oracle (the tables themselves are pinned bit-exactly by tests/golden/schedule_T100.npz on the CPU side)."""
    net, sd = sid_net
    sched = dict(schedule="linear", n_timestep=100, linear_start=1e-6, linear_end=0.4)
    dev = torch.device("cuda")
    net.set_new_noise_schedule(sched, dev)
    tab = O.schedule_tables(sched)
    assert net.num_timesteps == 100
    cond = torch.from_numpy(synth_inputs(32, 256, 256, seed=13)[0]).to(dev)
    net.noise_seed = 99
    try:
        with torch.no_grad():
            a = net.super_resolution(cond, False).clone()
            b = net.super_resolution(cond, False)
    finally:
        net.noise_seed = None
    assert a.shape == (32, 3, 256, 256) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b)
    assert float((a[0] - a[1]).abs().max()) > 1e-3
    # three steps of the T = 100 chain (t = 99, 98, 97) for one sample against the oracle
    c1 = cond[:1].cpu()
    gi = C.rng(7)
    nz = [torch.randn(1, 3, 256, 256, generator=gi) for _ in range(4)]
    guide = O.predictor_forward(sd, c1)
    img = nz[0]
    for j, t in enumerate((99, 98, 97)):
        eps = O.dy3h_forward(sd, torch.cat([c1, img], 1), O.noise_level_for(tab, t, 1), guide)
        img = O.p_sample_step(tab, img, eps, t, nz[j + 1])
    with torch.no_grad():
        g = net.predictor(c1.to(dev))
        x = nz[0].to(dev)
        for j, t in enumerate((99, 98, 97)):
            net.noise_source = lambda shape, device, k, _j=j: nz[_j + 1].to(device)
            x = net.p_sample(x, t, condition_x=c1.to(dev), kwargs={"guide": g}, _k=0)
        net.noise_source = None
    m = C.metrics(x, img)
    assert m["rel_rms"] < 5e-3, m             # x_t is dominated by the (identical) noise; eps enters with small weights


def test_jpeg_config_patch_split_fp16_attention():
    """BASELINE configs[4]: 512^2 image, inter-step patch split with explicit skip 256 / padding 32 (nine 256^2 windows
    per step; the reference's own threshold would not split 512^2), fp16 attention operands; one denoiser call through
    DY3h.forward vs the oracle's sequential window loop (utils/util.py:108-146)."""
    from ucdir_amd import networks
    from ucdir_amd.weights import synth_state_dict
    import bench
    opt = bench.sid_opt()
    opt["model"]["unet"]["attn_dtype"] = "fp16"
    net = networks.define_G(opt)
    np_sd = synth_state_dict(net.denoise_fn.cfg, 0)
    from ucdir_amd import model as M
    M.load_checkpoint_state(net, {k: torch.from_numpy(v) for k, v in np_sd.items()}, strict=True)
    net = net.cuda().eval()
    sd = O.to_torch_sd(np_sd)
    dn = net.denoise_fn
    dn.patch_threshold, dn.patch_skip, dn.patch_padding, dn.patch_max_batch = 0, 256, 32, 9
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(1, 512, 512, seed=41))
    lvl = torch.tensor([[0.6]])
    x6 = torch.cat([cond, x_t], 1)
    from ucdir_amd import patch
    assert len(patch.patch_windows(576, 576, 256, 32)) == 9
    ref = O.dy3h_forward(sd, x6, lvl, guide, patch_threshold=0, skip=256, padding=32)
    with torch.no_grad():
        got = dn(x6.cuda(), lvl.cuda(), guide.cuda())
    m = C.metrics(got, ref)
    print("configs[4] forward:", m)
    assert not m["nan"] and m["rel_rms"] < PATCH_TOL, m


# ---- B = 1 latency path -----------------------------------------------------------------------------------------------------
def test_graph_replay_is_bit_identical(sid_net):
    """ucdir_set_graph: a forward replayed from the captured HIP graph equals the eager launch sequence bit for bit, across
    changing inputs in the same buffers and across a restoration (persistent buffers in p_sample_loop)."""
    net, sd = sid_net
    dn = net.denoise_fn
    dev = torch.device("cuda")
    cond, guide, x_t = (torch.from_numpy(a).to(dev) for a in synth_inputs(1, 256, 256, seed=51))
    lvl = torch.full((1, 1), 0.3, device=dev)
    eps = torch.empty_like(x_t)
    with torch.no_grad():
        e0 = dn.forward_split(cond, x_t, lvl, guide).clone()
        dn.set_graph(True)
        try:
            g1 = dn.forward_split(cond, x_t, lvl, guide, out=eps).clone()          # captured + launched
            g2 = dn.forward_split(cond, x_t, lvl, guide, out=eps).clone()          # replayed
            x_t.mul_(0.5); lvl.fill_(0.7)
            g3 = dn.forward_split(cond, x_t, lvl, guide, out=eps).clone()          # same pointers, new contents
            dn.set_graph(False)
            e3 = dn.forward_split(cond, x_t, lvl, guide).clone()
            # a whole restoration with graphs on == graphs off
            net.set_new_noise_schedule(dict(schedule="linear", n_timestep=6, linear_start=1e-6, linear_end=0.4), dev)
            net.noise_seed = 5
            r_eager = net.super_resolution(cond, False).clone()
            dn.set_graph(True)
            r_graph = net.super_resolution(cond, False).clone()
            r_graph2 = net.super_resolution(cond * 0.5, False).clone()              # second image: same buffers, replays
            dn.set_graph(False)
            r_eager2 = net.super_resolution(cond * 0.5, False).clone()
        finally:
            dn.set_graph(False)
            net.noise_seed = None
    assert torch.equal(e0, g1) and torch.equal(g1, g2)
    assert torch.equal(e3, g3) and not torch.equal(g1, g3)
    assert torch.equal(r_eager, r_graph) and torch.equal(r_eager2, r_graph2)


def test_forward_rejects_mismatched_and_host_tensors(sid_net):
    """a guide of another shape, a CPU tensor, or a wrong batch must raise, never reach the kernels."""
    from ucdir_amd import lib
    net, sd = sid_net
    dn = net.denoise_fn
    dev = torch.device("cuda")
    cond, guide, x_t = (torch.from_numpy(a).to(dev) for a in synth_inputs(2, 64, 64, seed=3))
    lvl = torch.full((2, 1), 0.3, device=dev)
    with pytest.raises(ValueError):
        dn.forward_split(cond, x_t, lvl, guide[:1])
    with pytest.raises(ValueError):
        dn.forward_split(cond, x_t[..., :32], lvl, guide)
    with pytest.raises(lib.UcdirError):
        dn.forward_split(cond, x_t, lvl.cpu(), guide)
    with pytest.raises(lib.UcdirError):
        dn.forward_split(cond.cpu(), x_t, lvl, guide)
    with pytest.raises(ValueError):
        dn.forward_split(cond, x_t, lvl[:1], guide)
    # the C ABI itself refuses a shape that is not the planned one
    L = lib.load()
    dn.forward_split(cond, x_t, lvl, guide)
    eps = torch.empty_like(x_t)
    rc = L.ucdir_unet_forward(dn._handle(), C._p(cond), C._p(x_t), C._p(lvl), C._p(eps), 3, 64, 64, C._st())
    assert rc != 0 and b"does not match" in L.ucdir_last_error()


# ---- sr.py -p val outputs vs the oracle -----------------------------------------------------------------------------------------
def test_sr_val_outputs_match_oracle(tmp_path, monkeypatch):
    """f2: the image `sr.py -p val` writes and the PSNR it logs, against the oracle's ddpm_test -> tensor2img on the same
    image with the same injected noise (reference: sr.py:518-575, core/metrics.py:14-55)."""
    import importlib.util
    import yaml
    from PIL import Image
    from ucdir_amd import model as M
    from ucdir_amd.weights import synth_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(1)
    for d in ("lq", "gt"):
        os.makedirs(tmp_path / d)
    gt = (rs.rand(72, 88, 3) * 255).astype(np.uint8)
    lq = (gt * 0.25).astype(np.uint8)
    Image.fromarray(gt).save(tmp_path / "gt" / "000.png")
    Image.fromarray(lq).save(tmp_path / "lq" / "000.png")
    cfg = yaml.safe_load(open(os.path.join(root, "config", "sid.yaml")))
    cfg["datasets"]["val"]["data_args"]["dataroot"] = {"lq": str(tmp_path / "lq"), "gt": str(tmp_path / "gt")}
    cfg["model"]["unet"].update(channel_mults=[1, 2, 4], res_blocks=1, attn_res=[32])
    yaml.safe_dump(cfg, open(tmp_path / "sid_small.yaml", "w"))
    monkeypatch.chdir(tmp_path)
    T = 50                                              # forced by the 'sid' name override (core/logger.py:58-61)
    g = C.rng(77)
    noises = [torch.randn(1, 3, 72 + 128, 88 + 128, generator=g) for _ in range(T)]
    real_create = M.create_model

    def create(opt, device=None):
        m = real_create(opt, device)
        m.netG.noise_source = lambda shape, device, k: noises[k].to(device)
        return m
    monkeypatch.setattr(M, "create_model", create)
    spec = importlib.util.spec_from_file_location("sr_entry2", os.path.join(root, "sr.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    psnr, ssim = sr.main(["-p", "val", "-c", str(tmp_path / "sid_small.yaml"), "--synthetic-weights"])
    # oracle
    sd = O.to_torch_sd(synth_state_dict(SMALL, 0))
    tab = O.schedule_tables(SCHED50)
    x = torch.from_numpy(lq.astype(np.float32).transpose(2, 0, 1) / 255.0)[None] * 2 - 1
    ref = O.ddpm_test(sd, tab, x, noises, continous=True)
    ref_img = O.tensor2img(ref[-1])
    ref_psnr = O.psnr(ref_img, gt)
    out = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path / "experiments") for f in fs if f.endswith("_sr.jpg")]
    assert len(out) == 1
    saved = np.asarray(Image.open(out[0]).convert("RGB"))
    assert saved.shape == ref_img.shape
    assert O.psnr(saved, ref_img) > 33.0                  # build vs oracle through a quality-100 JPEG
    assert abs(psnr - ref_psnr) < 0.25, (psnr, ref_psnr)   # the logged metric is the oracle's within bf16 noise


# ---- BASELINE configs[2] at its real size ----------------------------------------------------------------------------------------
def test_full_resolution_patch_batch_equals_single_windows(sid_net):
    """configs[2]: a 1424 x 2128 image (DDPM.test pads it to 1552 x 2256) through DY3h.forward - six 1024^2 windows as ONE
    engine batch (ActPlanner buffer recycling at B = 6, flash attention at N = 16384) - equals six single-window naiveforward
    calls pasted in the reference's order (utils/util.py:119-145); finite; workspace bounded."""
    from ucdir_amd import patch as P
    net, sd = sid_net
    dn = net.denoise_fn
    H, W = 1424 + 128, 2128 + 128
    g = torch.Generator().manual_seed(41)
    x6 = (torch.rand(1, 6, H, W, generator=g) * 2 - 1).cuda()
    guide = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).cuda()
    lvl = torch.tensor([[0.35]]).cuda()
    assert H * W > dn.patch_threshold and dn.patch_group is None
    with torch.no_grad():
        got = dn(x6, lvl, guide).clone()
    torch.cuda.synchronize()
    ws = C.ulib.load().ucdir_workspace_bytes(dn._handle())
    print("workspace bytes for the six-window batch:", ws)
    assert got.shape == (1, 3, H, W) and bool(torch.isfinite(got).all())
    assert ws < 16e9, ws                                   # 14.0 GB measured in round 2 (2.3 GB per window + weights)
    dn.clear_patch_cache()
    # the reference's sequential loop with the product's single-window call
    pd = P.patch_pad(H, W, 1024, 64)
    wins = P.patch_windows(H + 2 * pd, W + 2 * pd, 1024, 64)
    assert len(wins) == 6
    xp = F.pad(x6, (pd, pd, pd, pd), mode="reflect")
    gp = F.pad(guide, (pd, pd, pd, pd), mode="reflect")
    den = torch.zeros_like(xp)[:, :3]
    with torch.no_grad():
        for (a, b, c, d) in wins:
            o = dn.naiveforward(xp[..., a:b, c:d].contiguous(), lvl, gp[..., a:b, c:d].contiguous())
            den[..., a + 64:b - 64, c + 64:d - 64] = o[..., 64:-64, 64:-64]
    ref = den[..., pd:-pd, pd:-pd]
    m = C.metrics(got, ref)
    print("six-window batch vs six single-window launches:", m)
    # Batch 6 vs batch 1 changes grid-size-dependent launch choices (128- vs 64-row tiles and split-K at the 64^2 / 128^2 levels of
    # a window): another fp32 summation order = another realisation of the bf16 rounding noise, which this random-weight network
    # amplifies to the size of the build-vs-oracle error itself (9.4e-3 measured here; cf. test_forward_bit_reproducible_at_bench_size).
    # A wrong window order, paste offset or recycled-buffer corruption would be O(1).
    assert m["rel_rms"] < FWD_TOL, m
    dn.clear_patch_cache()


def _nccl_one_rank_worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    net, sd = C.build_net(SMALL)
    dn = net.denoise_fn
    dn.patch_threshold, dn.patch_skip, dn.patch_padding = 0, 128, 32
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=4, linear_start=1e-6, linear_end=0.4), torch.device("cuda"))
    cond = torch.from_numpy(synth_inputs(1, 160, 200, seed=5)[0]).cuda()
    net.noise_seed = 3
    with torch.no_grad():
        plain = net.super_resolution(cond, False).clone()
    dn.patch_group, dn.patch_force_gather, dn.patch_timers = dist.group.WORLD, True, []
    with torch.no_grad():
        gathered = net.super_resolution(cond, False)
    torch.cuda.synchronize()
    n_gather = len(dn.patch_timers)
    ms = sum(a.elapsed_time(b) for a, b in dn.patch_timers)
    q.put((bool(torch.equal(plain, gathered)), n_gather, ms))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_all_gather_branch_on_one_gpu():
    """The sharded patch split's collective on real hardware without a second GPU: a one-rank RCCL group with the gather
    branch forced (DY3h.patch_force_gather): RCCL initialises, all_gather_into_tensor runs on device tensors once per
    denoising step into the pre-allocated buffers, and the restoration equals the un-gathered one bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    p = ctx.Process(target=_nccl_one_rank_worker, args=(free_port(), q))
    p.start()
    same, n_gather, ms = q.get(timeout=600)
    p.join(60)
    print("one-rank RCCL all-gathers: %d, %.3f ms in total" % (n_gather, ms))
    assert same and n_gather == 4


def test_bench_under_the_distributed_launcher_on_one_gpu(tmp_path):
    """The launch line of the driver's scaling run, on the one GPU this box has: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1 ...` for the headline mode and for the patch mode.
    RCCL initialises, the barriers / max-over-ranks run, the patch mode's per-step all-gather runs on device buffers, and rank 0
    prints ONE parseable JSON line with the scaling keys (round-3 verdict: no N > 1 hardware run exists; this is what can be
    exercised without a second GPU)."""
    import json
    import subprocess
    import sys
    from conftest import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(base + ["--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                               "--timesteps", "4", "--batch", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak" and d["roofline"]["frac"] > 0, d
    r = subprocess.run(base + ["--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "patch", "--steps", "1",
                               "--warmup", "0", "--timesteps", "2", "--height", "1100", "--width", "1300"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert d["scaling"] == "strong" and c["windows_per_rank"] == [c["windows_per_step"]] and len(c["all_gather_ms_per_step_per_rank"]) == 1, c
    assert c["all_gather_ms_per_step_per_rank"][0] > 0, c          # the collective ran (one-rank RCCL group, gather branch)


def test_in_place_weight_update_reaches_the_engine(sid_net):
    """an in-place parameter update (optimizer.step / EMA copy through .data) must re-pack the engine's
    weights before the next image - the per-image signature check notices it without mark_weights_dirty()."""
    net, sd = C.build_net(SMALL)
    dn = net.denoise_fn
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(1, 64, 64, seed=51))
    x6, lvl = torch.cat([cond, x_t], 1).cuda(), torch.tensor([[0.4]]).cuda()
    g1, g2 = guide.cuda(), guide.cuda().clone()
    with torch.no_grad():
        a = dn(x6, lvl, g1).clone()
        w = dict(dn.named_parameters())["final_conv.3.weight"]
        w.data.copy_(w.data * 2.0)                        # through .data: no version counter moves
        b = dn(x6, lvl, g2).clone()                       # a new guide tensor = a new image: the signature is re-checked
    r = float((b.abs().mean() / a.abs().mean()))
    assert r > 1.5, r                                     # the doubled final conv weights are in effect (bias unchanged)


# ---- multi-GPU (needs >= 2 GPUs; the driver's 1-GPU box skips) --------------------------------------------------------------------
def _nccl_patch_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    net, sd = C.build_net(SMALL)
    dn = net.denoise_fn
    dn.patch_threshold, dn.patch_skip, dn.patch_padding = 0, 128, 32
    dn.patch_group = dist.group.WORLD
    net.noise_seed = 3
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=4, linear_start=1e-6, linear_end=0.4), torch.device("cuda"))
    cond = torch.from_numpy(synth_inputs(1, 160, 200, seed=5)[0]).cuda()
    with torch.no_grad():
        out = net.super_resolution(cond, False)
    q.put((rank, out.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_sharded_restoration_two_gpus_equals_one():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_nccl_patch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(60)
    net, sd = C.build_net(SMALL)
    dn = net.denoise_fn
    dn.patch_threshold, dn.patch_skip, dn.patch_padding = 0, 128, 32
    net.noise_seed = 3
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=4, linear_start=1e-6, linear_end=0.4), torch.device("cuda"))
    cond = torch.from_numpy(synth_inputs(1, 160, 200, seed=5)[0]).cuda()
    with torch.no_grad():
        one = net.super_resolution(cond, False).cpu()
    assert torch.equal(outs[0], outs[1])
    assert C.metrics(outs[0], one)["rel_rms"] < 1e-3          # other batch composition per engine call: other tilings


# ---- sr.py -p val in batches (round-5 verdict, item 5) ------------------------------------------------------------------------------
def test_batched_rng_streams_equal_single_sample_streams():
    """ucdir_fill_normal_batched / ucdir_sampler_step_rng_batched (ABI 5): sample b of a batch draws exactly what a buffer holding
    that sample alone draws with seed = seeds[b] - bit for bit - so an image's noise does not depend on its batch."""
    from ucdir_amd.ucdir import fill_normal_, sampler_step_rng_
    dev = torch.device("cuda")
    seeds = [11, 2 ** 40 + 5, 123456789]
    st = torch.tensor(seeds, dtype=torch.int64, device=dev)
    x = fill_normal_(torch.empty(3, 3, 40, 56, device=dev), 0, 0, seeds=st)
    for b, s in enumerate(seeds):
        one = fill_normal_(torch.empty(1, 3, 40, 56, device=dev), s, 0)
        assert torch.equal(x[b:b + 1], one)
    assert not torch.equal(x[0], x[1])
    eps = torch.randn(3, 3, 40, 56, device=dev)
    xb = x.clone()
    sampler_step_rng_(xb, eps, 0, 7, 1.3, 0.8, 0.4, 0.55, 0.2, seeds=st)
    for b, s in enumerate(seeds):
        one = x[b:b + 1].clone()
        sampler_step_rng_(one, eps[b:b + 1].contiguous(), s, 7, 1.3, 0.8, 0.4, 0.55, 0.2)
        assert torch.equal(xb[b:b + 1], one)
    from ucdir_amd import lib
    with pytest.raises(lib.UcdirError):
        fill_normal_(torch.empty(3, 3, 40, 56, device=dev), 0, 0, seeds=st[:2])
    with pytest.raises(lib.UcdirError):
        fill_normal_(torch.empty(3, 3, 40, 56, device=dev), 0, 0, seeds=st.cpu())


def test_sr_val_batches_same_sized_images(tmp_path, monkeypatch):
    """`sr.py -p val --batch 16` (reference sr.py:518-561 runs the val loader one image at a time, data/__init__.py:47): five same-sized
    pairs go through ONE DDPM.test call, the odd-sized one alone (HIP-graph replay); every image draws the noise stream of its own
    index, so the grouped run and the one-by-one run (--batch 1) restore the same images: same files, same logged PSNR / SSIM
    within the rounding noise of the two kernel dispatches (a batch of 5 and a batch of 1 pick different tiles - the images agree
    like build and oracle do, not bit for bit)."""
    import importlib.util
    import yaml
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(3)
    for d in ("lq", "gt"):
        os.makedirs(tmp_path / d)
    sizes = [(72, 88)] * 3 + [(80, 72)] + [(72, 88)] * 2
    for i, (h, w) in enumerate(sizes):
        gt = (rs.rand(h // 8, w // 8, 3) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)
        Image.fromarray(gt).save(tmp_path / "gt" / f"{i:03d}.png")
        Image.fromarray((gt * 0.25).astype(np.uint8)).save(tmp_path / "lq" / f"{i:03d}.png")
    cfg = yaml.safe_load(open(os.path.join(root, "config", "sid.yaml")))
    cfg["datasets"]["val"]["data_args"]["dataroot"] = {"lq": str(tmp_path / "lq"), "gt": str(tmp_path / "gt")}
    cfg["model"]["unet"].update(channel_mults=[1, 2, 4], res_blocks=1, attn_res=[32])
    yaml.safe_dump(cfg, open(tmp_path / "sid_small.yaml", "w"))
    spec = importlib.util.spec_from_file_location("sr_entry3", os.path.join(root, "sr.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    from ucdir_amd import model as M
    calls = []
    real_test = M.DDPM.test

    def spy(self, continous=False):
        calls.append(tuple(self.data["SR"].shape))
        return real_test(self, continous)
    monkeypatch.setattr(M.DDPM, "test", spy)
    res = {}
    for tag, batch in (("grouped", "16"), ("single", "1")):
        wd = tmp_path / tag
        os.makedirs(wd)
        monkeypatch.chdir(wd)
        calls.clear()
        res[tag] = sr.main(["-p", "val", "-c", str(tmp_path / "sid_small.yaml"), "--synthetic-weights", "--batch", batch, "--seed", "7"])
        res[tag + "_calls"] = list(calls)
        res[tag + "_files"] = {f: os.path.join(dp, f) for dp, _, fs in os.walk(wd / "experiments") for f in fs if f.endswith("_sr.jpg")}
    assert sorted(res["grouped_calls"]) == sorted([(5, 3, 72, 88), (1, 3, 80, 72)]), res["grouped_calls"]
    assert len(res["single_calls"]) == 6 and all(c[0] == 1 for c in res["single_calls"])
    assert sorted(res["grouped_files"]) == sorted(res["single_files"]) and len(res["grouped_files"]) == 6
    for f in res["grouped_files"]:
        a = np.asarray(Image.open(res["grouped_files"][f]).convert("RGB"))
        b = np.asarray(Image.open(res["single_files"][f]).convert("RGB"))
        assert a.shape == b.shape
        assert O.psnr(a, b) > 36.0, (f, O.psnr(a, b))
    assert abs(res["grouped"][0] - res["single"][0]) < 0.15 and abs(res["grouped"][1] - res["single"][1]) < 0.01, res
