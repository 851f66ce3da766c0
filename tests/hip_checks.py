"""Parity checks of the HIP path against the CPU oracle (shared by pytest -m gpu and tools/gpu_check.py).

Every function returns a dict of error metrics; thresholds live in the tests.
Tolerances (stated once): the HIP path computes convolutions / attention with bf16 operands and
fp32 accumulation and stores activations in bf16, so vs the fp32 oracle we expect
  * single operator on bf16-representable inputs : rel-RMS <~ 3e-3 (weight + output rounding), bound OP_TOL = 4e-3
  * full 61-GroupNorm-deep forward               : rel-RMS 1.2e-2 (small configuration) / 1.43-1.51e-2 (full SID) measured; the oracle's
    bf16-emulation mode (oracle.dy3h_naive_forward_emu: rounding where the kernels round) sits at 1.47-1.50e-2 from the fp32 oracle on the
    same inputs, i.e. the whole difference IS the numerics plan; bound FWD_TOL = 1.7e-2 (tests/test_hip_gpu.py)
  * one layer against the emulation on the HIP path's own input activations (layerwise_emu_case): <= 6.7e-4 measured, bound 2e-3
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ucdir_oracle as O
from ucdir_amd import lib as ulib
from ucdir_amd.spec import UNetConfig
from ucdir_amd.weights import synth_inputs, synth_state_dict

DEV = "cuda"


def bfr(t):
    return t.to(torch.bfloat16).float()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def metrics(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    d = got - ref
    rms = ref.pow(2).mean().sqrt().item()
    return {"rel_rms": (d.pow(2).mean().sqrt().item() / max(rms, 1e-12)), "max_abs": d.abs().max().item(),
            "ref_rms": rms, "nan": bool(torch.isnan(got).any())}


def rng(seed):
    return torch.Generator().manual_seed(seed)


def conv_case(B, H, W, c0, c1, cout, ksize, mode, gn, silu, residual, seed=0):
    """ucdir_op_conv vs torch on bf16-representable inputs."""
    L = ulib.load()
    g = rng(seed)
    cin = c0 + c1
    x0 = bfr(torch.randn(B, c0, H, W, generator=g) * 1.3 + 0.6)
    x1 = bfr(torch.randn(B, c1, H, W, generator=g) * 0.7 - 0.4) if c1 else None
    w = torch.randn(cout, cin, ksize, ksize, generator=g) * math.sqrt(1.5 / (cin * ksize * ksize))
    b = torch.randn(cout, generator=g) * 0.1
    gamma = (1 + 0.25 * torch.randn(cin, generator=g)) if gn else None
    beta = (0.2 * torch.randn(cin, generator=g)) if gn else None
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
    res = bfr(torch.randn(B, cout, Ho, Wo, generator=g)) if residual else None
    # reference
    x = torch.cat([x0, x1], 1) if c1 else x0
    h = F.group_norm(x, 1, gamma, beta, eps=1e-5) if gn else x
    if mode == 1:
        y = F.conv2d(h, w, b, stride=2, padding=1)
    elif mode == 2:
        y = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), w, b, padding=1)
    else:
        y = F.conv2d(h, w, b, padding=ksize // 2)
    if silu:
        y = O.swish(y)
    if residual:
        y = y + res
    # device
    dx0, dx1 = x0.to(DEV), (x1.to(DEV) if c1 else None)
    dres = res.to(DEV) if residual else None
    dy = torch.empty(B, cout, Ho, Wo, device=DEV)
    stats = np.zeros((B, 2), dtype=np.float64)
    wn, bn = w.numpy().copy(), b.numpy().copy()
    gn_, bt_ = (gamma.numpy().copy(), beta.numpy().copy()) if gn else (None, None)
    ulib.check(L.ucdir_op_conv(_p(dx0), c0, _p(dx1), c1, B, H, W, _hp(wn), _hp(bn), _hp(gn_), _hp(bt_), cout, ksize,
                               mode, int(silu), _p(dres), _p(dy), _hp(stats), _st()))
    torch.cuda.synchronize()
    m = metrics(dy, y)
    ref_stats = np.stack([y.double().sum(dim=(1, 2, 3)).numpy(), y.double().pow(2).sum(dim=(1, 2, 3)).numpy()], 1)
    m["stats_rel"] = float(np.abs(stats - ref_stats).max() / np.abs(ref_stats).max())
    # border vs interior error (a wrong GroupNorm border class shows up here)
    d = (dy.cpu() - y).abs()
    m["max_abs_border"] = float(torch.cat([d[..., 0, :].flatten(), d[..., -1, :].flatten(), d[..., :, 0].flatten(),
                                           d[..., :, -1].flatten()]).max())
    return m


def conv_res_case(B, H, W, c0, c1, cout, seed=0):
    """ucdir_op_conv_res (conv1 with GroupNorm fold + swish, and the block's 1x1 res_conv, one launch) vs torch."""
    L = ulib.load()
    g = rng(seed)
    cin = c0 + c1
    x0 = bfr(torch.randn(B, c0, H, W, generator=g) * 1.3 + 0.6)
    x1 = bfr(torch.randn(B, c1, H, W, generator=g) * 0.7 - 0.4) if c1 else None
    w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(1.5 / (cin * 9))
    b = torch.randn(cout, generator=g) * 0.1
    wr = torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(1.5 / cin)
    br = torch.randn(cout, generator=g) * 0.1
    gamma = 1 + 0.25 * torch.randn(cin, generator=g)
    beta = 0.2 * torch.randn(cin, generator=g)
    x = torch.cat([x0, x1], 1) if c1 else x0
    y = O.swish(F.conv2d(F.group_norm(x, 1, gamma, beta, eps=1e-5), w, b, padding=1))
    yr = F.conv2d(x, wr, br)
    dx0, dx1 = x0.to(DEV), (x1.to(DEV) if c1 else None)
    dy = torch.empty(B, cout, H, W, device=DEV); dyr = torch.empty(B, cout, H, W, device=DEV)
    stats = np.zeros((B, 2), dtype=np.float64)
    ulib.check(L.ucdir_op_conv_res(_p(dx0), c0, _p(dx1), c1, B, H, W, _hp(w.numpy().copy()), _hp(b.numpy().copy()),
                                   _hp(gamma.numpy().copy()), _hp(beta.numpy().copy()), _hp(wr.numpy().copy()), _hp(br.numpy().copy()),
                                   cout, 1, _p(dy), _p(dyr), _hp(stats), _st()))
    torch.cuda.synchronize()
    m = metrics(dy, y)
    mr = metrics(dyr, yr)
    m["res_rel_rms"], m["res_nan"] = mr["rel_rms"], mr["nan"]
    got = dy.double().cpu()
    st_ref = np.stack([got.sum(dim=(1, 2, 3)).numpy(), got.pow(2).sum(dim=(1, 2, 3)).numpy()], 1)
    m["stats_rel"] = float(np.abs(stats - st_ref).max() / np.abs(st_ref).max())
    d = (dy.cpu() - y).abs()
    m["max_abs_border"] = float(torch.cat([d[..., 0, :].flatten(), d[..., -1, :].flatten(), d[..., :, 0].flatten(),
                                           d[..., :, -1].flatten()]).max())
    return m


def akgm_case(B, C, H, W, seed=0):
    L = ulib.load()
    g = rng(seed)
    h = bfr(torch.randn(B, C, H, W, generator=g).abs() * 0.8 - 0.2)
    att = torch.randn(B, 8, H, W, generator=g) * 0.5
    res = bfr(torch.randn(B, C, H, W, generator=g))
    wsp = torch.randn(8 * C, C // 8, 3, 3, generator=g) * math.sqrt(1.5 / (9 * C // 8))
    bsp = torch.randn(8 * C, generator=g) * 0.1
    gamma = 1 + 0.25 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    hn = F.group_norm(h, 1, gamma, beta, eps=1e-5)
    hset = F.conv2d(hn, wsp, bsp, padding=1, groups=8).view(B, C, 8, H, W)
    y = O.swish((hset * att.unsqueeze(1)).sum(2)) + res
    dy = torch.empty(B, C, H, W, device=DEV)
    dh, datt, dres = h.to(DEV), att.to(DEV), res.to(DEV)      # keep alive: raw pointers cross the ABI
    stats = np.zeros((B, 2), dtype=np.float64)
    ulib.check(L.ucdir_op_akgm(_p(dh), _p(datt), _p(dres), B, C, H, W, _hp(wsp.numpy().copy()),
                               _hp(bsp.numpy().copy()), _hp(gamma.numpy().copy()), _hp(beta.numpy().copy()), _p(dy), _hp(stats), _st()))
    torch.cuda.synchronize()
    m = metrics(dy, y)
    # (sum, sum of squares) the launch accumulated for its output vs float64 sums of the output it stored (bf16-rounded after
    # the statistics were taken: the rounding noise averages out)
    got = dy.double().cpu()
    st_ref = np.stack([got.sum(dim=(1, 2, 3)).numpy(), got.pow(2).sum(dim=(1, 2, 3)).numpy()], 1)
    m["stats_rel"] = float(np.abs(stats - st_ref).max() / np.abs(st_ref).max())
    m["stats"] = stats.tolist()
    d = (dy.cpu() - y).abs()
    m["max_abs_border"] = float(torch.cat([d[..., 0, :].flatten(), d[..., -1, :].flatten(), d[..., :, 0].flatten(),
                                           d[..., :, -1].flatten()]).max())
    return m


def attention_case(B, C, H, W, seed=0, fp16=False, flash=1):
    """flash: 1 forces the flash kernel (the engine's own choice sends grids of < 32 query blocks through the three-launch
    materialised path), 0 forces the materialised path, -1 leaves the choice to the engine."""
    L = ulib.load()
    g = rng(seed)
    x = bfr(torch.randn(B, C, H, W, generator=g) * 1.2 + 0.3)
    sd = {"a.norm.weight": 1 + 0.25 * torch.randn(C, generator=g), "a.norm.bias": 0.2 * torch.randn(C, generator=g),
          "a.qkv.weight": torch.randn(3 * C, C, 1, 1, generator=g) * math.sqrt(3.0 / C),
          "a.out.weight": torch.randn(C, C, 1, 1, generator=g) * math.sqrt(1.5 / C),
          "a.out.bias": torch.randn(C, generator=g) * 0.1}
    y = O.self_attention(sd, "a.", x)
    dy = torch.empty(B, C, H, W, device=DEV)
    n = lambda k: sd[k].numpy().copy()
    dx = x.to(DEV)
    ulib.check(L.ucdir_debug_flag(b"flash", flash))
    try:
        ulib.check(L.ucdir_op_attention(_p(dx), B, C, H, W, _hp(n("a.norm.weight")), _hp(n("a.norm.bias")),
                                        _hp(n("a.qkv.weight")), _hp(n("a.out.weight")), _hp(n("a.out.bias")), int(fp16),
                                        _p(dy), _st()))
        torch.cuda.synchronize()
    finally:
        ulib.check(L.ucdir_debug_flag(b"flash", -1))
    # the residual dominates y; report the error relative to the attention branch alone
    m = metrics(dy, y)
    branch = y - x
    m["rel_rms_branch"] = float(((dy.cpu() - y).pow(2).mean().sqrt() / branch.pow(2).mean().sqrt()).item())
    return m


def build_net(cfg: UNetConfig, seed=0):
    """Product netG (DY3h on the HIP engine) + oracle state dict with identical synthetic weights."""
    from ucdir_amd import networks
    opt = {"model": {"which_model_G": "ucdir", "unet_name": "DY3h", "diffusion_name": "ResiGaussianGuideDY",
                     "unet": dict(in_channel=cfg.in_channel, out_channel=cfg.out_channel,
                                  inner_channel=cfg.inner_channel, channel_mults=list(cfg.channel_mults),
                                  attn_res=list(cfg.attn_res), res_blocks=cfg.res_blocks, dropout=cfg.dropout,
                                  norm_groups=1, image_size=cfg.image_size),
                     "diffusion": dict(image_size=128, channels=3, conditional=True)}}
    net = networks.define_G(opt)
    np_sd = synth_state_dict(cfg, seed)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in np_sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net = net.to(DEV).eval()
    return net, O.to_torch_sd(np_sd)


def forward_case(cfg: UNetConfig, B, H, W, levels, seed=11, taps=False, net_sd=None, emu=False):
    """HIP forward vs the fp32 oracle (``eps``, and per layer with ``taps``).  ``emu``: also against the oracle's bf16-emulation
    mode (oracle.dy3h_naive_forward_emu: rounding where the kernels round), final output ``eps_emu`` and per layer ``<name>@emu``:
    what is left between HIP and the emulation is summation order, not the rounding plan, so that bound is several times tighter."""
    net, sd = net_sd if net_sd is not None else build_net(cfg)
    cond, guide, x_t = synth_inputs(B, H, W, seed=seed)
    cond, guide, x_t = map(torch.from_numpy, (cond, guide, x_t))
    lvl = torch.tensor(levels, dtype=torch.float32).view(B, 1)
    x6 = torch.cat([cond, x_t], 1)
    otaps = {} if taps else None
    etaps = {} if (taps and emu) else None
    ph, pw = O.pad32(H), O.pad32(W)
    if taps:
        ref_full = O.dy3h_naive_forward(sd, F.pad(x6, (0, pw, 0, ph), mode="reflect"), lvl,
                                        F.pad(guide, (0, pw, 0, ph), mode="reflect"), taps=otaps)
        ref = ref_full[..., :-ph, :-pw]
    else:
        ref = O.dy3h_forward(sd, x6, lvl, guide)
    ref_emu = O.dy3h_forward(sd, x6, lvl, guide, emulate_bf16=True, taps=etaps) if emu else None
    with torch.no_grad():
        eps = net.denoise_fn(x6.to(DEV), lvl.to(DEV), guide.to(DEV))
    torch.cuda.synchronize()
    out = {"eps": metrics(eps, ref)}
    if emu:
        out["eps_emu"] = metrics(eps, ref_emu)
        out["emu_vs_oracle"] = metrics(ref_emu, ref)
    if taps:
        from ucdir_amd.spec import unet_layers
        for Ld in unet_layers(cfg):
            key = "denoise_fn." + Ld.name
            got = net.denoise_fn.debug_read(Ld.name, "out")
            torch.cuda.synchronize()
            out[Ld.name] = metrics(got, otaps[key])
            if emu:
                out[Ld.name + "@emu"] = metrics(got, etaps[key])
            if Ld.kind == "block":
                got = net.denoise_fn.debug_read(Ld.name, "h1")
                torch.cuda.synchronize()
                out[Ld.name + ":h1"] = metrics(got, otaps[key + ".res_block.h1"])
                if emu:
                    out[Ld.name + ":h1@emu"] = metrics(got, etaps[key + ".res_block.h1"])
    return out, eps.cpu(), ref


def layerwise_emu_case(cfg: UNetConfig, B, H, W, levels, seed=11, net_sd=None):
    """Every layer of a HIP forward against the oracle's bf16-emulation mode evaluated ON THE HIP PATH'S OWN INPUTS (teacher
    forcing: oracle.dy3h_naive_forward_emu(force=...)): per layer {name: metrics(HIP stored activation, bf16(emulated layer
    output))}, plus 'eps' = the final conv on the HIP path's last activation.  Two realisations of the rounding noise decorrelate
    over the depth of the network (HIP vs emulation end to end: 1.2e-2, like HIP vs the fp32 oracle); layer by layer on identical
    inputs they differ by summation order only, so a systematic error of a few 1e-3 in ANY single layer shows."""
    net, sd = net_sd if net_sd is not None else build_net(cfg)
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(B, H, W, seed=seed))
    lvl = torch.tensor(levels, dtype=torch.float32).view(B, 1)
    x6 = torch.cat([cond, x_t], 1)
    with torch.no_grad():
        eps = net.denoise_fn(x6.to(DEV), lvl.to(DEV), guide.to(DEV))
    torch.cuda.synchronize()
    from ucdir_amd.spec import unet_layers
    force = {}
    for Ld in unet_layers(cfg):
        key = "denoise_fn." + Ld.name
        force[key] = net.denoise_fn.debug_read(Ld.name, "out").float().cpu()
        if Ld.kind == "block":
            force[key + ".res_block.h1"] = net.denoise_fn.debug_read(Ld.name, "h1").float().cpu()
    torch.cuda.synchronize()
    ph, pw = O.pad32(H), O.pad32(W)
    etaps = {}
    e = O.dy3h_naive_forward_emu(sd, F.pad(x6, (0, pw, 0, ph), mode="reflect"), lvl, F.pad(guide, (0, pw, 0, ph), mode="reflect"),
                                 taps=etaps, force=force)[..., :-ph, :-pw]
    out = {"eps": metrics(eps, e)}
    for k, v in force.items():
        out[k[len("denoise_fn."):]] = metrics(v, etaps[k].to(torch.bfloat16).float())
    return out


def conv_stats_case(B, H, W, cin, cout, ksize, mode, gn, runs=3, seed=0):
    """GroupNorm statistics a conv launch accumulates for its OUTPUT (fixed-point atomics) against float64 sums of the
    output it stored, and their run-to-run reproducibility.  A size-independent property: usable at bench size."""
    L = ulib.load()
    g = rng(seed)
    x = bfr(torch.randn(B, cin, H, W, generator=g) * 1.3 + 0.6).to(DEV)
    w = (torch.randn(cout, cin, ksize, ksize, generator=g) * math.sqrt(1.5 / (cin * ksize * ksize))).numpy().copy()
    b = (torch.randn(cout, generator=g) * 0.1).numpy().copy()
    gm = (1 + 0.25 * torch.randn(cin, generator=g)).numpy().copy() if gn else None
    bt = (0.2 * torch.randn(cin, generator=g)).numpy().copy() if gn else None
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
    outs, sts = [], []
    for _ in range(runs):
        y = torch.empty(B, cout, Ho, Wo, device=DEV)
        st = np.zeros((B, 2), dtype=np.float64)
        ulib.check(L.ucdir_op_conv(_p(x), cin, _p(None), 0, B, H, W, _hp(w), _hp(b), _hp(gm), _hp(bt), cout, ksize, mode, 1,
                                   _p(None), _p(y), _hp(st), _st()))
        torch.cuda.synchronize()
        outs.append(y); sts.append(st.copy())
    ref = np.stack([outs[0].double().sum(dim=(1, 2, 3)).cpu().numpy(), outs[0].double().pow(2).sum(dim=(1, 2, 3)).cpu().numpy()], 1)
    return {"stats_rel": float(np.abs((sts[0] - ref) / ref).max()),
            "outputs_reproducible": all(torch.equal(outs[0], o) for o in outs[1:]),
            "stats_reproducible": all(np.array_equal(sts[0], s_) for s_ in sts[1:])}


def predictor_case(B, H, W, seed=3, net_sd=None):
    """UNetSeeInDark on the HIP engine vs the oracle (model/ucdir.py:352-403)."""
    net, sd = net_sd if net_sd is not None else build_net(UNetConfig(inner_channel=64, channel_mults=(1, 2), res_blocks=1,
                                                                    attn_res=(64,), image_size=128))
    x = torch.from_numpy(synth_inputs(B, H, W, seed=seed)[0])
    ref = O.predictor_forward(sd, x)
    with torch.no_grad():
        got = net.predictor(x.to(DEV))
    torch.cuda.synchronize()
    return metrics(got, ref)


def sampler_step_case(seed=0):
    from ucdir_amd.ucdir import sampler_step_
    g = rng(seed)
    tab = O.schedule_tables(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4))
    out = {}
    for t in (49, 25, 1, 0):
        x = torch.randn(2, 3, 40, 56, generator=g)
        eps = torch.randn(2, 3, 40, 56, generator=g)
        nz = torch.randn(2, 3, 40, 56, generator=g)
        ref = O.p_sample_step(tab, x, eps, t, nz)
        sig = float(np.exp(np.float32(0.5) * tab["posterior_log_variance_clipped"][t])) if t > 0 else 0.0
        dx = x.to(DEV).clone()
        sampler_step_(dx, eps.to(DEV), nz.to(DEV) if t > 0 else None, tab["sqrt_recip_alphas_cumprod"][t],
                      tab["sqrt_recipm1_alphas_cumprod"][t], tab["posterior_mean_coef1"][t],
                      tab["posterior_mean_coef2"][t], sig)
        torch.cuda.synchronize()
        out[f"t{t}"] = metrics(dx, ref)
    return out


def sampler_case(cfg: UNetConfig, H, W, T, seed=5, net_sd=None):
    """T-step restoration with injected noise: HIP path vs oracle; returns PSNR on uint8 images."""
    net, sd = net_sd if net_sd is not None else build_net(cfg)
    sched = dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=0.4)
    tab = O.schedule_tables(sched)
    net.set_new_noise_schedule(sched, torch.device(DEV))
    cond = torch.from_numpy(synth_inputs(1, H, W, seed=seed)[0])
    g = rng(seed + 100)
    noises = [torch.randn(1, 3, H, W, generator=g) for _ in range(T)]
    ref = O.super_resolution(sd, tab, cond, noises, continous=False)
    net.noise_source = lambda shape, device, k: noises[k].to(device)
    with torch.no_grad():
        got = net.super_resolution(cond.to(DEV), False)
    net.noise_source = None
    torch.cuda.synchronize()
    m = metrics(got, ref.view_as(got.cpu()))
    m["psnr_u8"] = O.psnr(O.tensor2img(got.cpu()), O.tensor2img(ref))
    return m
