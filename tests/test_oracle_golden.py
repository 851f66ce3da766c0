"""Pin the CPU oracle against fixtures produced by the real reference (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ucdir_oracle as O
from ucdir_amd.spec import UNetConfig
from ucdir_amd.weights import synth_state_dict

TINY = UNetConfig(inner_channel=8, channel_mults=(1, 2), res_blocks=1, attn_res=(64,), image_size=128)
SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag,T", [("T50", 50), ("T100", 100), ("T8", 8)])
def test_schedule_tables(golden_dir, tag, T):
    g = _g(golden_dir, f"schedule_{tag}.npz")
    tab = O.schedule_tables(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=0.4))
    for k in g.files:
        assert tab[k].dtype == g[k].dtype, k
        np.testing.assert_array_equal(tab[k], g[k], err_msg=k)   # same numpy algebra -> bit exact


def test_schedule_known_answers():
    # SURVEY.md §8(a1) probe values (T=50, linear 1e-6 -> 0.4)
    tab = O.schedule_tables(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4))
    np.testing.assert_allclose(tab["betas"][[0, 1, 25, 49]], [1e-6, 8.1642447e-3, 0.20408212, 0.4], rtol=1e-6)
    np.testing.assert_allclose(tab["sqrt_recipm1_alphas_cumprod"][[0, 25, 49]],
                               [9.9995045e-4, 4.0553584, 349.00836], rtol=1e-6)
    lv = [tab["sqrt_alphas_cumprod_prev"][t + 1] for t in (49, 25, 0)]
    np.testing.assert_allclose(lv, [0.002865232, 0.239415851, 0.9999995], rtol=1e-6)


def test_weight_generator_matches_fixture(golden_dir):
    g = _g(golden_dir, "tiny_forward.npz")
    sd = synth_state_dict(TINY, 0)
    n = 0
    for k in g.files:
        if k.startswith("w::"):
            np.testing.assert_array_equal(sd[k[3:]], g[k])
            n += 1
    assert n > 50


def test_tiny_forward_and_parts(golden_dir):
    g = _g(golden_dir, "tiny_forward.npz")
    sd = O.to_torch_sd(synth_state_dict(TINY, 0))
    x6 = torch.cat([_t(g["cond"]), _t(g["x_t"])], 1)
    lvl = _t(g["level"])
    eps = O.dy3h_forward(sd, x6, lvl, _t(g["guide"]))
    np.testing.assert_allclose(eps.numpy(), g["eps"], rtol=0, atol=2e-5)
    temb = O.noise_embedding(sd, lvl, "denoise_fn.")
    np.testing.assert_allclose(temb.numpy(), g["temb"], rtol=0, atol=1e-6)
    yb = O.resblock_dy3h(sd, "denoise_fn.downs.1.res_block.", _t(g["block_x"]), temb, _t(g["block_guide"]))
    np.testing.assert_allclose(yb.numpy(), g["block_y"], rtol=0, atol=1e-5)
    ya = O.self_attention(sd, "denoise_fn.mid.0.attn.", _t(g["attn_x"]))
    np.testing.assert_allclose(ya.numpy(), g["attn_y"], rtol=0, atol=1e-5)
    pred = O.predictor_forward(sd, _t(g["cond"]))
    np.testing.assert_allclose(pred.numpy(), g["predictor"], rtol=0, atol=2e-5)


def test_tiny_sampler_8_steps(golden_dir):
    g = _g(golden_dir, "tiny_forward.npz")
    sd = O.to_torch_sd(synth_state_dict(TINY, 0))
    tab = O.schedule_tables(dict(schedule="linear", n_timestep=8, linear_start=1e-6, linear_end=0.4))
    noises = [_t(n) for n in g["sampler_noise"]]
    out = O.super_resolution(sd, tab, _t(g["cond"][:1]), noises, continous=True)
    assert out.shape == g["sampler_out"].shape          # cond + 8 snapshots (sample_inter = 1)
    np.testing.assert_allclose(out.numpy(), g["sampler_out"], rtol=0, atol=5e-5)


def test_tiny_ddim(golden_dir):
    g = _g(golden_dir, "tiny_ddim.npz")
    sd = O.to_torch_sd(synth_state_dict(TINY, 0))
    tab = O.schedule_tables(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4))
    out = O.ddim_sample(sd, tab, _t(g["cond"]), _t(g["guide"]), [_t(n) for n in g["noise"]])
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=5e-5)


def test_tiny_patch_split(golden_dir):
    g = _g(golden_dir, "tiny_patch.npz")
    sd = O.to_torch_sd(synth_state_dict(TINY, 0))
    x6 = torch.cat([_t(g["cond"]), _t(g["x_t"])], 1)
    out = O.dy3h_forward(sd, x6, _t(g["level"]), _t(g["guide"]), patch_threshold=0, skip=128, padding=32)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-5)


def test_patch_windows_reference_geometry():
    # 1424x2128 -> +128 (DDPM.test) -> +128 (patch pad) = 1680x2384 -> 6 windows
    w = O.patch_windows(1680, 2384, 1024, 64)
    assert [(a, c) for a, _, c, _ in w] == [(0, 0), (0, 896), (0, 1360), (656, 0), (656, 896), (656, 1360)]


def test_small_forward(golden_dir):
    g = _g(golden_dir, "small_forward.npz")
    sd = O.to_torch_sd(synth_state_dict(SMALL, 0))
    x6 = torch.cat([_t(g["cond"]), _t(g["x_t"])], 1)
    eps = O.dy3h_forward(sd, x6, _t(g["level"]), _t(g["guide"]))
    ref = g["eps"].astype(np.float32)                     # stored as fp16 to keep the fixture small
    np.testing.assert_allclose(eps.numpy(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("i", [0, 1, 2], ids=["t49", "t25", "t0"])
def test_sid_forward_full_config(golden_dir, i):
    """All three stored noise levels: t = 49 (level 0.0029, where sqrt_recipm1 = 349 amplifies eps errors most), t = 25, t = 0."""
    g = _g(golden_dir, "sid_forward.npz")
    from ucdir_amd.weights import synth_inputs
    sd = O.to_torch_sd(synth_state_dict(SID, 0))
    cond, guide, x_t = synth_inputs(1, 256, 256, seed=21)
    x6 = torch.cat([_t(cond), _t(x_t)], 1)
    e = O.dy3h_forward(sd, x6, torch.tensor([[g["levels"][i]]], dtype=torch.float32), _t(guide))
    np.testing.assert_allclose(e[0, :, 100:132, 60:92].numpy(), g[f"eps{i}_crop"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(e[0, :, ::8, ::8].numpy(), g[f"eps{i}_ds"], rtol=0, atol=1e-4)
    st = np.array([e.mean(), e.std(), e.min(), e.max()], dtype=np.float64)
    np.testing.assert_allclose(st, g[f"eps{i}_stats"], rtol=1e-4, atol=1e-5)
    pred = O.predictor_forward(sd, _t(cond))
    np.testing.assert_allclose(pred[0, :, 100:132, 60:92].numpy(), g["pred_crop"], rtol=0, atol=1e-4)


def test_sid_forward_full_config_b2(golden_dir):
    """SURVEY 8c fixture (ii), B = 2: two samples with different levels in one call of the reference."""
    g = _g(golden_dir, "sid_forward_b2.npz")
    from ucdir_amd.weights import synth_inputs
    sd = O.to_torch_sd(synth_state_dict(SID, 0))
    cond, guide, x_t = synth_inputs(2, 256, 256, seed=22)
    e = O.dy3h_forward(sd, torch.cat([_t(cond), _t(x_t)], 1), _t(g["levels"]), _t(guide))
    for b in range(2):
        np.testing.assert_allclose(e[b, :, 100:132, 60:92].numpy(), g[f"b{b}_crop"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(e[b, :, ::8, ::8].numpy(), g[f"b{b}_ds"], rtol=0, atol=1e-4)
        st = np.array([e[b].mean(), e[b].std(), e[b].min(), e[b].max()], dtype=np.float64)
        np.testing.assert_allclose(st, g[f"b{b}_stats"], rtol=1e-4, atol=1e-5)


def test_sid_real_image(golden_dir):
    """SURVEY 8c fixture (v): a 256^2 crop of the reference's dataset/celebahq_64_512/sr_64_512/00030.png as the condition,
    predictor output as the guide (super_resolution's wiring), one denoiser call."""
    g = _g(golden_dir, "sid_real_image.npz")
    from ucdir_amd.weights import synth_inputs
    sd = O.to_torch_sd(synth_state_dict(SID, 0))
    c = torch.from_numpy(g["cond_u8"]).permute(2, 0, 1)[None].float() / 255.0 * 2.0 - 1.0
    pred = O.predictor_forward(sd, c)
    np.testing.assert_allclose(pred[0, :, 100:132, 60:92].numpy(), g["pred_crop"], rtol=0, atol=1e-4)
    xt = _t(synth_inputs(1, 256, 256, seed=23)[2])
    e = O.dy3h_forward(sd, torch.cat([c, xt], 1), _t(g["level"]), pred)
    np.testing.assert_allclose(e[0, :, 100:132, 60:92].numpy(), g["eps_crop"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(e[0, :, ::8, ::8].numpy(), g["eps_ds"], rtol=0, atol=1e-4)
    st = np.array([e.mean(), e.std(), e.min(), e.max()], dtype=np.float64)
    np.testing.assert_allclose(st, g["eps_stats"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cfg", [TINY, SMALL], ids=["tiny", "small"])
def test_bf16_emulation_mode_is_the_same_network(cfg):
    """dy3h_naive_forward_emu (the engine's numerics plan: GroupNorm folded into bf16 weights + border-class tables, Upsample as
    four 2x2 parity classes, the out projection folded into the value rows) with its rounding switched OFF is an algebraic
    re-arrangement of dy3h_naive_forward, which the reference's fixtures pin: final output and every layer tap agree to fp32
    re-association.  With rounding ON it stays within the documented bf16 error of the fp32 oracle (and is not identical to it)."""
    from ucdir_amd.weights import synth_inputs
    sd = O.to_torch_sd(synth_state_dict(cfg, 0))
    H = 64                                # (reflect padding of the pad-32 wrapper needs H, W > 32)
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(2, H, H, seed=3))
    x6 = torch.cat([cond, x_t], 1)
    lvl = torch.tensor([[0.3], [0.7]])
    ta, tb = {}, {}
    a = O.dy3h_naive_forward(sd, x6, lvl, guide, taps=ta)
    b = O.dy3h_naive_forward_emu(sd, x6, lvl, guide, taps=tb, rnd=False)
    rel = lambda u, v: float((u - v).pow(2).mean().sqrt() / v.pow(2).mean().sqrt())
    assert rel(b, a) < 1e-5, rel(b, a)
    assert set(tb) == set(ta)
    for k in ta:
        assert rel(tb[k], ta[k]) < 1e-5, (k, rel(tb[k], ta[k]))
    c = O.dy3h_naive_forward_emu(sd, x6, lvl, guide, rnd=True)
    assert 1e-3 < rel(c, a) < 2e-2, rel(c, a)
    d = O.dy3h_forward(sd, x6, lvl, guide, emulate_bf16=True)             # the pad-32 wrapper routes to the same function
    assert d.shape == a.shape and 1e-3 < rel(d, O.dy3h_forward(sd, x6, lvl, guide)) < 2e-2
