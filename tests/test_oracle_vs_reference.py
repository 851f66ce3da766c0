"""Live comparison of the oracle with the REAL reference (build container only: /root/reference is read-only
and never travels; on the GPU box these tests skip and the committed fixtures of tests/golden do the pinning)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference not present")

from oracle import ucdir_oracle as O  # noqa: E402
from ucdir_amd.spec import UNetConfig  # noqa: E402
from ucdir_amd.weights import synth_inputs, synth_state_dict  # noqa: E402

TINY = dict(inner_channel=8, channel_mults=[1, 2], res_blocks=1, attn_res=[64], image_size=128)


@pytest.fixture(scope="module")
def ref_net():
    import yaml
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("lpips", types.ModuleType("lpips"))   # imported at module top, unused on this path
    from model import networks
    opt = yaml.safe_load(open(os.path.join(REF, "config", "sid.yaml")))
    opt["model"]["unet"].update(TINY)
    net = networks.define_G(opt).eval()
    cfg = UNetConfig.from_opt(opt["model"]["unet"])
    sd = synth_state_dict(cfg, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return net, O.to_torch_sd(sd)


def test_forward_and_predictor(ref_net):
    net, sd = ref_net
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(2, 40, 56, seed=31))
    lvl = torch.tensor([[0.1], [0.8]])
    x6 = torch.cat([cond, x_t], 1)
    with torch.no_grad():
        a = net.denoise_fn(x6, lvl, guide=guide)
        p = net.predictor(cond)
    assert torch.allclose(O.dy3h_forward(sd, x6, lvl, guide), a, atol=2e-5)
    assert torch.allclose(O.predictor_forward(sd, cond), p, atol=2e-5)


def test_ddim_sample(ref_net):
    """Pins oracle.ddim_sample (model/diffusion.py:247-294): same RNG draws recorded and injected."""
    net, sd = ref_net
    sched = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)
    net.set_new_noise_schedule(sched, torch.device("cpu"))
    tab = O.schedule_tables(sched)
    cond, guide, _ = map(torch.from_numpy, synth_inputs(1, 48, 40, seed=32))
    torch.manual_seed(11)
    st = torch.get_rng_state()
    with torch.no_grad():
        ref = net.ddim_sample(cond, False, kwargs={"guide": guide})
    torch.set_rng_state(st)
    draws = [torch.randn(cond.shape) for _ in range(6)]
    got = O.ddim_sample(sd, tab, cond, guide, draws)
    assert torch.allclose(got, ref, atol=5e-5), (got - ref).abs().max()


def test_p_sample_loop_50_steps(ref_net):
    net, sd = ref_net
    sched = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)
    net.set_new_noise_schedule(sched, torch.device("cpu"))
    tab = O.schedule_tables(sched)
    cond = torch.from_numpy(synth_inputs(1, 40, 40, seed=33)[0])
    torch.manual_seed(12)
    st = torch.get_rng_state()
    with torch.no_grad():
        ref = net.super_resolution(cond, True)
    torch.set_rng_state(st)
    draws = [torch.randn(cond.shape) for _ in range(50)]
    got = O.super_resolution(sd, tab, cond, draws, continous=True)
    assert got.shape == ref.shape == (11, 3, 40, 40)            # cond + 10 snapshots (sample_inter = 5)
    assert torch.allclose(got, ref, atol=1e-4), (got - ref).abs().max()
