"""DPM-Solver++ sampler (reference sr.py:129-241 wires it through a third-party package that is not in the reference
tree): the build's restatement against closed forms of the schedule, against the solver's exactness property and
against the oracle's independent restatement.  CPU only; the HIP path is covered in test_hip_gpu.py."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ucdir_oracle as O  # noqa: E402
from ucdir_amd import dpm_solver as D  # noqa: E402

SCHED = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)


def _ns():
    tab = O.schedule_tables(SCHED)
    return tab, D.NoiseScheduleVP(np.asarray(tab["betas"], dtype=np.float64))


def test_schedule_closed_forms():
    tab, ns = _ns()
    ac = np.cumprod(1.0 - np.asarray(tab["betas"], dtype=np.float64))      # float64 from the (fp32) betas the solver is given
    assert np.allclose(ac, np.asarray(tab["alphas_cumprod"], dtype=np.float64), rtol=1e-5, atol=1e-12)
    assert ns.total_N == 50 and ns.T == 1.0
    for n in (1, 2, 25, 50):                                   # grid points t = n / N: alpha_t^2 = alphas_cumprod[n-1]
        t = n / 50
        assert abs(ns.marginal_alpha(t) ** 2 - ac[n - 1]) < 1e-12 * max(1.0, ac[n - 1])
        assert abs(ns.marginal_alpha(t) ** 2 + ns.marginal_std(t) ** 2 - 1.0) < 1e-12
        assert abs(ns.marginal_lambda(t) - 0.5 * math.log(ac[n - 1] / (1 - ac[n - 1]))) < 1e-9
    mid = ns.marginal_log_mean_coeff(0.03)                     # between t = 1/50 and 2/50: linear in log alpha
    assert abs(mid - 0.5 * (0.5 * math.log(ac[0]) + 0.5 * math.log(ac[1]))) < 1e-12
    lam = [ns.marginal_lambda(t) for t in np.linspace(1.0, 0.02, 21)]
    assert all(b > a for a, b in zip(lam, lam[1:]))             # half log-SNR grows as t falls
    assert abs(ns.model_input_time(1.0) - 980.0) < 1e-9 and abs(ns.model_input_time(0.02)) < 1e-12


def test_exact_for_constant_data_prediction():
    """With x0_theta == const the probability-flow ODE is linear and every DPM-Solver++ step is exact:
    x(t0) = (sigma_0 / sigma_T) x_T + (alpha_0 - alpha_T sigma_0 / sigma_T) x0."""
    _, ns = _ns()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 3, 8, 8, generator=g, dtype=torch.float64)
    xT = torch.randn(2, 3, 8, 8, generator=g, dtype=torch.float64)

    def eps(x, t):
        return (x - ns.marginal_alpha(t) * x0) / ns.marginal_std(t)

    for steps, order in ((20, 2), (6, 2), (5, 1)):
        got = D.sample(eps, ns, xT, steps=steps, order=order)
        s0, sT = ns.marginal_std(0.02), ns.marginal_std(1.0)
        want = (s0 / sT) * xT + (ns.marginal_alpha(0.02) - ns.marginal_alpha(1.0) * s0 / sT) * x0
        assert float((got - want).abs().max()) < 1e-9, (steps, order)


def test_matches_oracle_restatement_on_a_nonlinear_model():
    tab, ns = _ns()
    g = torch.Generator().manual_seed(1)
    xT = torch.randn(1, 3, 12, 10, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, generator=g, dtype=torch.float64) * 0.5

    def eps(x, t):                                             # any smooth time-dependent map will do
        return torch.tanh(torch.einsum("oc,bchw->bohw", w, x)) * (0.5 + t) + 0.1 * x

    for steps, order in ((20, 2), (7, 2), (4, 1)):
        a = D.sample(eps, ns, xT, steps=steps, order=order)
        b = O.dpm_solver_pp_sample(None, tab, None, None, xT, steps=steps, order=order, eps_fn=eps)
        assert float((a - b).abs().max()) < 1e-9 * max(1.0, float(b.abs().max())), (steps, order)


def test_first_step_is_first_order_and_coefficients_sum():
    _, ns = _ns()
    a, b0, b1 = D.multistep_coefficients(ns, [1.0], 0.951, 1)
    assert b1 == 0.0 and abs(a - ns.marginal_std(0.951) / ns.marginal_std(1.0)) < 1e-15
    a2, c0, c1 = D.multistep_coefficients(ns, [1.0, 0.951], 0.902, 2)
    h = ns.marginal_lambda(0.902) - ns.marginal_lambda(0.951)
    assert abs((c0 + c1) - ns.marginal_alpha(0.902) * (-math.expm1(-h))) < 1e-12     # weights of the data predictions
