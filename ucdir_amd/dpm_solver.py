"""DPM-Solver++ (multistep, order <= 2) at the denoiser boundary — the sampler the reference wires up in
``sr.py:129-241`` through the third-party ``dpm_solver_pytorch`` (LuChengTHU/dpm-solver; not vendored in the
reference, no version pinned).  That package is absent here, so this is a restatement of the published algorithm
(Lu et al., "DPM-Solver++", 2022: data-prediction multistep solver, discrete-time VP schedule), limited to what
the reference's call uses: ``NoiseScheduleVP('discrete', betas)``, ``algorithm_type='dpmsolver++'``,
``sample(steps, order=2, skip_type='time_uniform', method='multistep')``, ``lower_order_final`` only below
10 steps, no thresholding, no denoise-to-zero.  **Parity unpinned** against the package itself;
``tests`` pin it against an independent oracle restatement and against closed forms of the schedule.

All schedule arithmetic is host-side float64; the per-step tensor update is three fused multiply-adds over
B x 3 x H x W (HBM-trivial, plain torch ops); the cost is the UNet forward, which runs on the HIP engine.
The reference's wrapper feeds ``(t - 1/N) * 1000`` as the network's noise-level input (``sr.py:141-147``) —
kept as is.
"""
import math

import numpy as np
import torch


class NoiseScheduleVP:
    """Discrete-time VP schedule: log(alpha_t) tabulated at t_n = n / N from ``betas``, piecewise linear between."""

    def __init__(self, betas):
        b = np.asarray(betas.detach().cpu().numpy() if torch.is_tensor(betas) else betas, dtype=np.float64)
        self.total_N = int(b.shape[0])
        self.T = 1.0
        self.t_array = np.linspace(0.0, 1.0, self.total_N + 1)[1:]
        self.log_alpha_array = 0.5 * np.log(np.cumprod(1.0 - b))

    def marginal_log_mean_coeff(self, t: float) -> float:
        ta, la = self.t_array, self.log_alpha_array
        i = int(np.clip(np.searchsorted(ta, t), 1, len(ta) - 1))      # linear inter/extrapolation on segment [i-1, i]
        w = (t - ta[i - 1]) / (ta[i] - ta[i - 1])
        return float(la[i - 1] + w * (la[i] - la[i - 1]))

    def marginal_alpha(self, t: float) -> float:
        return math.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t: float) -> float:
        return math.sqrt(1.0 - math.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t: float) -> float:
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * math.log(1.0 - math.exp(2.0 * lm))

    def model_input_time(self, t: float) -> float:
        """``get_model_input_time`` of the reference wrapper (sr.py:141-147), discrete case."""
        return (t - 1.0 / self.total_N) * 1000.0


def multistep_coefficients(ns: NoiseScheduleVP, t_prev_list, t: float, order: int):
    """Coefficients of the DPM-Solver++ multistep update  x_t = a*x + b0*m0 + b1*m1  (m0 = newest data prediction)."""
    t0 = t_prev_list[-1]
    lam0, lam_t = ns.marginal_lambda(t0), ns.marginal_lambda(t)
    h = lam_t - lam0
    a = ns.marginal_std(t) / ns.marginal_std(t0)
    alpha_t = ns.marginal_alpha(t)
    phi1 = math.expm1(-h)
    if order == 1:
        return a, -alpha_t * phi1, 0.0
    t1 = t_prev_list[-2]
    r0 = (lam0 - ns.marginal_lambda(t1)) / h
    # x_t = a x - alpha_t phi1 m0 - 0.5 alpha_t phi1 (m0 - m1) / r0
    c = -0.5 * alpha_t * phi1 / r0
    return a, -alpha_t * phi1 + c, -c


@torch.no_grad()
def sample(model_eps, ns: NoiseScheduleVP, x_T: torch.Tensor, steps: int = 20, order: int = 2):
    """``DPM_Solver(model_fn, ns, algorithm_type='dpmsolver++').sample(x_T, steps, order, 'time_uniform', 'multistep')``.

    ``model_eps(x, t)`` returns the noise prediction at continuous time t in [1/N, 1]."""
    assert order in (1, 2) and steps >= order
    t_T, t_0 = ns.T, 1.0 / ns.total_N
    ts = [float(v) for v in np.linspace(t_T, t_0, steps + 1)]

    def data_pred(x, t):
        return (x - ns.marginal_std(t) * model_eps(x, t)) / ns.marginal_alpha(t)

    x = x_T
    t_prev, m_prev = [ts[0]], [data_pred(x, ts[0])]
    for step in range(1, steps + 1):
        t = ts[step]
        o = min(order, step)                                           # warm-up: first step is first order
        if steps < 10:
            o = min(o, steps + 1 - step)                               # lower_order_final
        a, b0, b1 = multistep_coefficients(ns, t_prev, t, o)
        x = a * x + b0 * m_prev[-1] + (b1 * m_prev[-2] if o == 2 else 0.0)
        if step < steps:
            t_prev.append(t); m_prev.append(data_pred(x, t))
            t_prev, m_prev = t_prev[-2:], m_prev[-2:]
    return x
