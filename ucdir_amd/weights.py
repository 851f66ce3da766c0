"""Deterministic synthetic weights.

The reference ships no checkpoint (``experiments/init`` is empty), so
parity tests, golden fixtures and ``bench.py`` all use weights produced by this
generator: numpy PCG64, one stream per parameter name, fan-in scaled normals and
deliberately non-trivial GroupNorm affines / biases so every term of every layer
is exercised.  The same call regenerates bit-identical tensors on the GPU box.
"""
import zlib
from collections import OrderedDict

import numpy as np

from .spec import UNetConfig, netg_param_shapes


def _rng_for(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def synth_param(name: str, shape, seed: int = 0) -> np.ndarray:
    g = _rng_for(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ".norm" in name or name.endswith("final_conv.0.weight") or name.endswith("final_conv.0.bias")
    if is_norm:
        if leaf == "weight":
            return (1.0 + 0.25 * g.standard_normal(shape)).astype(np.float32)
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:
        if ".upv" in name:                      # ConvTranspose2d (cin, cout, 2, 2): fan-in = cin
            fan_in = shape[0]
        else:
            fan_in = shape[1] * shape[2] * shape[3]
    else:
        fan_in = shape[-1]
    std = float(np.sqrt(1.5 / max(fan_in, 1)))
    return (std * g.standard_normal(shape)).astype(np.float32)


def synth_state_dict(cfg: UNetConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Parameters of the full netG (``denoise_fn.*`` + ``predictor.*``) as numpy fp32 arrays."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in netg_param_shapes(cfg).items():
        out[name] = synth_param(name, shape, seed)
    return out


def synth_inputs(B: int, H: int, W: int, seed: int = 0):
    """cond in [-1,1), a guide-like image, x_t ~ N(0,1): numpy fp32, NCHW."""
    g = np.random.Generator(np.random.PCG64([seed, 0xC0DE]))
    cond = (2.0 * g.random((B, 3, H, W)) - 1.0).astype(np.float32)
    guide = np.clip(cond * 0.5 + 0.2 * g.standard_normal((B, 3, H, W)), -1, 1).astype(np.float32)
    x_t = g.standard_normal((B, 3, H, W)).astype(np.float32)
    return cond, guide, x_t
