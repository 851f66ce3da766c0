"""Static description of the DY3h denoiser and the UNetSeeInDark predictor.

Everything the host side and the C-ABI library need to know about the network
topology is derived here from the ``model.unet`` section of the reference's
YAML config (reference: config/sid.yaml:41-56, model/ucdir.py:204-268 for the
layer order, model/ucdir.py:310-350 for the predictor).

The functions return plain Python data (no torch modules): a flat list of
layer records in execution order and an ordered ``name -> shape`` table that
reproduces the reference's ``state_dict`` grammar, so
reference checkpoints load unchanged.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

NSET = 8  # number of kernel sets in the conditional integration module (ucdir.py:104)


@dataclass
class UNetConfig:
    in_channel: int = 6
    out_channel: int = 3
    inner_channel: int = 32
    norm_groups: int = 1
    channel_mults: Tuple[int, ...] = (1, 2, 4, 8, 8)
    attn_res: Tuple[int, ...] = (8,)
    res_blocks: int = 3
    dropout: float = 0.0
    image_size: int = 128

    @staticmethod
    def from_opt(unet_opt: dict) -> "UNetConfig":
        known = {k: unet_opt[k] for k in
                 ("in_channel", "out_channel", "inner_channel", "norm_groups", "channel_mults",
                  "attn_res", "res_blocks", "dropout", "image_size") if k in unet_opt}
        if "channel_mults" in known:
            known["channel_mults"] = tuple(int(v) for v in known["channel_mults"])
        if "attn_res" in known:
            known["attn_res"] = tuple(int(v) for v in known["attn_res"])
        cfg = UNetConfig(**known)
        if cfg.norm_groups != 1:
            raise NotImplementedError("DY3h is only defined by the reference config with norm_groups=1")
        return cfg


@dataclass
class Layer:
    """One entry of downs / mid / ups in execution order."""
    kind: str                 # 'stem' | 'block' | 'down' | 'up'
    name: str                 # state_dict prefix, e.g. 'downs.4'
    level: int                # resolution level (0 = full resolution)
    cin: int = 0              # input channels (for ups blocks: x channels + skip channels)
    cout: int = 0
    attn: bool = False
    skip_c: int = 0           # channels taken from the skip stack (ups blocks only)
    push_skip: bool = False   # output is appended to the skip stack (all downs layers)


def unet_layers(cfg: UNetConfig) -> List[Layer]:
    """Execution-order layer list; mirrors DY3h.__init__ (model/ucdir.py:219-260)."""
    layers: List[Layer] = []
    inner = cfg.inner_channel
    nm = len(cfg.channel_mults)
    pre = inner
    feat = [pre]
    res = cfg.image_size
    level = 0
    layers.append(Layer("stem", "downs.0", 0, cfg.in_channel, inner, push_skip=True))
    idx = 1
    for ind in range(nm):
        last = ind == nm - 1
        use_attn = res in cfg.attn_res
        cm = inner * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks):
            layers.append(Layer("block", f"downs.{idx}", level, pre, cm, attn=use_attn, push_skip=True))
            feat.append(cm)
            pre = cm
            idx += 1
        if not last:
            layers.append(Layer("down", f"downs.{idx}", level, pre, pre, push_skip=True))
            feat.append(pre)
            idx += 1
            res //= 2
            level += 1
    layers.append(Layer("block", "mid.0", level, pre, pre, attn=True))
    layers.append(Layer("block", "mid.1", level, pre, pre, attn=False))
    idx = 0
    for ind in reversed(range(nm)):
        last = ind < 1
        use_attn = res in cfg.attn_res
        cm = inner * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks + 1):
            sc = feat.pop()
            layers.append(Layer("block", f"ups.{idx}", level, pre + sc, cm, attn=use_attn, skip_c=sc))
            pre = cm
            idx += 1
        if not last:
            layers.append(Layer("up", f"ups.{idx}", level, pre, pre))
            idx += 1
            res *= 2
            level -= 1
    return layers


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every parameter of DY3h, reference key order."""
    inner = cfg.inner_channel
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["noise_level_mlp.1.weight"] = (inner * 4, inner)
    out["noise_level_mlp.1.bias"] = (inner * 4,)
    out["noise_level_mlp.3.weight"] = (inner, inner * 4)
    out["noise_level_mlp.3.bias"] = (inner,)
    for L in unet_layers(cfg):
        p = L.name
        if L.kind == "stem":
            out[f"{p}.weight"] = (L.cout, L.cin, 3, 3)
            out[f"{p}.bias"] = (L.cout,)
        elif L.kind in ("down", "up"):
            out[f"{p}.conv.weight"] = (L.cout, L.cin, 3, 3)
            out[f"{p}.conv.bias"] = (L.cout,)
        else:
            r = f"{p}.res_block"
            out[f"{r}.noise_func.0.weight"] = (NSET, inner)
            out[f"{r}.noise_func.0.bias"] = (NSET,)
            out[f"{r}.noise_func.2.weight"] = (NSET, NSET)
            out[f"{r}.noise_func.2.bias"] = (NSET,)
            out[f"{r}.norm1.weight"] = (L.cin,)
            out[f"{r}.norm1.bias"] = (L.cin,)
            out[f"{r}.conv1.weight"] = (L.cout, L.cin, 3, 3)
            out[f"{r}.conv1.bias"] = (L.cout,)
            out[f"{r}.norm2.weight"] = (L.cout,)
            out[f"{r}.norm2.bias"] = (L.cout,)
            out[f"{r}.conv2.0.weight"] = (2 * NSET, 3, 1, 1)
            out[f"{r}.conv2.0.bias"] = (2 * NSET,)
            out[f"{r}.conv2.2.weight"] = (NSET, NSET, 3, 3)
            out[f"{r}.conv2.2.bias"] = (NSET,)
            out[f"{r}.spdyconv.weight"] = (L.cout * NSET, L.cout // NSET, 3, 3)
            out[f"{r}.spdyconv.bias"] = (L.cout * NSET,)
            if L.cin != L.cout:
                out[f"{r}.res_conv.weight"] = (L.cout, L.cin, 1, 1)
                out[f"{r}.res_conv.bias"] = (L.cout,)
            if L.attn:
                a = f"{p}.attn"
                out[f"{a}.norm.weight"] = (L.cout,)
                out[f"{a}.norm.bias"] = (L.cout,)
                out[f"{a}.qkv.weight"] = (3 * L.cout, L.cout, 1, 1)
                out[f"{a}.out.weight"] = (L.cout, L.cout, 1, 1)
                out[f"{a}.out.bias"] = (L.cout,)
    final_c = inner * cfg.channel_mults[0]
    out["final_conv.0.weight"] = (final_c,)
    out["final_conv.0.bias"] = (final_c,)
    out["final_conv.3.weight"] = (cfg.out_channel, final_c, 3, 3)
    out["final_conv.3.bias"] = (cfg.out_channel,)
    return out


# UNetSeeInDark (model/ucdir.py:315-350): (name, kind, cin, cout)
_PREDICTOR = [
    ("conv1_1", "c3", 3, 32), ("conv1_2", "c3", 32, 32),
    ("conv2_1", "c3", 32, 64), ("conv2_2", "c3", 64, 64),
    ("conv3_1", "c3", 64, 128), ("conv3_2", "c3", 128, 128),
    ("conv4_1", "c3", 128, 256), ("conv4_2", "c3", 256, 256),
    ("conv5_1", "c3", 256, 512), ("conv5_2", "c3", 512, 512),
    ("upv6", "t2", 512, 256), ("conv6_1", "c3", 512, 256), ("conv6_2", "c3", 256, 256),
    ("upv7", "t2", 256, 128), ("conv7_1", "c3", 256, 128), ("conv7_2", "c3", 128, 128),
    ("upv8", "t2", 128, 64), ("conv8_1", "c3", 128, 64), ("conv8_2", "c3", 64, 64),
    ("upv9", "t2", 64, 32), ("conv9_1", "c3", 64, 32), ("conv9_2", "c3", 32, 32),
    ("conv10_1", "c1", 32, 3),
]


def predictor_param_shapes(in_channels: int = 3, out_channels: int = 3):
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, kind, cin, cout in _PREDICTOR:
        if name == "conv1_1":
            cin = in_channels
        if name == "conv10_1":
            cout = out_channels
        if kind == "c3":
            out[f"{name}.weight"] = (cout, cin, 3, 3)
        elif kind == "c1":
            out[f"{name}.weight"] = (cout, cin, 1, 1)
        else:  # ConvTranspose2d weight is (cin, cout, kh, kw)
            out[f"{name}.weight"] = (cin, cout, 2, 2)
        out[f"{name}.bias"] = (cout,)
    return out


def netg_param_shapes(cfg: UNetConfig):
    """Full ``netG`` parameter table (denoise_fn.* then predictor.*), without schedule buffers."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for k, v in unet_param_shapes(cfg).items():
        out[f"denoise_fn.{k}"] = v
    for k, v in predictor_param_shapes().items():
        out[f"predictor.{k}"] = v
    return out


SCHEDULE_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def padded_size(n: int, fac: int = 32) -> int:
    """DY3h.forward pads every side length to the next multiple of 32 *above* it
    (model/ucdir.py:303-304): 256 -> 288, 288 -> 320."""
    return (n // fac + 1) * fac
