"""Host-side mirror of the reference's ``model/ucdir.py`` for the sampling path.

``DY3h`` keeps the reference's constructor arguments, parameter names/shapes (so reference
checkpoints load with ``load_state_dict``) and call signature
``denoise_fn(x (B,6,H,W), noise_level (B,1), guide=(B,3,H,W)) -> (B,3,H,W)``
(reference: model/ucdir.py:204-307), but it has no PyTorch forward: every call goes to the HIP
engine behind include/ucdir_hip.h.  On a machine without the library or without a GPU it raises.

``UNetSeeInDark`` (the one-shot predictor, model/ucdir.py:310-416) runs on the same engine
(``ucdir_predictor_*``); it is 0.13 % of a 50-step restoration.

Inter-step patch split (model/ucdir.py:298-300 -> utils/util.py:108-146): ``DY3h.forward`` evaluates the
windows of a step as batches and, when ``patch_group`` is set (``sr.py`` / ``model.DDPM`` set it to the
world group when WORLD_SIZE > 1), shards them over the ranks with one all-gather per step.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from .patch import patch_forward_guide
from .spec import UNetConfig, predictor_param_shapes, unet_param_shapes


def _stream_ptr(device=None):
    """Current PyTorch stream of ``device`` (default: the current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    m.register_parameter(parts[-1], param)


def _default_init(name, shape):
    """nn.Conv2d / nn.Linear / nn.GroupNorm default initialisers."""
    leaf = name.rsplit(".", 1)[-1]
    if "norm" in name or name.startswith("final_conv.0"):
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else None
    t = torch.empty(shape)
    if leaf == "weight":
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    else:
        t.zero_()            # bias bound needs the weight's fan-in; filled in by the caller
    return t


def _param_signature(ps):
    """Cheap fingerprint of a parameter list: storage identity, autograd version counters (in-place ops, optimizer.step,
    nn.init) and a CONTENT checksum (sum of the L1 and L2 norms: ``p.data.copy_(ema)`` writes through the ``.data`` alias and
    bumps no version counter).  One multi-tensor launch and one host read - evaluated once per image, not per step."""
    with torch.no_grad():
        n1 = torch.stack(torch._foreach_norm(ps, 1)).double().sum()
        n2 = torch.stack(torch._foreach_norm(ps, 2)).double().sum()
        chk = (float(n1), float(n2))
    return (len(ps), sum(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr(), chk)


class DY3h(nn.Module):
    """Conditional UNet denoiser; same signature as the reference's ``DY3h`` (model/ucdir.py:204-207)."""

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=1, channel_mults=(1, 2, 4, 8, 8),
                 attn_res=(8,), res_blocks=3, dropout=0, with_noise_level_emb=True, image_size=128,
                 resname="ResnetBlockDY3h", attn_dtype="bf16"):
        super().__init__()
        if not with_noise_level_emb or resname != "ResnetBlockDY3h":
            raise NotImplementedError("only the DY3h configuration of config/sid.yaml is implemented")
        if attn_dtype not in ("bf16", "fp16"):
            raise ValueError("attn_dtype must be 'bf16' or 'fp16'")
        # not a reference argument: operand type of the attention MFMAs (q, k, v', P); 'fp16' is the JPEG
        # configuration's "fp16 attention MFMA path" (BASELINE.json configs[4]); accumulation / softmax stay fp32
        self.attn_dtype = attn_dtype
        self.cfg = UNetConfig(in_channel=in_channel, out_channel=out_channel, inner_channel=inner_channel,
                              norm_groups=norm_groups, channel_mults=tuple(channel_mults), attn_res=tuple(attn_res),
                              res_blocks=res_blocks, dropout=dropout, image_size=image_size)
        shapes = unet_param_shapes(self.cfg)
        self._pnames = list(shapes)
        for name, shape in shapes.items():
            _attach(self, name, nn.Parameter(_default_init(name, shape)))
        for name, shape in shapes.items():      # bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch
            if name.endswith(".bias") and "norm" not in name and not name.startswith("final_conv.0"):
                w = dict(self.named_parameters())[name[:-5] + ".weight"]
                bound = 1.0 / math.sqrt(max(int(np.prod(w.shape[1:])), 1))
                nn.init.uniform_(dict(self.named_parameters())[name], -bound, bound)
        self.patch_threshold = 1024 * 1024   # model/ucdir.py:298
        self.patch_skip, self.patch_padding = 1024, 64
        self.patch_group = None              # torch.distributed group: windows of a step are sharded over its ranks
        self.patch_max_batch = 8             # windows per engine call (1024^2 windows: 2.3 GB of workspace each)
        self.patch_force_gather = False      # tests: run the all-gather branch even on a one-rank group (RCCL init + collective on hardware)
        self.patch_timers = None             # optional list: (start, end) CUDA events of every all-gather (bench.py --mode patch)
        self._patch_cache = {}               # padded guide windows + gather / paste buffers of the running restoration
        self.use_graph = False               # replay each forward from a HIP graph (B = 1 latency path)
        self._h = None
        self._hdev = None                    # device index the engine handle was created on
        self._wdirty = True                  # parameters changed since the engine packed them
        self._wsig = None                    # signature of the parameter storage the engine packed (see _weights_signature)
        self._gkey = None
        self._sig_hold = False               # a sampler checked the weight signature for the running restoration (hold_weight_check)

    def hold_weight_check(self, on=True):
        """Samplers call this at the start (True) and end (False) of a restoration: the parameter signature is evaluated ONCE per
        restoration instead of at every guide change - in the patch path with more than one chunk per rank the chunk guides
        alternate, which made it a re-check (two multi-tensor norms + host syncs) per chunk per step (round-3 advice)."""
        if on and not self._wdirty and self._wsig is not None and self._wsig != self._weights_signature():
            self._wdirty = True
        self._sig_hold = bool(on)

    def clear_patch_cache(self):
        """Drop the padded guide windows and gather buffers of the last patch-split restoration (hundreds of MB at full size)."""
        self._patch_cache.clear()

    # ---- weight-change tracking: O(1) per forward ------------------------------------------------
    # Every path that replaces or rewrites parameter storage in this repo's scope goes through one of these hooks
    # (load_state_dict, .to() / .cuda() / .float() via _apply).  Code that mutates ``p.data`` in place must call
    # ``mark_weights_dirty()``.  In addition ``prepare_guide`` re-checks a cheap signature of the parameters (_param_signature:
    # version counters + content checksum) whenever the guide changes, i.e. once per image: in-place updates that bypass the
    # hooks (optimizer.step, ``p.data.copy_(ema)``, ``vector_to_parameters``, ``nn.init.*``) are picked up before the next restoration.
    def mark_weights_dirty(self):
        self._wdirty = True

    def _weights_signature(self):
        return _param_signature(list(self.parameters()))

    def _apply(self, fn, *a, **k):
        self._wdirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._wdirty = True
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):       # also reached when a parent module loads
        self._wdirty = True
        return super()._load_from_state_dict(*a, **k)

    # ---- engine plumbing -------------------------------------------------------------------------
    def _device_index(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise _lib.UcdirError("DY3h runs only on an MI355X (move the module to 'cuda'); there is no CPU path")
        return p.device.index if p.device.index is not None else torch.cuda.current_device()

    def _handle(self):
        L = _lib.load()
        if self._h is not None and self._hdev != self._device_index():
            # the module moved to another GPU (.to('cuda:1')): the handle, its packed weights and workspace live on the old one
            L.ucdir_destroy(self._h)
            self._h, self._wdirty, self._gkey = None, True, None
        if self._h is None:
            c = _lib.UcdirConfig()
            cfg = self.cfg
            c.in_channel, c.out_channel, c.inner_channel = cfg.in_channel, cfg.out_channel, cfg.inner_channel
            c.n_mults = len(cfg.channel_mults)
            for i, v in enumerate(cfg.channel_mults):
                c.channel_mults[i] = v
            c.n_attn_res = len(cfg.attn_res)
            for i, v in enumerate(cfg.attn_res):
                c.attn_res[i] = v
            c.res_blocks, c.image_size, c.device = cfg.res_blocks, cfg.image_size, self._device_index()
            c.attn_fp16 = 1 if self.attn_dtype == "fp16" else 0
            h = ctypes.c_void_p()
            _lib.check(L.ucdir_create(ctypes.byref(c), ctypes.byref(h)))
            self._h = h
            self._hdev = c.device
        return self._h

    def _sync_weights(self):
        """(Re)pack weights into the engine when parameters changed (load_state_dict, .to, mark_weights_dirty)."""
        if not self._wdirty and self._h is not None and self._hdev == self._device_index():
            return
        L = _lib.load()
        h = self._handle()
        for name, p in self.named_parameters():
            a = np.ascontiguousarray(p.detach().float().cpu().numpy())
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            _lib.check(L.ucdir_load_weight(h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim))
        _lib.check(L.ucdir_finalize_weights(h))
        _lib.check(L.ucdir_set_graph(h, 1 if self.use_graph else 0))
        self._wdirty = False
        self._wsig = self._weights_signature()
        self._gkey = None

    def set_graph(self, on=True):
        """Replay forwards from a HIP graph captured once per (cond, x_t, level, eps) buffer set."""
        self.use_graph = bool(on)
        if self._h is not None:
            _lib.check(_lib.load().ucdir_set_graph(self._h, 1 if on else 0))

    def _dev(self):
        return torch.device("cuda", self._device_index())

    def _check_dev(self, name, t):
        if not (torch.is_tensor(t) and t.is_cuda and t.device == self._dev()):
            raise _lib.UcdirError(f"{name} must be a CUDA tensor on {self._dev()} (got "
                                  f"{t.device if torch.is_tensor(t) else type(t).__name__}); there is no CPU path")

    def prepare_guide(self, guide, pad_mode=1):
        L = _lib.load()
        key = (guide.data_ptr(), guide._version, tuple(guide.shape), tuple(guide.stride()), pad_mode)
        if key != self._gkey and not self._sig_hold and not self._wdirty and self._wsig != self._weights_signature():
            self._wdirty = True                 # a parameter was updated in place since the engine packed the weights
        self._sync_weights()
        self._check_dev("guide", guide)
        if key != self._gkey:
            g = guide.contiguous().float()
            B, _, H, W = g.shape
            _lib.check(L.ucdir_prepare_guide(self._handle(), _ptr(g), B, H, W, pad_mode, _stream_ptr(self._dev())))
            self._gkey = key
            self._guide_keepalive = (guide, g)

    def forward_split(self, cond, x_t, noise_level, guide, pad_mode=1, out=None):
        """eps for cat[cond, x_t] without materialising the concat (model/diffusion.py:166).
        ``out``: optional persistent (B,3,H,W) fp32 buffer for eps (graph replay needs stable pointers)."""
        L = _lib.load()
        for n, t in (("cond", cond), ("x_t", x_t), ("noise_level", noise_level), ("guide", guide)):
            self._check_dev(n, t)
        if cond.dim() != 4 or cond.shape != x_t.shape or cond.shape[1] != 3:
            raise ValueError(f"cond {tuple(cond.shape)} and x_t {tuple(x_t.shape)} must both be (B,3,H,W)")
        if tuple(guide.shape) != tuple(cond.shape):
            raise ValueError(f"guide {tuple(guide.shape)} must match cond {tuple(cond.shape)}")
        self.prepare_guide(guide, pad_mode)
        cond = cond.contiguous().float()
        x_t = x_t.contiguous().float()
        lvl = noise_level.reshape(-1).contiguous().float()
        if lvl.numel() != cond.shape[0]:
            raise ValueError("noise_level must have one entry per sample")
        B, _, H, W = x_t.shape
        if out is not None:
            if not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.shape == x_t.shape
                    and out.device == x_t.device):
                raise ValueError("out must be a contiguous fp32 CUDA tensor shaped like x_t")
            eps = out
        else:
            eps = torch.empty_like(x_t)
        self._last_shape = (B, (H // 32 + 1) * 32, (W // 32 + 1) * 32) if pad_mode else (B, H, W)
        _lib.check(L.ucdir_unet_forward(self._handle(), _ptr(cond), _ptr(x_t), _ptr(lvl), _ptr(eps), B, H, W,
                                        _stream_ptr(self._dev())))
        self._io_keepalive = (cond, x_t, lvl)
        return eps

    def naiveforward(self, x, time, guide):
        """model/ucdir.py:270-293 (H, W multiples of 32, no padding)."""
        return self.forward_split(x[:, :3], x[:, 3:], time, guide, pad_mode=0)

    def forward(self, x, time, guide):
        """model/ucdir.py:295-307."""
        _, _, h, w = x.shape
        if h * w > self.patch_threshold:
            return patch_forward_guide(x, self.naiveforward, params={"time": time, "guide": guide},
                                       skip=self.patch_skip, padding=self.patch_padding, group=self.patch_group,
                                       max_batch=self.patch_max_batch, cache=self._patch_cache,
                                       force_gather=self.patch_force_gather, timers=self.patch_timers)
        return self.forward_split(x[:, :3], x[:, 3:], time, guide, pad_mode=1)

    def debug_read(self, layer, what="out"):
        """Activation of ``layer`` from the last forward as (B,C,Hc,Wc) fp32 (tests)."""
        L = _lib.load()
        from .spec import unet_layers
        B, Hc, Wc = self._last_shape
        for Ld in unet_layers(self.cfg):
            if Ld.name == layer and what == "attw":       # the block's time weights, (B, 8) fp32
                out = torch.empty(B, 8, device=next(self.parameters()).device)
                _lib.check(L.ucdir_debug_read(self._handle(), layer.encode(), b"attw", _ptr(out), out.numel(),
                                              _stream_ptr(self._dev())))
                return out
            if Ld.name == layer:
                lvl = Ld.level + (1 if Ld.kind == "down" else (-1 if Ld.kind == "up" else 0))
                out = torch.empty(B, Ld.cout, Hc >> lvl, Wc >> lvl, device=next(self.parameters()).device)
                _lib.check(L.ucdir_debug_read(self._handle(), layer.encode(), what.encode(), _ptr(out), out.numel(),
                                              _stream_ptr(self._dev())))
                return out
        raise KeyError(layer)

    def forward_flops(self):
        return float(_lib.load().ucdir_forward_flops(self._handle()))

    def __del__(self):
        try:
            if self._h is not None:
                _lib.load().ucdir_destroy(self._h)
        except Exception:
            pass


def sampler_step_(x_t, eps, noise, c_recip, c_recipm1, coef1, coef2, sigma):
    """In-place ancestral update on the device (model/diffusion.py:150-158,171-183)."""
    L = _lib.load()
    if not (x_t.is_cuda and x_t.is_contiguous() and x_t.dtype == torch.float32):
        raise _lib.UcdirError("sampler_step_ needs contiguous fp32 CUDA tensors")
    if noise is not None and not (noise.is_cuda and noise.is_contiguous() and noise.dtype == torch.float32
                                  and noise.numel() == x_t.numel()):
        raise _lib.UcdirError("sampler_step_: noise must be a contiguous fp32 CUDA tensor shaped like x_t")
    eps = eps.contiguous()
    if not (eps.is_cuda and eps.dtype == torch.float32 and eps.numel() == x_t.numel()):
        raise _lib.UcdirError("sampler_step_: eps must be an fp32 CUDA tensor shaped like x_t")
    nz = _ptr(noise) if noise is not None else ctypes.c_void_p(0)
    _lib.check(L.ucdir_sampler_step(_ptr(x_t), _ptr(eps), nz, x_t.numel(), float(c_recip),
                                    float(c_recipm1), float(coef1), float(coef2), float(sigma), _stream_ptr(x_t.device)))
    return x_t


def _check_seeds(seeds, x, who):
    """Per-sample seeds: one int64 per sample of ``x`` on its device (bit pattern = the uint64 Philox key)."""
    if not (torch.is_tensor(seeds) and seeds.is_cuda and seeds.device == x.device and seeds.dtype == torch.int64
            and seeds.is_contiguous() and seeds.numel() == x.shape[0] and x.dim() >= 2 and (x.numel() // x.shape[0]) % 4 == 0):
        raise _lib.UcdirError(who + ": seeds must be a contiguous int64 CUDA tensor with one entry per sample of x "
                                    "(samples of a multiple of 4 elements)")


def sampler_step_rng_(x_t, eps, seed, step, c_recip, c_recipm1, coef1, coef2, sigma, seeds=None):
    """The same update with the noise of (seed, step, element) generated inside the kernel (no noise tensor).
    ``seeds`` (int64 CUDA tensor, one per sample): every sample draws its own stream, counters local to the sample - the noise
    of an image is then independent of the batch it is restored in (``seed`` is ignored)."""
    L = _lib.load()
    if not (x_t.is_cuda and x_t.is_contiguous() and x_t.dtype == torch.float32):
        raise _lib.UcdirError("sampler_step_rng_ needs contiguous fp32 CUDA tensors")
    eps = eps.contiguous()
    if not (eps.is_cuda and eps.dtype == torch.float32 and eps.numel() == x_t.numel()):
        raise _lib.UcdirError("sampler_step_rng_: eps must be an fp32 CUDA tensor shaped like x_t")
    if seeds is not None:
        _check_seeds(seeds, x_t, "sampler_step_rng_")
        _lib.check(L.ucdir_sampler_step_rng_batched(_ptr(x_t), _ptr(eps), x_t.numel(), x_t.numel() // x_t.shape[0], float(c_recip),
                                                    float(c_recipm1), float(coef1), float(coef2), float(sigma), _ptr(seeds), int(step),
                                                    _stream_ptr(x_t.device)))
        return x_t
    _lib.check(L.ucdir_sampler_step_rng(_ptr(x_t), _ptr(eps), x_t.numel(), float(c_recip), float(c_recipm1), float(coef1),
                                        float(coef2), float(sigma), int(seed) & (2 ** 64 - 1), int(step), _stream_ptr(x_t.device)))
    return x_t


def gather_windows(x, pad, win_dev, skip):
    """Window batch of the inter-step patch split in one launch (utils/util.py:113-137: reflect pad + one slice per window):
    x (B, C, H, W) fp32 CUDA, win_dev (nwin, 2) int32 CUDA = (h0, w0) in padded coordinates -> (nwin * B, C, skip, skip)."""
    L = _lib.load()
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and win_dev.is_cuda and win_dev.dtype == torch.int32
            and win_dev.is_contiguous() and win_dev.dim() == 2 and win_dev.shape[1] == 2):
        raise _lib.UcdirError("gather_windows needs a contiguous fp32 CUDA canvas and an (nwin, 2) int32 CUDA window list")
    B, C, H, W = x.shape
    nwin = win_dev.shape[0]
    out = torch.empty((nwin * B, C, skip, skip), dtype=torch.float32, device=x.device)
    _lib.check(L.ucdir_gather_windows(_ptr(x), B, C, H, W, int(pad), _ptr(win_dev), nwin, int(skip), _ptr(out), _stream_ptr(x.device)))
    return out


def fill_normal_(x, seed, step=0, seeds=None):
    """x <- N(0, 1) from the sampler's counter-based generator (x_T = step 0 of the stream the update kernel draws from);
    ``seeds``: per-sample streams as in ``sampler_step_rng_``."""
    L = _lib.load()
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        raise _lib.UcdirError("fill_normal_ needs a contiguous fp32 CUDA tensor")
    if seeds is not None:
        _check_seeds(seeds, x, "fill_normal_")
        _lib.check(L.ucdir_fill_normal_batched(_ptr(x), x.numel(), x.numel() // x.shape[0], _ptr(seeds), int(step), _stream_ptr(x.device)))
        return x
    _lib.check(L.ucdir_fill_normal(_ptr(x), x.numel(), int(seed) & (2 ** 64 - 1), int(step), _stream_ptr(x.device)))
    return x


class UNetSeeInDark(nn.Module):
    """Initial-restoration predictor (model/ucdir.py:310-416) on the HIP engine; parameter names and
    shapes match the reference, there is no PyTorch forward."""

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__()
        if in_channels != 3 or out_channels != 3:
            raise NotImplementedError("the UCDIR predictor is 3 -> 3 channels")
        for name, shape in predictor_param_shapes(in_channels, out_channels).items():
            t = torch.empty(shape)
            if name.endswith("weight"):
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            else:
                nn.init.uniform_(t, -0.05, 0.05)
            _attach(self, name, nn.Parameter(t))
        self._h = None
        self._wdirty = True

    def mark_weights_dirty(self):
        self._wdirty = True

    def _apply(self, fn, *a, **k):
        self._wdirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._wdirty = True
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._wdirty = True
        return super()._load_from_state_dict(*a, **k)

    def _device_index(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise _lib.UcdirError("UNetSeeInDark runs only on an MI355X (move the module to 'cuda'); there is no CPU path")
        return p.device.index if p.device.index is not None else torch.cuda.current_device()

    def _handle(self):
        L = _lib.load()
        dev = self._device_index()
        if self._h is not None and getattr(self, "_hdev", None) != dev:      # moved to another GPU: rebuild there
            L.ucdir_predictor_destroy(self._h)
            self._h, self._wdirty = None, True
        if self._h is None:
            h = ctypes.c_void_p()
            _lib.check(L.ucdir_predictor_create(dev, ctypes.byref(h)))
            self._h, self._hdev = h, dev
        return self._h

    def _weights_signature(self):
        return _param_signature(list(self.parameters()))

    def _sync_weights(self):
        # the predictor runs once per image: the in-place-update check (see DY3h._weights_signature) costs 46 attribute reads
        if not self._wdirty and self._h is not None and getattr(self, "_wsig", None) != self._weights_signature():
            self._wdirty = True
        if not self._wdirty and self._h is not None and getattr(self, "_hdev", None) == self._device_index():
            return
        L = _lib.load()
        h = self._handle()
        for name, p in self.named_parameters():
            a = np.ascontiguousarray(p.detach().float().cpu().numpy())
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            _lib.check(L.ucdir_predictor_load_weight(h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim))
        _lib.check(L.ucdir_predictor_finalize(h))
        self._wdirty = False
        self._wsig = self._weights_signature()

    def forward(self, x):
        L = _lib.load()
        self._sync_weights()
        if not (torch.is_tensor(x) and x.is_cuda and x.device == next(self.parameters()).device):
            raise _lib.UcdirError("UNetSeeInDark input must be a CUDA tensor on the module's device; there is no CPU path")
        x = x.contiguous().float()
        B, _, H, W = x.shape
        y = torch.empty_like(x)
        _lib.check(L.ucdir_predictor_forward(self._handle(), _ptr(x), _ptr(y), B, H, W, _stream_ptr(x.device)))
        return y

    def __del__(self):
        try:
            if self._h is not None:
                _lib.load().ucdir_predictor_destroy(self._h)
        except Exception:
            pass
