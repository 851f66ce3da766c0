"""``DDPM`` model wrapper for validation (reference: model/model.py:35-267, model/__init__.py:5-10).

Only the inference surface ``sr.py -p val`` uses: device placement, EMA checkpoint loading,
``feed_data`` / ``test`` (reflect-pad 64, restore, crop; model/model.py:124-138),
``set_new_noise_schedule`` and ``get_current_visuals``.  One process per GPU; no DDP wrapper is
needed for sampling (the reference's wrapper only broadcasts parameters).

Multi-GPU (``torch.distributed`` initialised, world size > 1): images whose padded area exceeds the
denoiser's patch threshold have the windows of every step sharded over the ranks of ``patch_group``
(default: the world group) with one all-gather per step; all ranks draw identical noise from
``noise_seed`` so the sampler update needs no second exchange (utils/util.py:108-146).
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import networks

logger = logging.getLogger("base")


class DDPM:
    def __init__(self, opt, device=None):
        self.opt = opt
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.begin_step, self.begin_epoch = 0, 0
        self.netG = networks.define_G(opt).to(self.device)
        self.netG.set_loss(self.device)
        self.set_new_noise_schedule(opt["model"]["beta_schedule"]["train"], schedule_phase="train")
        self.load_network()
        self.setup_distributed()

    def setup_distributed(self, group=None, noise_seed=1234):
        """Shard the patch-split windows over the ranks of ``group`` (default: world) when running multi-process."""
        import torch.distributed as dist
        if group is None and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        g = group if group is not None else dist.group.WORLD
        self.netG.denoise_fn.patch_group = g
        self._shared_noise_seed = noise_seed       # every rank must apply the identical sampler update to a SHARDED image;
        self.netG.noise_seed = None                # test() installs the seed only for images that take the sharded patch split

    def feed_data(self, data):
        self.data = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data.items()}

    def test(self, continous=False):
        self.netG.eval()
        pd = 64
        sr = F.pad(self.data["SR"], (pd, pd, pd, pd), mode="reflect")
        dn = self.netG.denoise_fn
        seed = getattr(self, "_shared_noise_seed", None)
        sharded = seed is not None and dn.patch_group is not None and sr.shape[-1] * sr.shape[-2] > dn.patch_threshold
        idx = self.data.get("Index", 0)
        if torch.is_tensor(idx):
            idxs = [int(v) for v in idx.reshape(-1).tolist()]
        elif isinstance(idx, (list, tuple)):
            idxs = [int(v) for v in idx]
        else:
            idxs = [int(idx or 0)]
        if seed is not None:
            # rank-identical noise only where the ranks cooperate on one image; its seed is offset by the image index
            # (data["Index"] when the loader provides it) so that images stay independent of each other and of the world size
            self.netG.noise_seed = seed if sharded else None
            self.netG.noise_index = idxs[0]
        base = getattr(self, "image_seed_base", None)
        if base is not None:
            # sr.py --batch: every image of the batch draws the in-kernel noise stream of ITS index, with counters local to the image -
            # the same noise whether it is restored alone, in a batch of 16 or on another rank (identical on all ranks of a sharded one)
            if len(idxs) != sr.shape[0]:
                raise ValueError("data['Index'] must hold one index per image of the batch")
            self.netG.sample_seeds = [int(base) + 1000003 * i for i in idxs]
        else:
            self.netG.sample_seeds = None
        try:
            with torch.no_grad():
                out = self.netG.super_resolution(sr, continous)
        finally:
            self.netG.sample_seeds = None
        self.SR = out[..., pd:-pd, pd:-pd]

    def set_new_noise_schedule(self, schedule_opt, schedule_phase="train"):
        if getattr(self, "schedule_phase", None) != schedule_phase:
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(dict(schedule_opt), self.device)

    def get_current_visuals(self, need_LR=True, sample=False):
        out = OrderedDict()
        out["SR"] = self.SR.detach().float().cpu()
        out["INF"] = self.data["SR"].detach().float().cpu()
        out["HR"] = self.data["HR"].detach().float().cpu()
        out["LR"] = self.data["LR"].detach().float().cpu() if need_LR and "LR" in self.data else out["INF"]
        return out

    def visuals_u8(self, j=0):
        """uint8 HWC images of the val loop (final SR, HR, LR, INF = predictor output) of image ``j`` of the batch, converted ON THE
        DEVICE: only 4 x H x W x 3 bytes cross PCIe instead of the 11 fp32 snapshots of get_current_visuals.
        (``continous=True`` stacks the snapshots along dim 0 in blocks of B images: the final ones are the last block.)"""
        from .metrics import tensor2img_u8_device as cv
        B = self.data["SR"].shape[0]
        sr = self.SR[self.SR.shape[0] - B + j] if self.SR.dim() == 4 else self.SR
        lr = self.data["LR"] if "LR" in self.data else self.data["SR"]
        out = OrderedDict(SR=cv(sr), HR=cv(self.data["HR"][j]), LR=cv(lr[j]))
        pre = getattr(self.netG, "pre_initx", None)
        out["INF"] = cv(pre[j, :, 64:-64, 64:-64]) if pre is not None else cv(self.data["SR"][j])
        return out

    def load_network(self):
        """model/model.py:224-251: in val phase with EMA on, ``{prefix}_gen_ema.pth`` is loaded strict=False."""
        prefix = self.opt["path"]["resume_state"]
        if not prefix:
            return
        use_ema = self.opt["train"]["ema_scheduler"] and self.opt["train"]["ema_scheduler"]["use"] \
            and self.opt["phase"] == "val"
        path = "{}_gen_ema.pth".format(prefix) if use_ema else "{}_gen.pth".format(prefix)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        logger.info("Loading pretrained model for G [{:s}] ...".format(path))
        sd = torch.load(path, map_location="cpu")
        report = load_checkpoint_state(self.netG, sd, strict=not use_ema and not self.opt["model"].get("finetune_norm"))
        logger.info("checkpoint %s: %d tensors loaded, %d schedule buffers skipped", path, report["loaded"], len(report["skipped_buffers"]))


SCHEDULE_BUFFERS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def load_checkpoint_state(netG, sd, strict=False):
    """``network.load_state_dict`` of model/model.py:236-243 with the result inspected.

    * a leading ``module.`` (DataParallel / DDP wrapper) is stripped;
    * the twelve schedule buffers (saved at training length, re-created by ``set_new_noise_schedule``) are skipped;
    * EVERY ``denoise_fn.*`` / ``predictor.*`` parameter of ``netG`` must be present with the right shape and no
      unknown ``denoise_fn.*`` / ``predictor.*`` key may remain: a checkpoint of another architecture or with renamed
      keys raises instead of leaving the random initialisation in place (the reference's strict=False is silent);
    * ``strict`` additionally rejects any other unexpected key (the reference's non-EMA path, strict = not finetune_norm).
    """
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    skipped = [k for k in sd if k in SCHEDULE_BUFFERS]
    sd = {k: v for k, v in sd.items() if k not in SCHEDULE_BUFFERS}
    own = dict(netG.named_parameters())
    need = [k for k in own if k.startswith(("denoise_fn.", "predictor."))]
    missing = [k for k in need if k not in sd]
    bad_shape = [k for k in need if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
    unknown = [k for k in sd if k not in own]
    unknown_net = [k for k in unknown if k.startswith(("denoise_fn.", "predictor."))]
    if missing or bad_shape or unknown_net or (strict and unknown):
        def head(v):
            return ", ".join(v[:6]) + (" ..." if len(v) > 6 else "")
        raise RuntimeError("checkpoint does not match the network: missing [%s]; wrong shape [%s]; unexpected [%s]"
                           % (head(missing), head(bad_shape), head(unknown_net if not strict else unknown)))
    for k in unknown:
        logger.warning("checkpoint key ignored: %s", k)
    res = netG.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return {"loaded": len(sd) - len(unknown), "skipped_buffers": skipped, "ignored": unknown,
            "missing_non_network": list(res.missing_keys)}


def create_model(opt, device=None):
    m = DDPM(opt, device)
    logger.info("Model [{:s}] is created.".format(m.__class__.__name__))
    return m
