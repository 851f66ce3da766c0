"""``DDPM`` model wrapper for validation (reference: model/model.py:35-267, model/__init__.py:5-10).

Only the inference surface ``sr.py -p val`` uses: device placement, EMA checkpoint loading,
``feed_data`` / ``test`` (reflect-pad 64, restore, crop; model/model.py:124-138),
``set_new_noise_schedule`` and ``get_current_visuals``.  One process per GPU; no DDP wrapper is
needed for sampling (the reference's wrapper only broadcasts parameters).
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

    def feed_data(self, data):
        self.data = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data.items()}

    def test(self, continous=False):
        self.netG.eval()
        pd = 64
        sr = F.pad(self.data["SR"], (pd, pd, pd, pd), mode="reflect")
        with torch.no_grad():
            out = self.netG.super_resolution(sr, continous)
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
        # schedule buffers saved at training length (2000) are re-created by set_new_noise_schedule
        sd = {k: v for k, v in sd.items() if "." in k}
        self.netG.load_state_dict(sd, strict=False)


def create_model(opt, device=None):
    m = DDPM(opt, device)
    logger.info("Model [{:s}] is created.".format(m.__class__.__name__))
    return m
