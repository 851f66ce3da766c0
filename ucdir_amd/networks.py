"""``define_G`` — the drop-in boundary (reference: model/networks.py:88-95)."""
import logging

from . import diffusion, ucdir

logger = logging.getLogger("base")


def define_G(opt):
    model_opt = opt["model"]
    if model_opt["which_model_G"] == "ucdir":
        unet_args = dict(model_opt["unet"])
        model = getattr(ucdir, model_opt["unet_name"])(**unet_args)
        netG = getattr(diffusion, model_opt["diffusion_name"])(model, **dict(model_opt["diffusion"]))
        logger.info("**model net G %s %s" % (model.__class__.__name__, netG.__class__.__name__))
        return netG
    raise NotImplementedError(model_opt["which_model_G"])
