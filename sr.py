#!/usr/bin/env python
"""``sr.py -p val -c config/sid.yaml --checkpoint <prefix>`` — validation entry point.

Counterpart of the reference's ``sr.py`` val branch (sr.py:320-400, 505-586): same flags, same YAML
schema, same per-image outputs ``{results}/{fname}_{name}_{sr,hr,lr,inf}.jpg`` and the
``# Validation # PSNR/SSIM`` log lines.  One process per GPU; with N ranks
(``python -m torch.distributed.run --nproc-per-node N sr.py ...``):
  * images small enough for one denoiser call are strided over ranks like the reference's EnlargedSampler
    (data/data_sampler.py:44-45), no data-path collective;
  * images that take the inter-step patch split (padded area > 1024^2, model/ucdir.py:298) are restored by ALL ranks
    together: the windows of every step are sharded over the ranks with one RCCL all-gather per step
    (ucdir_amd/patch.py), every rank draws the same noise; rank 0 writes the outputs.
Training (``-p train``) is out of scope for this build.

Without a checkpoint (none ships with the reference) ``--synthetic-weights`` fills the network with
the deterministic generator used by the tests, so the plumbing can be exercised end to end.
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ucdir_amd import config as Config  # noqa: E402
from ucdir_amd import metrics as Metrics  # noqa: E402
from ucdir_amd import model as Model  # noqa: E402
from ucdir_amd.data import PairDataset  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, default="config/sid.yaml")
    parser.add_argument("-p", "--phase", type=str, choices=["train", "val"], default="val")
    parser.add_argument("-gpu", "--gpu_ids", type=str, default=None)
    parser.add_argument("-debug", "-d", action="store_true")
    parser.add_argument("-enable_wandb", action="store_true")
    parser.add_argument("-log_wandb_ckpt", action="store_true")
    parser.add_argument("-log_eval", action="store_true")
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("-launcher", default="pytorch")
    parser.add_argument("--checkpoint", type=str, default=None)
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="fill netG with the deterministic test weights (no checkpoint ships with the reference)")
    parser.add_argument("--max-images", type=int, default=-1)
    args = parser.parse_args(argv)
    if args.phase != "val":
        raise SystemExit("only -p val is implemented (sampling path); training is out of scope of this build")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(args.local_rank)))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    opt = Config.parse(args, world_size=world)
    opt["rank"], opt["world_size"] = rank, world
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.ERROR, format="%(asctime)s %(message)s")
    logger = logging.getLogger("base")
    fh = logging.FileHandler(os.path.join(opt["path"]["log"], "val.log"))
    logging.getLogger("val").addHandler(fh)

    val_set = PairDataset(opt["datasets"]["val"]["data_args"], phase="val")
    if args.synthetic_weights:
        opt["path"]["resume_state"] = None
    diffusion = Model.create_model(opt)
    if args.synthetic_weights:
        from ucdir_amd.weights import synth_state_dict
        sd = synth_state_dict(diffusion.netG.denoise_fn.cfg, 0)
        Model.load_checkpoint_state(diffusion.netG, {k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    diffusion.set_new_noise_schedule(opt["model"]["beta_schedule"]["val"], schedule_phase="val")

    logger.info("Begin Model Evaluation. len %d" % len(val_set))
    result_path = opt["path"]["results"]
    os.makedirs(result_path, exist_ok=True)
    tot_psnr = tot_ssim = 0.0
    n = 0
    idxs = list(range(len(val_set)))
    if args.max_images > 0:
        idxs = idxs[:args.max_images * world]
    thr = diffusion.netG.denoise_fn.patch_threshold
    nsmall = 0
    for i in idxs:
        item = val_set[i]
        h, w = item["SR"].shape[-2:]
        shared = world > 1 and (h + 128) * (w + 128) > thr        # DDPM.test pads 64 per side: this image is patch-split
        if not shared:
            mine = (nsmall % world) == rank
            nsmall += 1
            if not mine:
                continue
        fname = os.path.splitext(os.path.basename(val_set.sr_path[i]))[0]
        data = {k: (v.unsqueeze(0) if torch.is_tensor(v) else v) for k, v in item.items()}
        data["Index"] = i                                         # DDPM.test offsets the rank-identical noise seed by the image index
        with torch.no_grad():
            diffusion.feed_data(data)
            diffusion.test(continous=True)
        if shared and rank != 0:
            continue                                              # every rank holds the same result; rank 0 reports it
        vis = diffusion.visuals_u8()
        hr_img, lr_img, fake_img, sr_img = vis["HR"], vis["LR"], vis["INF"], vis["SR"]
        name = opt["name"]
        Metrics.save_jpg(sr_img, "{}/{}_{}_sr.png".format(result_path, fname, name))
        Metrics.save_jpg(hr_img, "{}/{}_{}_hr.png".format(result_path, fname, name))
        Metrics.save_jpg(lr_img, "{}/{}_{}_lr.png".format(result_path, fname, name))
        Metrics.save_jpg(fake_img, "{}/{}_{}_inf.png".format(result_path, fname, name))
        tot_psnr += Metrics.calculate_psnr(sr_img, hr_img)
        tot_ssim += Metrics.calculate_ssim(sr_img, hr_img)
        n += 1
        logger.info("val index %d" % i)
    acc = torch.tensor([tot_psnr, tot_ssim, float(n)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(acc)
    avg_psnr, avg_ssim = (acc[0] / acc[2]).item(), (acc[1] / acc[2]).item()
    logger.info("# Validation # PSNR: {:.4e}".format(avg_psnr))
    logger.info("# Validation # SSIM: {:.4e}".format(avg_ssim))
    logging.getLogger("val").info("psnr: {:.4e}, ssim: {:.4e}".format(avg_psnr, avg_ssim))
    if world > 1:
        dist.destroy_process_group()
    return avg_psnr, avg_ssim


if __name__ == "__main__":
    main()
