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
import time

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
    parser.add_argument("--batch", type=int, default=16,
                        help="restore up to this many same-sized val images per DDPM.test call (1: the reference's one-by-one loop)")
    parser.add_argument("--seed", type=int, default=None, help="base of the per-image noise seeds (default: one random draw per run)")
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
    dn = diffusion.netG.denoise_fn
    thr = dn.patch_threshold
    # Per-image noise streams: image i draws Philox(seed_base + 1000003 i) with counters local to the image, whatever batch or rank it is
    # restored in.  --seed fixes the base; without it one draw from torch's CPU generator per run (torch.manual_seed makes it reproducible).
    diffusion.image_seed_base = args.seed if args.seed is not None else int(torch.randint(0, 2 ** 31, (1,)).item())
    t_restore, n_restored = 0.0, 0
    group_times = []                                              # (images, seconds) of every DDPM.test call (the first one packs the weights)

    def restore(group):
        """One DDPM.test call for a group of images of identical (H, W): a batch is B independent restorations (model/diffusion.py:185-211
        is written for a batch; the reference's val loader feeds it batch_size 1, data/__init__.py:47)."""
        nonlocal tot_psnr, tot_ssim, n, t_restore, n_restored
        items = [g[1] for g in group]
        data = {k: torch.stack([it[k] for it in items]) for k in ("HR", "SR", "LR") if k in items[0]}
        data["Index"] = [g[0] for g in group]                    # DDPM.test derives every image's noise stream from its index
        small = (items[0]["SR"].shape[-2] + 128) * (items[0]["SR"].shape[-1] + 128) <= thr
        dn.set_graph(len(group) == 1 and small)                  # batch-1 remainders: HIP-graph replay of the forward (latency path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            diffusion.feed_data(data)
            diffusion.test(continous=True)
        torch.cuda.synchronize()
        t_restore += time.perf_counter() - t0
        n_restored += len(group)
        group_times.append((len(group), time.perf_counter() - t0))
        if group[0][2] and rank != 0:
            return                                                # sharded image: every rank holds the same result; rank 0 reports it
        name = opt["name"]
        for j, (i, _, _) in enumerate(group):
            fname = os.path.splitext(os.path.basename(val_set.sr_path[i]))[0]
            vis = diffusion.visuals_u8(j)
            hr_img, lr_img, fake_img, sr_img = vis["HR"], vis["LR"], vis["INF"], vis["SR"]
            Metrics.save_jpg(sr_img, "{}/{}_{}_sr.png".format(result_path, fname, name))
            Metrics.save_jpg(hr_img, "{}/{}_{}_hr.png".format(result_path, fname, name))
            Metrics.save_jpg(lr_img, "{}/{}_{}_lr.png".format(result_path, fname, name))
            Metrics.save_jpg(fake_img, "{}/{}_{}_inf.png".format(result_path, fname, name))
            tot_psnr += Metrics.calculate_psnr(sr_img, hr_img)
            tot_ssim += Metrics.calculate_ssim(sr_img, hr_img)
            n += 1
            logger.info("val index %d" % i)

    pending = {}                                                  # (H, W) -> images of this rank waiting for a full batch (first-seen order)
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
        if shared or args.batch <= 1 or (h + 128) * (w + 128) > thr:
            restore([(i, item, shared)])                          # patch-split images (1024^2 windows as engine batches) go one at a time
            continue
        grp = pending.setdefault((h, w), [])
        grp.append((i, item, False))
        if len(grp) >= args.batch:
            restore(pending.pop((h, w)))
    for key in list(pending):
        restore(pending.pop(key))
    dn.set_graph(False)
    if n_restored:
        logger.info("restored %d images in %.2f s on this rank (%.2f img/s, batches of up to %d)" % (n_restored, t_restore, n_restored / t_restore, args.batch))
    main.last_throughput = (n_restored, t_restore)
    main.last_groups = group_times
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
