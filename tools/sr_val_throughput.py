"""Throughput of the `sr.py -p val` entry point on synthetic pairs (round-5 verdict item 5):
    python tools/sr_val_throughput.py [n_images] [size] > profiles/r06_sr_val.json
n same-sized PNG pairs through the full SID configuration (config/sid.yaml, synthetic weights), once grouped (--batch 16) and once one by one
(--batch 1, HIP-graph replay): images per second of the restoration itself (DDPM.test: reflect-pad 64, predictor, 50 steps, crop - the UNet
computes at (size + 128 -> next multiple of 32)^2), excluding PNG decode / JPEG encode / metrics, and the wall time of the whole loop."""
import importlib.util
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import yaml
    from PIL import Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    out = {"images": n, "size": size, "config": "config/sid.yaml (full SID UNet, T = 50), synthetic weights", "runs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        rs = np.random.RandomState(0)
        for d in ("lq", "gt"):
            os.makedirs(os.path.join(tmp, d))
        for i in range(n):
            gt = (rs.rand(size // 8, size // 8, 3) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)
            Image.fromarray(gt).save(os.path.join(tmp, "gt", f"{i:03d}.png"))
            Image.fromarray((gt * 0.25).astype(np.uint8)).save(os.path.join(tmp, "lq", f"{i:03d}.png"))
        cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "sid.yaml")))
        cfg["datasets"]["val"]["data_args"]["dataroot"] = {"lq": os.path.join(tmp, "lq"), "gt": os.path.join(tmp, "gt")}
        yaml.safe_dump(cfg, open(os.path.join(tmp, "sid.yaml"), "w"))
        spec = importlib.util.spec_from_file_location("sr_entry_tp", os.path.join(ROOT, "sr.py"))
        sr = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sr)
        for tag, batch in (("batch16", 16), ("batch1_graph", 1)):
            wd = os.path.join(tmp, tag)
            os.makedirs(wd)
            os.chdir(wd)
            t0 = time.perf_counter()
            psnr, ssim = sr.main(["-p", "val", "-c", os.path.join(tmp, "sid.yaml"), "--synthetic-weights", "--batch", str(batch), "--seed", "1"])
            wall = time.perf_counter() - t0
            nr, tr = sr.main.last_throughput
            gs = sr.main.last_groups
            steady = gs[1:] if len(gs) > 1 else gs               # the first DDPM.test call packs and uploads the weights and plans the shape
            out["runs"][tag] = {"batch": batch, "restore_images_per_s": nr / tr, "restore_s": tr,
                                "steady_state_images_per_s": sum(g[0] for g in steady) / sum(g[1] for g in steady),
                                "first_call_s": gs[0][1], "calls": len(gs), "loop_wall_s": wall,
                                "loop_images_per_s": n / wall, "psnr": psnr, "ssim": ssim}
        os.chdir(ROOT)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
