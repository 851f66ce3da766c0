"""Minimal paired-image loader for ``-p val`` (reference: data/LRHR_dataset.py:230-297 PairDataset,
val split, datatype img): sorted file lists of dataroot.lq / dataroot.gt, RGB, scaled to [-1, 1]
(data/util.py:76-83), dict with 'HR', 'SR', 'LR', 'Index'."""
import os

import numpy as np
import torch

IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")


def _listdir(d):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.lower().endswith(IMG_EXT))


def _load(path):
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a.transpose(2, 0, 1)) * 2.0 - 1.0


class PairDataset:
    def __init__(self, data_args, phase="val"):
        root = data_args["dataroot"]
        self.sr_path = _listdir(root["lq"])
        self.hr_path = _listdir(root["gt"])
        if len(self.sr_path) != len(self.hr_path):
            raise ValueError("lq / gt directories hold a different number of images")
        n = data_args.get("data_len", -1) if hasattr(data_args, "get") else -1
        if n and n > 0:
            self.sr_path, self.hr_path = self.sr_path[:n], self.hr_path[:n]
        self.crop = data_args.get("crop_size", None) if phase == "val_crop" else None

    def __len__(self):
        return len(self.sr_path)

    def __getitem__(self, i):
        sr, hr = _load(self.sr_path[i]), _load(self.hr_path[i])
        return {"HR": hr, "SR": sr, "LR": sr, "Index": i}
