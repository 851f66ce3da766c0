"""Image conversion and full-reference metrics of the val loop (reference: core/metrics.py).

cv2 / torchvision are not available here: JPEG writing uses PIL (quality 100 like
core/metrics.py:42-45), SSIM uses scipy's correlate with the same 11x11 sigma-1.5 Gaussian window
and valid-region crop as core/metrics.py:58-99.
"""
import math

import numpy as np


def tensor2img_u8_device(tensor, min_max=(-1, 1)):
    """tensor2img for ONE image that lives on the GPU: clamp / rescale / round / HWC on the device, then a uint8 copy
    (4x fewer bytes over PCIe than the reference's fp32 ``.cpu()``; same arithmetic, same rounding: torch.round and
    numpy.round are both round-half-to-even)."""
    t = tensor.detach().squeeze().float().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() != 3:
        raise ValueError("tensor2img_u8_device takes one (3,H,W) image")
    return (t * 255.0).round().to(__import__("torch").uint8).permute(1, 2, 0).contiguous().cpu().numpy()


def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    """core/metrics.py:8-34 for 3-D / single-image 4-D tensors: clamp, rescale to [0,1], HWC, round to uint8."""
    t = tensor.squeeze().float().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 4:          # a stack of images: tile them in a row (make_grid is unavailable)
        t = torch_hcat(t)
    img = t.numpy()
    if img.ndim == 3:
        img = np.transpose(img, (1, 2, 0))
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def torch_hcat(t):
    import torch
    return torch.cat(list(t), dim=-1)


def save_jpg(img, img_path, mode="RGB"):
    from PIL import Image
    Image.fromarray(img).save(img_path.replace(".png", ".jpg"), quality=100, subsampling=0)


def save_img(img, img_path, mode="RGB"):
    from PIL import Image
    Image.fromarray(img).save(img_path)


def calculate_psnr(img1, img2):
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _ssim(img1, img2):
    from scipy.ndimage import correlate
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    ax = np.arange(11) - 5
    k = np.exp(-(ax ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    window = np.outer(k, k)
    f = lambda a: correlate(a, window, mode="reflect")[5:-5, 5:-5]
    mu1, mu2 = f(img1), f(img2)
    s1 = f(img1 ** 2) - mu1 ** 2
    s2 = f(img2 ** 2) - mu2 ** 2
    s12 = f(img1 * img2) - mu1 * mu2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean()


def calculate_ssim(img1, img2):
    if img1.ndim == 2:
        return _ssim(img1, img2)
    return float(np.mean([_ssim(img1[..., c], img2[..., c]) for c in range(img1.shape[2])]))
