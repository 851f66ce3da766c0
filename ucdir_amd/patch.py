"""Inter-step patch splitting (reference: utils/util.py:108-146, called from model/ucdir.py:298-300).

Unlike the reference, which walks the windows one by one on a single GPU, the windows of one
denoising step are independent, so they are evaluated as one batch — and, when a process group is
given, sharded across ranks with a single all-gather of the window outputs per step.
"""
import torch
import torch.nn.functional as F


def patch_windows(H, W, skip, padding):
    """Window list of utils/util.py:119-137 for a padded canvas H x W, in evaluation order."""
    shift = skip - 2 * padding
    out = []
    for i in range(0, H, shift):
        for j in range(0, W, shift):
            h0, h1, w0, w1 = i, i + skip, j, j + skip
            if h1 > H:
                h1, h0 = H, H - skip
            if w1 > W:
                w1, w0 = W, W - skip
            out.append((h0, h1, w0, w1))
    return out


def patch_pad(H, W, skip, padding):
    pd = min(H, W)
    return skip - pd + padding if pd < skip else padding


_GUIDE_WINDOWS = {}      # the guide does not change over the steps of a restoration: its padded windows are cut once


def _guide_chunks(guide, pd, chunks):
    key = (guide.data_ptr(), guide._version, tuple(guide.shape), pd, tuple(map(tuple, chunks)))
    hit = _GUIDE_WINDOWS.get("k")
    if hit is not None and hit[0] == key:
        return hit[1]
    gp = F.pad(guide, (pd, pd, pd, pd), mode="reflect")
    out = [torch.cat([gp[..., a:b, c:d] for (a, b, c, d) in ch], dim=0).contiguous() for ch in chunks]
    _GUIDE_WINDOWS["k"] = (key, out, guide)          # one entry; keeps `guide` alive so the key cannot be recycled
    return out


def patch_forward_guide(noisy, net, params, skip=512, padding=32, group=None, max_batch=8):
    """Same result as the reference's sequential loop; ``net(x, time=..., guide=...)`` is called on
    batches of windows.  noisy (B,6,H,W); params = {'time': (B,1), 'guide': (B,3,H,W)}."""
    B = noisy.shape[0]
    pd = patch_pad(noisy.shape[-2], noisy.shape[-1], skip, padding)
    xp = F.pad(noisy, (pd, pd, pd, pd), mode="reflect")
    _, _, H, W = xp.shape
    wins = patch_windows(H, W, skip, padding)
    rank, world = 0, 1
    if group is not None:
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    per = (len(wins) + world - 1) // world
    mine = wins[rank * per:(rank + 1) * per]
    outs = []
    if mine:
        # equal chunks, the last one padded with a repeat of its last window: every engine call of every step has the
        # same batch, so the engine plans its (multi-GB) workspace once instead of re-planning when the batch alternates
        nchunk = (len(mine) + max_batch - 1) // max_batch
        size = (len(mine) + nchunk - 1) // nchunk
        chunks, reals = [], []
        for s in range(0, len(mine), size):
            chunk = mine[s:s + size]
            reals.append(len(chunk))
            chunks.append(chunk + [chunk[-1]] * (size - len(chunk)))
        gbs = _guide_chunks(params["guide"], pd, chunks)
        for chunk, real, gb in zip(chunks, reals, gbs):
            xb = torch.cat([xp[..., a:b, c:d] for (a, b, c, d) in chunk], dim=0).contiguous()
            tb = params["time"].repeat(len(chunk), 1)
            o = net(xb, tb, gb)
            outs.append(o[:real * B, :, padding:-padding, padding:-padding])
    inner = skip - 2 * padding
    if outs:
        local = torch.cat(outs, dim=0)
    else:
        local = xp.new_zeros((0, 3, inner, inner))
    if world > 1:
        import torch.distributed as dist
        slot = xp.new_zeros((per * B, 3, inner, inner))
        slot[:local.shape[0]] = local
        gathered = [torch.empty_like(slot) for _ in range(world)]
        dist.all_gather(gathered, slot, group=group)
        allo = torch.cat(gathered, dim=0)
    else:
        allo = local
    den = torch.zeros_like(xp)[:, :3]
    for k, (a, b, c, d) in enumerate(wins):        # reference order: later windows overwrite earlier ones
        r, q = divmod(k, per)
        base = (r * per + q) * B
        den[..., a + padding:b - padding, c + padding:d - padding] = allo[base:base + B]
    return den[..., pd:-pd, pd:-pd]
