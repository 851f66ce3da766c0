"""Inter-step patch splitting (reference: utils/util.py:108-146, called from model/ucdir.py:298-300).

Unlike the reference, which walks the windows one by one on a single GPU, the windows of one
denoising step are independent, so they are evaluated as one batch — and, when a process group is
given, sharded across ranks with a single all-gather of the window outputs per step.
"""
import torch
import torch.nn.functional as F


def _check_windows(wins, Hp, Wp, skip):
    """The gather kernel trusts the device-side (h0, w0) list (round-4 advice): every window is skip x skip and lies inside the
    padded canvas, or the call is refused before anything is uploaded."""
    for (a, b, c, d) in wins:
        if not (b - a == skip and d - c == skip and 0 <= a and b <= Hp and 0 <= c and d <= Wp):
            raise ValueError(f"patch window {(a, b, c, d)} is not a {skip} x {skip} window of the {Hp} x {Wp} padded canvas")


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


def _guide_chunks(guide, pd, chunks, cache):
    """Padded guide windows, cut once per restoration (the guide does not change over the steps).  ``cache`` is a dict
    owned by the CALLER (DY3h keeps one per module and clears it at the end of a sampling loop): nothing global keeps
    hundreds of MB of windows alive, and two modules never evict each other."""
    key = (guide.data_ptr(), guide._version, tuple(guide.shape), pd, tuple(map(tuple, chunks)))
    hit = cache.get("guide")
    if hit is not None and hit[0] == key:
        return hit[1]
    gp = F.pad(guide, (pd, pd, pd, pd), mode="reflect")
    out = [torch.cat([gp[..., a:b, c:d] for (a, b, c, d) in ch], dim=0).contiguous() for ch in chunks]
    cache["guide"] = (key, out, guide)               # one entry; keeps `guide` alive so the key cannot be recycled
    return out


def _buffer(cache, name, shape, like):
    """Persistent work buffer of a restoration (gather slots, denoised canvas): allocated once, reused every step."""
    key = (tuple(shape), like.dtype, like.device)
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, torch.zeros(shape, dtype=like.dtype, device=like.device))
        cache[name] = hit
    return hit[1]


def patch_forward_guide(noisy, net, params, skip=512, padding=32, group=None, max_batch=8, cache=None, force_gather=False,
                        timers=None):
    """Same result as the reference's sequential loop; ``net(x, time=..., guide=...)`` is called on
    batches of windows.  noisy (B,6,H,W); params = {'time': (B,1), 'guide': (B,3,H,W)}."""
    if cache is None:
        cache = {}
    B = noisy.shape[0]
    pd = patch_pad(noisy.shape[-2], noisy.shape[-1], skip, padding)
    # on the device the window batches are gathered straight from the un-padded canvas by one kernel per engine call
    # (ucdir_gather_windows: reflect indexing in the kernel, no padded copy, no per-window cat); CPU tensors (the gloo tests'
    # stub engines) take the reference's own route, F.pad + slices
    on_dev = noisy.is_cuda and noisy.dtype == torch.float32
    if on_dev:
        noisy = noisy.contiguous()
        xp = noisy
        H, W = noisy.shape[-2] + 2 * pd, noisy.shape[-1] + 2 * pd
    else:
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
        gbs = _guide_chunks(params["guide"], pd, chunks, cache)
        for chunk, real, gb in zip(chunks, reals, gbs):
            if on_dev:
                from .ucdir import gather_windows
                wkey = ("win", tuple(chunk), noisy.device)
                wd = cache.get(wkey)
                if wd is None:                                       # (h0, w0) of the chunk's windows on the device: once per restoration
                    _check_windows(chunk, noisy.shape[-2] + 2 * pd, noisy.shape[-1] + 2 * pd, skip)
                    wd = torch.tensor([[a, c] for (a, b, c, d) in chunk], dtype=torch.int32, device=noisy.device)
                    cache[wkey] = wd
                xb = gather_windows(noisy, pd, wd, skip)
            else:
                xb = torch.cat([xp[..., a:b, c:d] for (a, b, c, d) in chunk], dim=0).contiguous()
            tb = params["time"].repeat(len(chunk), 1)
            o = net(xb, tb, gb)
            outs.append(o[:real * B, :, padding:-padding, padding:-padding])
    inner = skip - 2 * padding
    if outs:
        local = torch.cat(outs, dim=0)
    else:
        local = xp.new_zeros((0, 3, inner, inner))
    if world > 1 or (force_gather and group is not None):
        # ONE collective per step into buffers that live for the whole restoration (fixed-size slots: `per` windows per rank)
        import torch.distributed as dist
        slot = _buffer(cache, "slot", (per * B, 3, inner, inner), xp)
        allo = _buffer(cache, "gathered", (world * per * B, 3, inner, inner), xp)
        slot[:local.shape[0]] = local
        if timers is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_gather_into_tensor(allo, slot, group=group)
        if timers is not None:
            ev[1].record()
            timers.append(ev)
    else:
        allo = local
    den = _buffer(cache, "den", (B, 3, H, W), xp)      # every interior pixel is overwritten below; the border is cropped
    for k, (a, b, c, d) in enumerate(wins):        # reference order: later windows overwrite earlier ones
        r, q = divmod(k, per)
        base = (r * per + q) * B
        den[..., a + padding:b - padding, c + padding:d - padding] = allo[base:base + B]
    # a fresh tensor, like the reference (utils/util.py:108-146): `den` is a module-owned buffer that the next forward of the
    # same shape overwrites, so a view into it would silently alias two results (round-3 advice)
    return den[..., pd:-pd, pd:-pd].clone()
