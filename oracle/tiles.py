"""Oracle for chop_forward tiling + overlap blend (test infrastructure).

Restates reference utils/utils.py:318-445 and run.py:167-202 from the tile
geometry rather than from unfold views: tile origins are computed explicitly
(numpy integers), tiles are plain slices, and the blend is the same sequence of
fp32 operations (product, slice +=, final divide) in the same (h, w) order, so
fp32 results are bit-identical to the reference.
"""
import numpy as np
import torch


def _axis_origins(size, patch, step_int):
    """Origins of sliding windows along one axis: a regular lattice of stride
    step_int plus, when (size-patch) is not a multiple of it, one ragged window
    anchored at size-patch (utils.py:354-362)."""
    n = (size - patch) // step_int + 1
    o = [i * step_int for i in range(n)]
    if (size - patch) % step_int != 0:
        o.append(size - patch)
    return o


def chop_geometry(height, width, patch_size=200, step=0.5):
    """Tile geometry of Model.chop_forward (run.py:176-181, utils.py:350-365).

    Returns (ps, ys, xs): clamped patch size and the row / column origins.
    Tiles are row-major: tile k = (ys[k // len(xs)], xs[k % len(xs)])."""
    ps = min(height, width, patch_size)                      # run.py:176
    step_int = int(ps * step)                                 # utils.py:351-352
    return ps, _axis_origins(height, ps, step_int), _axis_origins(width, ps, step_int)


def extract_patches_2d(img, patch_shape, step=(0.5, 0.5), batch_first=False):
    """utils.py:318-369 for the un-padded case (patch <= image, which
    chop_forward guarantees via run.py:176).  img [B,C,H,W] -> [n,B,C,ph,pw]
    (or [B,n,C,ph,pw] when batch_first)."""
    ph, pw = patch_shape
    B, C, H, W = img.shape
    assert H >= ph and W >= pw, "pad branch of the reference is unreachable from chop_forward"
    sh = int(ph * step[0]) if isinstance(step[0], float) else step[0]
    sw = int(pw * step[1]) if isinstance(step[1], float) else step[1]
    ys = _axis_origins(H, ph, sh)
    xs = _axis_origins(W, pw, sw)
    tiles = [img[:, :, y:y + ph, x:x + pw] for y in ys for x in xs]
    out = torch.stack(tiles, 0)
    return out.permute(1, 0, 2, 3, 4) if batch_first else out


def blend_profile(patch_size, step, scale, dtype=torch.float32):
    """1-D blending profile of recompose_tensor (utils.py:396,413-416)."""
    overlap = scale * int(round((1.0 - step) * (patch_size / scale)))
    return torch.cat([torch.linspace(0.1, 1.0, overlap, dtype=dtype),
                      torch.ones(patch_size - 2 * overlap, dtype=dtype),
                      torch.linspace(1.0, 0.1, overlap, dtype=dtype)], 0)


def recompose_origins(n_h_full, n_w_full, patch, step):
    """Placement origins used by the blend (utils.py:398,405-410,425-426).
    n_*_full: full output extent (scale*height / scale*width)."""
    eff = int(step * patch)
    step_int = int(patch * step)
    H = max(n_h_full, patch)
    W = max(n_w_full, patch)
    nh = 1 + (H - patch) // step_int + (1 if (H - patch) % step_int != 0 else 0)
    nw = 1 + (W - patch) // step_int + (1 if (W - patch) % step_int != 0 else 0)
    ys = [min(h * eff, n_h_full - patch) for h in range(nh)]
    xs = [min(w * eff, n_w_full - patch) for w in range(nw)]
    return ys, xs


def recompose_tensor(patches, height, width, step=0.5, scale=1):
    """utils.py:372-445: weighted overlap-add of [n,C,P,P] tiles and divide by
    the accumulated weights.  Same op order as the reference (fp32-bit-exact)."""
    assert isinstance(step, float) and 0.5 <= step <= 1.0
    full_h, full_w = scale * height, scale * width
    n, C, P, _ = patches.shape
    prof = blend_profile(P, step, scale, dtype=patches.dtype)
    wpatch = prof[None].repeat(P, 1) * prof[:, None].repeat(1, P)     # :418-420
    ys, xs = recompose_origins(full_h, full_w, P, step)
    per_img = len(ys) * len(xs)
    nb = n // per_img
    den = torch.zeros(1, C, full_h, full_w, dtype=patches.dtype)
    for y in ys:
        for x in xs:
            den[0, :, y:y + P, x:x + P] += wpatch[None]
    num = torch.zeros(nb, C, full_h, full_w, dtype=patches.dtype)
    k = 0
    for b in range(nb):
        for y in ys:
            for x in xs:
                num[b, :, y:y + P, x:x + P] += patches[k] * wpatch
                k += 1
    return num / den


def chop_forward(model_fn, data, scale, patch_size=200, step=0.5):
    """Model.chop_forward (run.py:167-202): serial batch-1 loop over tiles."""
    B, C, H, W = data.shape
    ps = min(H, W, patch_size)
    tiles = extract_patches_2d(data, (ps, ps), [step, step], batch_first=True).squeeze(0)
    outs = [model_fn(tiles[p:p + 1]) for p in range(tiles.shape[0])]
    return recompose_tensor(torch.cat(outs, 0), H, W, step=step, scale=scale)


def chop_forward_window(model_fn, data_fn, height, width, scale, window, patch_size=200, step=0.5, cache=None):
    """chop_forward (run.py:167-202 + utils.py:372-445) evaluated on ONE output window only: window = (y0, y1, x0, x1) in output pixels.  The tile origins,
    the blend profile and the (h, w) order of the `+=` are those of recompose_tensor above, but only the tiles whose output extent meets the window are run
    (model_fn on data_fn(y, y + ps, x, x + ps), the tile's input crop [1,C,ps,ps]) and only the window's part of num / den is kept: the result equals
    chop_forward(...)[:, :, y0:y1, x0:x1] bit for bit at a cost of <= a handful of forwards (tests at BASELINE's full sizes, where the whole frame would be
    3268 forwards).  cache: optional dict tile index -> model output, shared between calls."""
    ps, ys_in, xs_in = chop_geometry(height, width, patch_size, step)
    P = ps * scale
    full_h, full_w = scale * height, scale * width
    y0, y1, x0, x1 = window
    assert 0 <= y0 < y1 <= full_h and 0 <= x0 < x1 <= full_w
    ys, xs = recompose_origins(full_h, full_w, P, step)
    assert len(ys) == len(ys_in) and len(xs) == len(xs_in)
    prof = blend_profile(P, step, scale)
    wpatch = prof[None].repeat(P, 1) * prof[:, None].repeat(1, P)
    num = den = None
    for iy, y in enumerate(ys):
        for ix, x in enumerate(xs):
            a0, a1, b0, b1 = max(y, y0), min(y + P, y1), max(x, x0), min(x + P, x1)
            if a0 >= a1 or b0 >= b1:
                continue
            k = iy * len(xs) + ix
            if cache is not None and k in cache:
                out = cache[k]
            else:
                out = model_fn(data_fn(ys_in[iy], ys_in[iy] + ps, xs_in[ix], xs_in[ix] + ps))
                if cache is not None:
                    cache[k] = out
            if num is None:
                num = torch.zeros(1, out.shape[1], y1 - y0, x1 - x0, dtype=out.dtype)
                den = torch.zeros_like(num)
            wv = wpatch[a0 - y:a1 - y, b0 - x:b1 - x]
            num[0, :, a0 - y0:a1 - y0, b0 - x0:b1 - x0] += out[0, :, a0 - y:a1 - y, b0 - x:b1 - x] * wv
            den[0, :, a0 - y0:a1 - y0, b0 - x0:b1 - x0] += wv[None]
    return num / den


def geometry_table(shapes):
    """Golden G1 helper: {(H,W): dict(n, nh, nw, ys, xs)}."""
    out = {}
    for (h, w) in shapes:
        ps, ys, xs = chop_geometry(h, w)
        out[(h, w)] = dict(ps=ps, n=len(ys) * len(xs), nh=len(ys), nw=len(xs),
                           ys=np.asarray(ys), xs=np.asarray(xs))
    return out
