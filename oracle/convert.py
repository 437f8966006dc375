"""Oracle for image<->tensor conversion and colour helpers (test infrastructure).

numpy / torch restatement of reference utils/utils.py:136-248 and
utils/colors.py:5-60.
"""
import numpy as np
import torch

_MAXVAL = {np.dtype("uint8"): 255, np.dtype("uint16"): 65535,
           np.dtype("float32"): 1.0, np.dtype("float64"): 1.0}   # utils.py:22-33


def np2tensor(img, normalize=False, bgr2rgb=True, change_range=True, add_batch=True):
    """utils.py:164-194: HWC (BGR / BGRA / gray) image -> [1,C,H,W] float32,
    value/maxval, channel order flipped to RGB, optional [-1,1] normalisation."""
    assert isinstance(img, np.ndarray)
    x = img.astype(np.float32) / _MAXVAL.get(img.dtype, 1.0) if change_range else img
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).float()
    c = t.shape[0]
    if bgr2rgb and c % 3 == 0:
        t = t.flip(-3)                                   # colors.py:5-11
    elif bgr2rgb and c == 4:
        t = t[[2, 1, 0, 3]]                              # colors.py:19-21
    if add_batch:
        t = t.unsqueeze(0)
    if normalize:
        t = ((t - 0.5) * 2.0).clamp(-1, 1)               # utils.py:152-161
    return t


def tensor2np(t, denormalize=False, rgb2bgr=True, data_range=255, imtype=np.uint8):
    """utils.py:197-248: [1,C,H,W] / [C,H,W] / [H,W] RGB float -> HWC BGR uint8 (or imtype);
    clip(data_range*x, 0, data_range).round() with numpy round-half-to-even."""
    x = t.float().cpu()
    if x.dim() == 2:
        a = x.numpy()
    else:
        if x.dim() == 4:
            x = x.squeeze(0)
        if x.shape[0] == 3 and rgb2bgr:
            x = x.flip(-3)
        elif x.shape[0] == 4 and rgb2bgr:
            x = x[[2, 1, 0, 3]]
        a = x.numpy().transpose(1, 2, 0)
    if denormalize:
        a = np.clip((a - (-1.0)) / (1.0 - (-1.0)), 0, 1)  # utils.py:136-150
    a = np.clip(data_range * a, 0, data_range).round()
    return a.astype(imtype)


def srgb2linear(srgb, gamma=2.4, th=0.04045):
    """colors.py:29-46 (input in [0,255])."""
    lin = np.float32(srgb) / 255.0
    return np.where(lin <= th, lin / 12.92, np.power((lin + 0.055) / 1.055, gamma))


def linear2srgb(linear, gamma=2.4, th=0.0031308):
    """colors.py:49-60: clip, piecewise gamma, *255, TRUNCATING uint8 cast."""
    s = np.clip(linear.copy(), 0.0, 1.0)
    s = np.where(s <= th, s * 12.92, 1.055 * np.power(s, 1.0 / gamma) - 0.055)
    return np.clip(s * 255.0, 0.0, 255).astype(np.uint8)
