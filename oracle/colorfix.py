"""Oracle for `color_fix` (-cf): reference utils/utils.py:278-315 (test infrastructure).

PARITY UNPINNED for the two OpenCV calls: the reference calls cv2.resize(..., INTER_CUBIC) and
cv2.GaussianBlur(diff, (3, 3), 0); OpenCV is a third-party dependency that is not vendored under
/root/reference, is not installed in this image and is not version-pinned by the reference (no
requirements file).  Both are restated here from OpenCV's published algorithm (imgproc resize.cpp /
smooth: float32 path):
  * resize INTER_CUBIC: src = (dst + 0.5) * (src_size / dst_size) - 0.5; s = floor(src); t = src - s;
    Keys kernel with A = -0.75: w0 = ((A*(t+1) - 5A)*(t+1) + 8A)*(t+1) - 4A, w1 = ((A+2)*t - (A+3))*t*t + 1,
    w2 = ((A+2)*(1-t) - (A+3))*(1-t)*(1-t) + 1, w3 = 1 - w0 - w1 - w2; taps s-1..s+2 with the index clamped to
    the image (replicate); horizontal pass, then vertical pass, float32, no antialiasing when shrinking;
  * GaussianBlur ksize 3, sigma 0: the fixed kernel [0.25, 0.5, 0.25], separable (rows, then columns),
    BORDER_REFLECT_101 (x[-1] = x[1]).
Everything else (srgb2linear / linear2srgb, the difference, the sums) follows the reference line by line.
Cross-checked (tests/test_colorfix_cpu.py) against a second implementation of the same published scheme: ATen's upsample_bicubic2d
(A = -0.75, half-pixel grid, clamped taps) to 5e-6 on up- and down-scales, and a reflect-padded conv2d for the blur -- still not OpenCV itself.
"""
import numpy as np

from .convert import srgb2linear, linear2srgb

_A = np.float32(-0.75)


def _cubic_taps(dst_size, src_size):
    scale = np.float32(src_size) / np.float32(dst_size)
    f = (np.arange(dst_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    s = np.floor(f).astype(np.int64)
    t = (f - s.astype(np.float32)).astype(np.float32)
    one = np.float32(1.0)
    w0 = ((_A * (t + one) - np.float32(5) * _A) * (t + one) + np.float32(8) * _A) * (t + one) - np.float32(4) * _A
    w1 = ((_A + np.float32(2)) * t - (_A + np.float32(3))) * t * t + one
    w2 = ((_A + np.float32(2)) * (one - t) - (_A + np.float32(3))) * (one - t) * (one - t) + one
    w3 = one - w0 - w1 - w2
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src_size - 1)
    return idx, np.stack([w0, w1, w2, w3], 1).astype(np.float32)


def resize_cubic(img, dsize):
    """cv2.resize(img, dsize=(w, h), interpolation=cv2.INTER_CUBIC) for float32 HWC images."""
    wd, hd = dsize
    img = np.asarray(img, dtype=np.float32)
    hs, ws = img.shape[:2]
    ix, wx = _cubic_taps(wd, ws)
    iy, wy = _cubic_taps(hd, hs)
    rows = np.zeros((hs, wd) + img.shape[2:], np.float32)
    for k in range(4):
        rows = rows + img[:, ix[:, k]] * wx[:, k].reshape((1, wd) + (1,) * (img.ndim - 2))
    out = np.zeros((hd, wd) + img.shape[2:], np.float32)
    for k in range(4):
        out = out + rows[iy[:, k]] * wy[:, k].reshape((hd, 1) + (1,) * (img.ndim - 2))
    return out.astype(np.float32)


def gauss3(img):
    """cv2.GaussianBlur(img, (3, 3), 0) for float32 HWC images (BORDER_REFLECT_101)."""
    img = np.asarray(img, dtype=np.float32)
    def pass1d(a, axis):
        p = np.take(a, np.r_[1, np.arange(a.shape[axis]), a.shape[axis] - 2], axis=axis) if a.shape[axis] > 1 else np.concatenate([a, a, a], axis)
        lo = np.take(p, np.arange(0, a.shape[axis]), axis=axis)
        mid = np.take(p, np.arange(1, a.shape[axis] + 1), axis=axis)
        hi = np.take(p, np.arange(2, a.shape[axis] + 2), axis=axis)
        return (mid * np.float32(0.5) + (lo + hi) * np.float32(0.25)).astype(np.float32)
    return pass1d(pass1d(img, 1), 0)


def color_fix(imgA, imgB):
    """utils.py:278-315: add the low-frequency difference (LR - SR) back to the SR image, in linear light."""
    a = srgb2linear(imgA).astype(np.float32)
    b = srgb2linear(imgB).astype(np.float32)
    hA, wA = a.shape[:2]
    hB, wB = b.shape[:2]
    scaling = hA < hB and wA < wB
    b_ds = resize_cubic(b, (wA, hA)) if scaling else b
    blurred = gauss3(a - b_ds)
    if scaling:
        blurred = resize_cubic(blurred, (wB, hB))
    return linear2srgb(blurred + b)
