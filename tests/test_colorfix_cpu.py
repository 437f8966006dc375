"""Oracle checks for the `-cf` colour fix restatement (oracle/colorfix.py).  The two OpenCV calls of the
reference (cv2.resize INTER_CUBIC, cv2.GaussianBlur 3x3) cannot be pinned against OpenCV itself -- it is
absent from the image -- so these are known-answer / property tests of the published algorithms."""
import numpy as np

from innfer_amd import synth
from oracle.colorfix import color_fix, gauss3, resize_cubic


def test_cubic_resize_known_answers():
    x = np.random.RandomState(0).rand(6, 5, 3).astype(np.float32)
    assert np.array_equal(resize_cubic(x, (5, 6)), x)                       # same size: t = 0 -> weights (0, 1, 0, 0)
    c = np.full((7, 9, 3), 0.3, np.float32)
    assert np.abs(resize_cubic(c, (18, 14)) - 0.3).max() < 1e-6             # weights sum to one
    # impulse response of a 2x upscale = the Keys kernel (A = -0.75) sampled at t = 0.25 / 0.75
    imp = np.zeros((1, 8, 1), np.float32); imp[0, 3, 0] = 1.0
    up = resize_cubic(imp, (16, 1))[0, :, 0]
    A = -0.75
    def w_near(t): return ((A + 2) * t - (A + 3)) * t * t + 1               # |x| <= 1
    def w_far(t): return ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A   # 1 < |x| < 2, x = t + 1
    want = np.zeros(16)
    want[6], want[7] = w_near(0.25), w_near(0.25)                            # dst 6 -> src 2.75, dst 7 -> src 3.25
    want[5], want[8] = w_near(0.75), w_near(0.75)
    want[4], want[9] = w_far(0.25), w_far(0.25)
    want[3], want[10] = w_far(0.75), w_far(0.75)
    assert np.abs(up - want).max() < 1e-6, (up, want)
    # shrinking samples 4 taps around the source position (no antialiasing): 4x down of a constant-per-4 pattern
    blocks = np.repeat(np.arange(6, dtype=np.float32), 4)[None, :, None]
    assert resize_cubic(blocks, (6, 1)).shape == (1, 6, 1)


def test_gauss3_is_the_fixed_small_kernel_with_reflect101():
    imp = np.zeros((5, 5, 1), np.float32); imp[2, 2, 0] = 1.0
    k = np.array([0.25, 0.5, 0.25], np.float32)
    assert np.allclose(gauss3(imp)[1:4, 1:4, 0], np.outer(k, k))
    edge = np.zeros((4, 4, 1), np.float32); edge[0, 0, 0] = 1.0             # reflect101: x[-1] = x[1], the corner keeps 0.5 * 0.5
    g = gauss3(edge)[:, :, 0]
    assert np.isclose(g[0, 0], 0.25) and np.isclose(g[0, 1], 0.125) and np.isclose(g[1, 1], 0.0625)
    assert np.isclose(gauss3(np.full((3, 3, 2), 0.7, np.float32)).mean(), 0.7)


def test_color_fix_properties():
    yy, xx = np.mgrid[0:20, 0:28]
    a = np.stack([60 + 5 * xx, 90 + 4 * yy, 200 - 3 * xx - 2 * yy], -1).astype(np.uint8)       # smooth LR image
    # identical images: the difference is zero, the result is the sRGB round trip of the image itself
    out = color_fix(a, a)
    assert out.shape == a.shape and out.dtype == np.uint8
    assert np.abs(out.astype(int) - a.astype(int)).max() <= 1               # linear2srgb truncates
    # a colour cast on the SR image is removed: the low-frequency LR - SR difference is added back
    from oracle.colorfix import resize_cubic as rc
    b = np.clip(rc(a.astype(np.float32), (112, 80)), 0, 255).astype(np.uint8)                  # a plausible SR image
    cast = np.clip(b.astype(int) + np.array([12, 0, -9]), 0, 255).astype(np.uint8)
    fixed = color_fix(a, cast)
    err_fixed = np.abs(fixed.astype(float) - b.astype(float))[8:-8, 8:-8].mean((0, 1))
    err_cast = np.abs(cast.astype(float) - b.astype(float))[8:-8, 8:-8].mean((0, 1))
    assert err_fixed.max() < 1.5 and err_cast.max() > 8, (err_fixed, err_cast)


def test_cubic_resize_agrees_with_aten_bicubic():
    """A second, independent implementation of the same published algorithm: ATen's upsample_bicubic2d (align_corners=False, no antialias) is the
    Keys kernel with A = -0.75 on the half-pixel grid with clamped taps -- the scheme OpenCV's INTER_CUBIC float path documents.  This does not pin
    the oracle against OpenCV itself (absent here); it rules out a slip in the restatement (weights, tap order, source coordinate, border)."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(3)
    for (h, w, hd, wd) in [(9, 13, 36, 52), (9, 13, 18, 39), (40, 36, 10, 9), (17, 5, 17, 20), (8, 8, 3, 29)]:
        x = rs.rand(h, w, 3).astype(np.float32)
        ours = resize_cubic(x, (wd, hd))
        ref = F.interpolate(torch.from_numpy(x).permute(2, 0, 1)[None], size=(hd, wd), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert ours.shape == ref.shape == (hd, wd, 3)
        assert np.abs(ours - ref).max() < 5e-6, (h, w, hd, wd, np.abs(ours - ref).max())


def test_gauss3_agrees_with_a_reflect_padded_convolution():
    """cv2.GaussianBlur(x, (3, 3), 0) = the fixed [1, 2, 1] / 4 kernel under BORDER_REFLECT_101, checked against torch's 'reflect' padding (the same
    border rule: the edge pixel is not repeated) and a plain conv2d."""
    import torch
    import torch.nn.functional as F
    x = np.random.RandomState(4).rand(11, 7, 3).astype(np.float32)
    k = torch.tensor([0.25, 0.5, 0.25])
    k2 = (k[:, None] * k[None, :])[None, None].repeat(3, 1, 1, 1)
    t = F.pad(torch.from_numpy(x).permute(2, 0, 1)[None], (1, 1, 1, 1), mode="reflect")
    ref = F.conv2d(t, k2, groups=3)[0].permute(1, 2, 0).numpy()
    assert np.abs(gauss3(x) - ref).max() < 1e-6
