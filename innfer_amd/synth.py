"""Deterministic synthetic data: integer hash -> float.

There are no model-database checkpoints or images without a network, so every
test, golden fixture and benchmark uses weights/images produced here.  The
generator is a pure integer hash (splitmix64) so that it reproduces across
torch/numpy versions ("do not depend on torch.manual_seed").

Scale of the weights follows PyTorch's default Conv2d init (uniform in
+-1/sqrt(fan_in)), which keeps activations O(1) through 23 RRDBs.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u64(n, seed):
    """n 64-bit hashes of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.zeros(1, dtype=np.uint64))
        return _splitmix64(idx ^ base)


def uniform(shape, seed, lo=0.0, hi=1.0):
    """float32 array, uniform in [lo, hi), 24 random mantissa bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(n, seed)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def image_u8(h, w, c=3, seed=0):
    """uint8 HWC image (BGR order by convention of the reference's cv2 reader)."""
    n = h * w * c
    return (hash_u64(n, 0xC0FFEE + seed) >> np.uint64(56)).astype(np.uint8).reshape(h, w, c)


def key_seed(key, seed=0):
    return (zlib.crc32(key.encode()) << 8) ^ seed


def fill_state_dict(shapes, seed=0, bias_scale=1.0):
    """shapes: {key: shape}.  Conv weights (OIHW / IOHW) get +-1/sqrt(fan_in);
    biases the same bound; BatchNorm weight in [0.5,1.5], bias in +-0.1."""
    out = {}
    for k, shp in shapes.items():
        shp = tuple(int(s) for s in shp)
        if len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            b = 1.0 / np.sqrt(fan_in)
            out[k] = uniform(shp, key_seed(k, seed), -b, b)
        elif len(shp) == 1:
            wk = k.rsplit(".", 1)[0] + ".weight"
            if wk in shapes and len(shapes[wk]) == 4 and k.endswith(".bias"):
                s4 = shapes[wk]
                b = bias_scale / np.sqrt(s4[1] * s4[2] * s4[3])
                out[k] = uniform(shp, key_seed(k, seed), -b, b)
            elif k.endswith(".weight"):
                out[k] = uniform(shp, key_seed(k, seed), 0.5, 1.5)
            elif k.endswith("running_var"):
                out[k] = np.ones(shp, np.float32)
            elif k.endswith("running_mean"):
                out[k] = np.zeros(shp, np.float32)
            else:
                out[k] = uniform(shp, key_seed(k, seed), -0.1, 0.1)
        elif len(shp) == 0:
            out[k] = np.zeros((), np.int64)
        else:
            out[k] = uniform(shp, key_seed(k, seed), -0.1, 0.1)
    return out


def fill_running_stats(sd, seed=0):
    """Non-trivial BatchNorm running statistics for a state dict made by fill_state_dict (which leaves a fresh BatchNorm's 0 / 1):
    running_mean uniform in +-0.25, running_var in [0.5, 1.5).  Returns a new dict."""
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith("running_mean"):
            out[k] = uniform(v.shape, key_seed(k, seed), -0.25, 0.25)
        elif k.endswith("running_var"):
            out[k] = uniform(v.shape, key_seed(k, seed), 0.5, 1.5)
    return out


from .architectures.keys import mrrdbnet_shapes, rrdbnet_shapes, srresnet_shapes  # noqa: E402,F401
