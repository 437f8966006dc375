"""Mirror of the reference's utils/colors.py sRGB helpers (colors.py:29-60), on the GPU.
Same names and argument meaning: numpy in, numpy out (the reference works on host arrays here)."""
import numpy as np
import torch

from .. import lib as L


def srgb2linear(srgb, gamma=2.4, th=0.04045, device='cuda'):
    """uint8 sRGB image (any shape) -> float32 linear RGB in [0,1]."""
    if gamma != 2.4 or th != 0.04045:
        raise NotImplementedError('srgb2linear: only the default gamma/threshold are built')
    a = np.ascontiguousarray(srgb)
    if a.dtype != np.uint8:
        raise NotImplementedError('srgb2linear: uint8 input only')
    d_in = torch.from_numpy(a).to(device)
    d_out = torch.empty(a.shape, dtype=torch.float32, device=d_in.device)
    L.check(L.lib.innfer_srgb_to_linear(d_in.data_ptr(), d_out.data_ptr(), a.size,
                                        torch.cuda.current_stream(d_in.device).cuda_stream))
    return d_out.cpu().numpy()


def linear2srgb(linear, gamma=2.4, th=0.0031308, device='cuda'):
    """float32 linear RGB -> uint8 sRGB (clip, gamma, *255, truncating cast)."""
    if gamma != 2.4 or th != 0.0031308:
        raise NotImplementedError('linear2srgb: only the default gamma/threshold are built')
    a = np.ascontiguousarray(linear, dtype=np.float32)
    d_in = torch.from_numpy(a).to(device)
    d_out = torch.empty(a.shape, dtype=torch.uint8, device=d_in.device)
    L.check(L.lib.innfer_linear_to_srgb(d_in.data_ptr(), d_out.data_ptr(), a.size,
                                        torch.cuda.current_stream(d_in.device).cuda_stream))
    return d_out.cpu().numpy()


# Channel-order helpers of the reference (utils/colors.py:5-26): pure index plumbing on tensors of any device -- the image path itself flips
# channels inside its fused pre / post kernels (np2tensor / tensor2np / forward_u8).
def bgr_to_rgb(image):
    """[.., C, H, W] -> the same with the channel axis reversed (utils/colors.py:5-11)."""
    return image.flip(-3)


def rgb_to_bgr(image):
    """The same flip as bgr_to_rgb (utils/colors.py:14-16)."""
    return bgr_to_rgb(image)


def bgra_to_rgba(image):
    """[4, H, W]: swap channels 0 and 2, keep alpha (utils/colors.py:19-21: it indexes the FIRST axis)."""
    return image[[2, 1, 0, 3], :, :]


def rgba_to_bgra(image):
    """The same permutation as bgra_to_rgba (utils/colors.py:24-26)."""
    return bgra_to_rgba(image)
