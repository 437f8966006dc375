"""Oracle for the guided filter the reference applies after the WBC UNet (utils/utils.py:548-626 with
filter2D :484-533 and get_box_kernel :536-545; run.py:427-429 calls it with r=1, eps=5e-3): 'regular' mode,
3x3 box means with reflect padding.  Test infrastructure."""
import torch
import torch.nn.functional as F


def box3(t):
    """filter2D(t, ones(3,3)/9): depthwise conv over a reflect-padded tensor (utils.py:514-533)."""
    c = t.shape[1]
    k = torch.full((c, 1, 3, 3), 1.0 / 9.0, dtype=t.dtype)
    return F.conv2d(F.pad(t, (1, 1, 1, 1), mode="reflect"), k, groups=c)


def guided_filter(x, y, eps=5e-3):
    """guided_filter(x, y, r=1, eps) (utils.py:584-624): x guidance, y filtering input, [B,C,H,W]."""
    n = box3(torch.ones((1, 1, x.shape[-2], x.shape[-1]), dtype=x.dtype))
    mean_x, mean_y = box3(x) / n, box3(y) / n
    cov_xy = box3(x * y) / n - mean_x * mean_y
    var_x = box3(x * x) / n - mean_x * mean_x
    a = cov_xy / (var_x + eps)
    b = mean_y - a * mean_x
    return (box3(a) / n) * x + box3(b) / n
