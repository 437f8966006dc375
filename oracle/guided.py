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


def box(t, ks):
    """filter2D(t, ones(ks,ks)/ks^2) for odd ks (utils.py:484-545): depthwise conv over a reflect-padded tensor."""
    c, r = t.shape[1], ks // 2
    k = torch.full((c, 1, ks, ks), 1.0, dtype=torch.float64).div(float(ks) * float(ks)).to(t.dtype)
    return F.conv2d(F.pad(t, (r, r, r, r), mode="reflect"), k, groups=c)


def guided_filter_ex(x, y, ks=3, eps=1e-2, x_hr=None):
    """guided_filter(x, y, x_HR, ks=ks, eps=eps, mode='regular' | 'fast') (utils.py:548-626)."""
    n = box(torch.ones((1, 1, x.shape[-2], x.shape[-1]), dtype=x.dtype), ks)
    mean_x, mean_y = box(x, ks) / n, box(y, ks) / n
    cov_xy = box(x * y, ks) / n - mean_x * mean_y
    var_x = box(x * x, ks) / n - mean_x * mean_x
    a = cov_xy / (var_x + eps)
    b = mean_y - a * mean_x
    if x_hr is not None:                                   # 'fast' (:611-619)
        size = (x_hr.shape[-2], x_hr.shape[-1])
        return F.interpolate(a, size, mode="bilinear", align_corners=True) * x_hr + F.interpolate(b, size, mode="bilinear", align_corners=True)
    return (box(a, ks) / n) * x + box(b, ks) / n
