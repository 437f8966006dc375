"""Mirror of the hot-path functions of the reference's utils/utils.py, backed by
HIP kernels (csrc/tiles.hip) through the C ABI:

    extract_patches_2d / recompose_tensor   utils/utils.py:318-369 / 372-445
    np2tensor / tensor2np                   utils/utils.py:164-194 / 197-248
    mod2normal / swa2normal                 utils/utils.py:666-698 / 701-720 (host, key renaming)

Same names, argument meaning and error behaviour.  Tensors must live on the
GPU; there is no CPU fallback.
"""
import os
import os.path as osp

import numpy as np
import torch

from .. import lib as L

try:                                    # the reference reads / writes through OpenCV (utils.py:3-4,68-95); PIL is the stand-in where it is absent
    import cv2
    cv2_available = True
except ImportError:
    cv2_available = False

IMG_EXTENSIONS = ['.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.webp', 'tga', '.tif', '.tiff', '.dng']       # utils.py:18-19 (sic: 'tga')
MODEL_EXTENSIONS = ['.pth', '.pt']                                                                       # utils.py:16

# what an integer image is divided by (utils.py:22-33; the image reader returns uint8 or uint16)
MAX_VALUES_BY_DTYPE = {np.dtype("uint8"): 255, np.dtype("uint16"): 65535}


def _dt(t):
    if t.dtype == torch.float16:
        return L.F16
    if t.dtype == torch.float32:
        return L.F32
    raise TypeError(f'unsupported dtype {t.dtype}')


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensor must be on the GPU (innfer_amd has no CPU path)')


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _on(t):
    """The library launches on the process's current HIP device: make it the tensor's for the call."""
    return torch.cuda.device(t.device)


def extract_patches_2d(img, patch_shape, step=None, batch_first=False, tile_range=None):
    """[1,C,H,W] -> [n,1,C,ph,pw] ([1,n,C,ph,pw] when batch_first): sliding tiles
    with stride int(patch*step) plus a ragged last row/column (utils.py:318-369).
    tile_range=(begin,count) extracts a contiguous sub-range (multi-GPU sharding)."""
    if step is None:
        step = [1.0, 1.0]
    _need_cuda(img, 'extract_patches_2d')
    B, C, H, W = img.shape
    ph, pw = patch_shape
    if B != 1:
        raise NotImplementedError('extract_patches_2d: batch > 1 (run.py only ever passes batch 1)')
    if ph != pw or step[0] != step[1] or not isinstance(step[0], float):
        raise NotImplementedError('extract_patches_2d: square patches with one fractional step only (chop_forward usage)')
    if H < ph or W < pw:
        # the reference's pad branch dereferences an undefined name `nn` (utils.py:341,346)
        raise NameError("name 'nn' is not defined")
    ps, ys, xs = L.chop_plan(H, W, ph, step[0])
    n = len(ys) * len(xs)
    begin, count = tile_range if tile_range is not None else (0, n)
    img = img.contiguous()
    tiles = torch.empty((count, C, ps, ps), dtype=img.dtype, device=img.device)
    with _on(img):
        L.check(L.lib.innfer_extract_tiles(img.data_ptr(), _dt(img), C, H, W, ph, step[0], begin, count,
                                       tiles.data_ptr(), _stream(img)))
    out = tiles.unsqueeze(1)                       # [n, B=1, C, ph, pw]
    return out.permute(1, 0, 2, 3, 4) if batch_first else out


def recompose_tensor(patches, height, width, step=None, scale=1, out_dtype=None):
    """Weighted overlap blend of [n,C,P,P] tiles into [n/(nh*nw),C,scale*H,scale*W]
    (utils.py:372-445).  fp32 accumulation in the reference's tile order."""
    if step is None:
        step = [1.0, 1.0]
    assert isinstance(step, float) and step >= 0.5 and step <= 1.0
    _need_cuda(patches, 'recompose_tensor')
    patches = patches.contiguous()
    n, C, P, P2 = patches.shape
    assert P == P2
    FH, FW = scale * height, scale * width
    eff = int(P * step)
    nh = 1 + (max(FH, P) - P) // eff + (1 if (max(FH, P) - P) % eff else 0)
    nw = 1 + (max(FW, P) - P) // eff + (1 if (max(FW, P) - P) % eff else 0)
    nb = n // (nh * nw)
    dtype = out_dtype or patches.dtype
    out = torch.empty((nb, C, FH, FW), dtype=dtype, device=patches.device)
    odt = L.F16 if dtype == torch.float16 else L.F32
    with _on(patches):
        L.check(L.lib.innfer_recompose(patches.data_ptr(), _dt(patches), n, C, P, height, width, float(step), scale,
                                       out.data_ptr(), odt, _stream(patches)))
    return out


def np2tensor(img, bgr2rgb=True, data_range=1., normalize=False, change_range=True, add_batch=True,
              device='cuda', dtype=torch.float32):
    """uint8 / uint16 HWC BGR(A) image -> [1,C,H,W] RGB(A) float tensor ON THE GPU
    (utils.py:164-194): the integer image crosses PCIe (4x / 2x fewer bytes than fp32) and
    /maxval (MAX_VALUES_BY_DTYPE, utils.py:22-33), HWC->CHW, channel flip and optional [-1,1] norm run in one HIP kernel.
    bgr2rgb / change_range / add_batch as in the reference (`data_range` is unused there too)."""
    if not isinstance(img, np.ndarray):
        raise TypeError("Got unexpected object type, expected np.ndarray")
    if img.dtype not in (np.uint8, np.uint16) or img.ndim != 3:
        raise NotImplementedError('np2tensor: uint8 / uint16 HWC images are built (the reader, utils.py:36-133, produces nothing else)')
    H, W, Cc = img.shape
    bits = 8 if img.dtype == np.uint8 else 16
    maxval = float(MAX_VALUES_BY_DTYPE[img.dtype]) if change_range else 1.0
    host = np.ascontiguousarray(img)
    if bits == 16:                                   # torch has no uint16 arithmetic; the bytes are what crosses PCIe
        host = host.view(np.int16)
    d_img = torch.from_numpy(host).to(device)
    out = torch.empty((1, Cc, H, W), dtype=dtype, device=d_img.device)
    with _on(out):
        L.check(L.lib.innfer_inthwc_to_nchw(d_img.data_ptr(), bits, H, W, Cc, int(bool(bgr2rgb)), int(bool(normalize)), maxval,
                                            out.data_ptr(), _dt(out), _stream(out)))
    return out if add_batch else out.squeeze(0)


def tensor2np(img, rgb2bgr=True, remove_batch=True, data_range=255, denormalize=False,
              change_range=True, imtype=np.uint8):
    """[1,C,H,W] / [C,H,W] / [H,W] RGB float GPU tensor -> HWC BGR uint8 (or uint16) numpy (utils.py:197-248);
    clip * data_range and round-half-to-even on the GPU, one integer D2H copy.  (data_range, imtype) = (255, np.uint8) or
    (65535, np.uint16); rgb2bgr as in the reference (3- and 4-channel images only)."""
    if not isinstance(img, torch.Tensor):
        raise TypeError("Got unexpected object type, expected torch.Tensor")
    n_dim = img.dim()
    if n_dim not in (2, 3, 4):
        raise TypeError(f'Only support 4D, 3D and 2D tensor. But received with dimension: {n_dim:d}')
    if n_dim == 4 and (img.shape[0] != 1 or not remove_batch):
        raise NotImplementedError('tensor2np: a 4D tensor must be one image with remove_batch=True (the reference transposes a kept batch axis into the image)')
    if not change_range or (data_range, np.dtype(imtype)) not in ((255, np.dtype(np.uint8)), (65535, np.dtype(np.uint16))):
        raise NotImplementedError('tensor2np: built for (data_range, imtype) = (255, uint8) and (65535, uint16) with change_range=True')
    _need_cuda(img, 'tensor2np')
    img = img.contiguous()
    Cc, (H, W) = (1 if n_dim == 2 else img.shape[-3]), img.shape[-2:]
    bits = 8 if np.dtype(imtype) == np.dtype(np.uint8) else 16
    out = torch.empty((H, W, Cc), dtype=torch.uint8 if bits == 8 else torch.int16, device=img.device)
    with _on(img):
        L.check(L.lib.innfer_nchw_to_inthwc(img.data_ptr(), _dt(img), H, W, Cc, int(bool(rgb2bgr) and n_dim != 2), int(bool(denormalize)), bits,
                                            out.data_ptr(), _stream(img)))
    arr = out.cpu().numpy()
    if bits == 16:
        arr = arr.view(np.uint16)
    return arr[:, :, 0] if n_dim == 2 else arr


# ---------------------------------------------------------------- files (utils.py:36-133): the image loop's codec hand-off
def _suffixes(extensions):
    return tuple(extensions) if not isinstance(extensions, str) else (extensions,)


def is_ext_file(filename, extensions=IMG_EXTENSIONS):
    """True when the name ends in one of `extensions` (case-sensitive, as the reference's lists spell both cases: utils.py:36-37)."""
    return os.fspath(filename).endswith(_suffixes(extensions))


def scan_dir(path, extensions=IMG_EXTENSIONS):
    """Every file below `path` whose name ends in one of `extensions`, ordered by (directory, name) -- the order of the reference's sorted walk
    (utils.py:40-49), which is the order the image loop processes and numbers its outputs in."""
    if not osp.isdir(path):
        raise AssertionError(f'{path:s} is not a valid directory')
    suffixes = _suffixes(extensions)
    hits = [(folder, name) for folder, _, names in os.walk(path) for name in names if name.endswith(suffixes)]
    return [osp.join(folder, name) for folder, name in sorted(hits)]


def _non_empty_scan(path, extensions, what):
    found = scan_dir(path, extensions)
    if not found:
        raise AssertionError(f'{path:s} has no valid {what} file')
    return found


def get_models_paths(path):
    return _non_empty_scan(path, MODEL_EXTENSIONS, 'model')


def get_images_paths(path):
    return _non_empty_scan(path, IMG_EXTENSIONS, 'image')


def read_img(path=None):
    """cv2.imread(path, IMREAD_UNCHANGED) (utils.py:68-89): HWC BGR / BGRA (HW for gray), uint8 or uint16, None when the file cannot
    be decoded.  Without OpenCV the file is decoded by PIL and put into OpenCV's channel order."""
    if not path:
        raise AssertionError("Empty path provided.")
    if cv2_available:
        return cv2.imread(path, cv2.IMREAD_UNCHANGED)
    from PIL import Image
    try:
        with Image.open(path) as im:
            if im.mode in ('P', 'CMYK', 'YCbCr', '1'):
                im = im.convert('RGBA' if 'transparency' in im.info else 'RGB')
            elif im.mode == 'LA':
                im = im.convert('RGBA')
            a = np.asarray(im)
    except Exception:
        return None
    if a.dtype == np.int32:                  # PIL's 16-bit gray ('I') arrives as int32
        a = a.astype(np.uint16)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]
    elif a.ndim == 3 and a.shape[2] == 4:
        a = a[:, :, [2, 1, 0, 3]]
    return np.ascontiguousarray(a)


def _resize_nearest(img, h, w):
    """cv2.resize(INTER_NEAREST): source index floor(dst * src / dst_size)."""
    ys = np.minimum((np.arange(h) * (img.shape[0] / h)).astype(np.int64), img.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (img.shape[1] / w)).astype(np.int64), img.shape[1] - 1)
    return img[ys][:, xs]


def save_img(img, img_path, mode='RGB', scale=None):
    """cv2.imwrite of a BGR(A) / gray image (utils.py:92-96), through PIL when OpenCV is absent."""
    if scale:
        img = _resize_nearest(img, int(round(img.shape[0] * scale)), int(round(img.shape[1] * scale)))
    if cv2_available:
        cv2.imwrite(img_path, img)
        return
    from PIL import Image
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]
    elif a.ndim == 3 and a.shape[2] == 4:
        a = a[:, :, [2, 1, 0, 3]]
    elif a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    Image.fromarray(np.ascontiguousarray(a)).save(img_path)


def merge_imgs(img_list):
    """Images side by side, the smaller ones enlarged (nearest) to the largest height / width (utils.py:99-124)."""
    if isinstance(img_list, np.ndarray):
        return img_list
    if not isinstance(img_list, list):
        raise NotImplementedError('To merge images img_list should be a list of cv2 images.')
    img_h = max(im.shape[0] for im in img_list)
    img_v = max(im.shape[1] for im in img_list)
    return np.concatenate([im if im.shape[:2] == (img_h, img_v) else _resize_nearest(im, img_h, img_v) for im in img_list], axis=1)


def save_img_comp(img_list, img_path, mode='RGB'):
    save_img(img=merge_imgs(img_list), img_path=img_path, mode=mode)


def modcrop(img_in, scale):
    """utils.py:250-264."""
    img = np.copy(img_in)
    if img.ndim not in (2, 3):
        raise ValueError('Wrong img ndim: [{:d}].'.format(img.ndim))
    H, W = img.shape[:2]
    return img[:H - H % scale, :W - W % scale]


def linear_resize(img, st=256, device='cuda'):
    """utils.py:267-276: enlarge to the next multiple of `st` in linear light (srgb2linear -> bicubic -> linear2srgb), on the GPU
    (csrc/colorfix.hip; the bicubic follows OpenCV's INTER_CUBIC formulas -- parity with OpenCV itself is unpinned, as for color_fix)."""
    h, w = img.shape[0:2]
    if h % st == 0 and w % st == 0:
        return img
    if img.dtype != np.uint8 or img.ndim != 3:
        raise TypeError('linear_resize: expected a uint8 HWC numpy image')
    oh, ow = -(-h // st) * st, -(-w // st) * st
    Cc = img.shape[2]
    d_in = torch.from_numpy(np.ascontiguousarray(img)).to(device)
    out = torch.empty((oh, ow, Cc), dtype=torch.uint8, device=d_in.device)
    ws = torch.empty(h * w * Cc * 4, dtype=torch.uint8, device=d_in.device)
    with _on(out):
        L.check(L.lib.innfer_linear_resize(d_in.data_ptr(), h, w, Cc, out.data_ptr(), oh, ow, ws.data_ptr(), ws.numel(), _stream(out)))
    return out.cpu().numpy()


def color_fix(imgA, imgB, device='cuda'):
    """`-cf` (utils.py:278-315): add the low-frequency LR - SR difference (linear light) back to the SR
    image.  imgA (LR) and imgB (SR) are uint8 HWC numpy images like the reference's; both cross PCIe as
    uint8, everything else runs in libinnfer_amd.so (csrc/colorfix.hip); returns a uint8 HWC numpy image.
    The reference's two OpenCV calls (bicubic resize, 3x3 Gaussian) follow OpenCV's published float32
    algorithms -- OpenCV itself is not available here to compare with."""
    for im in (imgA, imgB):
        if not isinstance(im, np.ndarray) or im.dtype != np.uint8 or im.ndim != 3:
            raise TypeError('color_fix: expected uint8 HWC numpy images')
    if imgA.shape[2] != imgB.shape[2]:
        raise ValueError('color_fix: channel counts differ')
    hA, wA, Cc = imgA.shape
    hB, wB, _ = imgB.shape
    d_a = torch.from_numpy(np.ascontiguousarray(imgA)).to(device)
    d_b = torch.from_numpy(np.ascontiguousarray(imgB)).to(device)
    out = torch.empty((hB, wB, Cc), dtype=torch.uint8, device=d_a.device)
    ws = torch.empty(L.lib.innfer_color_fix_workspace_bytes(hA, wA, hB, wB, Cc), dtype=torch.uint8, device=d_a.device)
    with _on(out):
        L.check(L.lib.innfer_color_fix(d_a.data_ptr(), hA, wA, d_b.data_ptr(), hB, wB, Cc, out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _stream(out)))
    return out.cpu().numpy()


def guided_filter(x, y, x_HR=None, ks=None, r=None, eps=1e-2, box_kernel=None, mode='regular', conv_a=None):
    """guided_filter (utils.py:548-626): 'regular' mode (what run.py:427-429 applies after the WBC UNet with r=1) and 'fast' mode (A, b of the
    low-resolution pair enlarged bilinearly to the high-resolution guidance x_HR), any odd window ks = 2 r + 1 (box means, reflect padding);
    x guidance, y input, [B,C,H,W] GPU tensors of one dtype.  Runs in libinnfer_amd.so (csrc/wbcunet.hip: one fused pass per stage).  The remaining
    forms -- 'conv' mode (the caller's nn.Sequential `conv_a` computes A from cat(cov_xy, var_x)), a precomputed box_kernel tensor, even window
    sizes -- follow the reference's formula step by step on the HIP filter2D (its box means) with the caller's module in between."""
    if mode not in ('regular', 'fast', 'conv'):
        raise NotImplementedError("guided_filter: modes 'regular', 'fast' and 'conv'")
    if not isinstance(box_kernel, torch.Tensor):
        if not ks:
            if not r:
                raise ValueError("Either kernel size (ks) or radius (r) for the window are required.")
            ks = 2 * r + 1
    if mode == 'conv' or isinstance(box_kernel, torch.Tensor) or int(ks) != ks or int(ks) % 2 == 0:
        return _guided_filter_stepwise(x, y, x_HR, box_kernel if isinstance(box_kernel, torch.Tensor) else get_box_kernel(kernel_size=ks), eps, mode, conv_a)
    if mode == 'fast' and not isinstance(x_HR, torch.Tensor):
        raise ValueError("guided_filter: mode 'fast' needs the high-resolution guidance x_HR")
    _need_cuda(x, 'guided_filter')
    if x.shape != y.shape or x.dtype != y.dtype or x.dim() != 4:
        raise ValueError('guided_filter: x and y must be [B,C,H,W] tensors of one shape and dtype')
    x, y = x.contiguous(), y.contiguous()
    B, Cc, H, W = x.shape
    hr = None
    if mode == 'fast':
        hr = x_HR.contiguous()
        if hr.dim() != 4 or hr.shape[:2] != x.shape[:2] or hr.dtype != x.dtype or hr.device != x.device:
            raise ValueError('guided_filter: x_HR must be a [B,C,Hh,Wh] tensor with the batch, channels, dtype and device of x')
    out = torch.empty_like(hr if hr is not None else x)
    ws = torch.empty(L.lib.innfer_guided_filter_workspace_bytes(B, Cc, H, W), dtype=torch.uint8, device=x.device)
    with _on(x):
        L.check(L.lib.innfer_guided_filter_ex(x.data_ptr(), y.data_ptr(), _dt(x), B, Cc, H, W, int(ks), float(eps),
                                              hr.data_ptr() if hr is not None else None, hr.shape[2] if hr is not None else 0,
                                              hr.shape[3] if hr is not None else 0, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x)))
    return out


def _guided_filter_stepwise(x, y, x_HR, box_kernel, eps, mode, conv_a):
    """utils.py:590-626 step by step: box means by the HIP filter2D, the products / quotients as tensor expressions, `conv_a` the caller's module."""
    import torch.nn.functional as F
    _need_cuda(x, 'guided_filter')
    if mode in ('fast', 'conv') and not isinstance(x_HR, torch.Tensor):
        raise ValueError(f"guided_filter: mode '{mode}' needs the high-resolution guidance x_HR")
    if mode == 'conv' and conv_a is None:
        raise ValueError("guided_filter: mode 'conv' needs conv_a")
    box_kernel = box_kernel.to(x.device)
    N = filter2D(torch.ones((1, 1, x.shape[-2], x.shape[-1]), device=x.device, dtype=x.dtype), box_kernel)
    mean_x = filter2D(x, box_kernel) / N
    mean_y = filter2D(y, box_kernel) / N
    cov_xy = (filter2D(x * y, box_kernel) / N) - mean_x * mean_y
    var_x = (filter2D(x * x, box_kernel) / N) - mean_x * mean_x
    A = conv_a(torch.cat([cov_xy, var_x], dim=1)) if mode == 'conv' else cov_xy / (var_x + eps)
    b = mean_y - A * mean_x
    if mode in ('fast', 'conv'):
        size = (x_HR.shape[-2], x_HR.shape[-1])
        return F.interpolate(A, size, mode='bilinear', align_corners=True) * x_HR + F.interpolate(b, size, mode='bilinear', align_corners=True)
    return (filter2D(A, box_kernel) / N) * x + filter2D(b, box_kernel) / N


# --------------------------------------------------------------- small helpers of the reference's utils (host math / elementwise plumbing)
def denorm(x, min_max=(-1.0, 1.0)):
    """[min, max] -> [0, 1], clamped (utils.py:136-150).  The image path applies it inside tensor2np / the uint8 epilogue; this is the free function."""
    out = (x - min_max[0]) / (min_max[1] - min_max[0])
    if isinstance(x, torch.Tensor):
        return out.clamp(0, 1)
    if isinstance(x, np.ndarray):
        return np.clip(out, 0, 1)
    raise TypeError("Got unexpected object type, expected torch.Tensor or np.ndarray")


def norm(x):
    """[0, 1] -> [-1, 1], clamped (utils.py:152-161)."""
    out = (x - 0.5) * 2.0
    if isinstance(x, torch.Tensor):
        return out.clamp(-1, 1)
    if isinstance(x, np.ndarray):
        return np.clip(out, -1, 1)
    raise TypeError("Got unexpected object type, expected torch.Tensor or np.ndarray")


def filter2D(x, kernel, border_type='reflect', dim=2, normalized=False):
    """Every channel of x [B,C,H,W] convolved (cross-correlated) with one [1,kH,kW] kernel behind F.pad(x, compute_padding((kH, kW)), border_type):
    the output keeps x's shape and dtype (utils.py:484-535).  HIP kernel (csrc/wbcunet.hip k_filter2d); dim 2 only."""
    borders = ['constant', 'reflect', 'replicate', 'circular']
    if border_type not in borders:
        raise ValueError("Invalid border_type, we expect the following: {0}.Got: {1}".format(borders, border_type))
    if dim != 2:
        raise NotImplementedError("filter2D: 2-D tensors are built (the reference's only use)")
    if not isinstance(x, torch.Tensor) or x.dim() != 4:
        raise ValueError('expected a 4D [B,C,H,W] tensor')
    if not x.is_cuda:
        raise RuntimeError('innfer_amd runs filter2D on an MI355X only: there is no CPU path')
    k = kernel.unsqueeze(0).to(x.device).to(x.dtype)
    if normalized:
        k = normalize_kernel2d(k)
    kH, kW = int(k.shape[-2]), int(k.shape[-1])
    pad = compute_padding((kH, kW))                                   # (left, right, top, bottom)
    if pad[0] + pad[1] != kW - 1 or pad[2] + pad[3] != kH - 1:
        raise NotImplementedError("filter2D: this kernel shape changes the output size in the reference (mixed even / odd sides)")
    x = x.contiguous() if x.dtype in (torch.float16, torch.float32) else x.float().contiguous()
    kd = k.reshape(kH, kW).float().contiguous()
    out = torch.empty_like(x)
    B, Cc, H, W = x.shape
    with _on(x):
        L.check(L.lib.innfer_filter2d(x.data_ptr(), _dt(x), B * Cc, H, W, kd.data_ptr(), kH, kW, int(pad[0]), int(pad[2]), borders.index(border_type),
                                      out.data_ptr(), _stream(x)))
    return out


def get_box_kernel(kernel_size=5, dim=2):
    """The mean filter of guided_filter as a tensor (utils.py:538-546): ones / (kx * ky)."""
    if isinstance(kernel_size, (int, float)):
        kernel_size = [kernel_size] * dim
    kx, ky = int(kernel_size[0]), int(kernel_size[1])
    return torch.full((kx, ky), 1.0 / (float(kx) * float(ky)), dtype=torch.float32)


def normalize_kernel2d(x):
    """L1-normalise the last two axes of a kernel (utils.py:448-454)."""
    if x.dim() < 2:
        raise TypeError("input should be at least 2D tensor. Got {}".format(x.size()))
    return x / x.abs().sum(dim=(-1, -2), keepdim=True)


def compute_padding(kernel_size):
    """Padding that keeps the size under a kernel (utils.py:457-481): k // 2 for an int; for a tuple / list one (before, after) pair per axis,
    last axis first, the 'before' side one less for even kernels."""
    if isinstance(kernel_size, int):
        return kernel_size // 2
    ks = list(kernel_size)
    half = [k // 2 for k in ks]
    out = []
    for i in range(len(ks)):
        after = half[-(i + 1)]
        out += [after - 1 if ks[i] % 2 == 0 else after, after]
    return out


# --------------------------------------------------------------- key converters
_NEW2OLD_FIXED = (('conv_first', 'model.0'), ('trunk_conv', 'model.1.sub.23'), ('upconv1', 'model.3'),
                  ('upconv2', 'model.6'), ('HRconv', 'model.8'), ('conv_last', 'model.10'))


def mod2normal(state_dict):
    """New-arch ESRGAN keys (conv_first / RRDB_trunk.N.RDBk.convj / trunk_conv /
    upconv1,2 / HRconv / conv_last) -> old-arch keys (utils.py:666-698).  Like the
    reference this assumes the 23-block 4x layout ('model.1.sub.23', 'model.3' ...)."""
    if 'conv_first.weight' not in state_dict:
        return state_dict
    print('Converting and loading a modified RRDB model to normal RRDB')
    out = {}
    for new, old in _NEW2OLD_FIXED[:1]:
        for p in ('weight', 'bias'):
            out[f'{old}.{p}'] = state_dict[f'{new}.{p}']
    for k, v in state_dict.items():
        if 'RDB' in k:
            k2 = k.replace('RRDB_trunk.', 'model.1.sub.')
            if '.weight' in k:
                k2 = k2.replace('.weight', '.0.weight')
            elif '.bias' in k:
                k2 = k2.replace('.bias', '.0.bias')
            out[k2] = v
    for new, old in _NEW2OLD_FIXED[1:]:
        for p in ('weight', 'bias'):
            out[f'{old}.{p}'] = state_dict[f'{new}.{p}']
    return out


def normal2mod(state_dict):
    """Old-arch ESRGAN keys -> new-arch keys (utils.py:629-663), the inverse of mod2normal; like the reference it assumes the 23-block 4x layout."""
    if 'model.0.weight' not in state_dict:
        return state_dict
    print('Converting and loading an RRDB model to modified RRDB')
    out = {}
    for new, old in _NEW2OLD_FIXED[:1]:
        for p in ('weight', 'bias'):
            out[f'{new}.{p}'] = state_dict[f'{old}.{p}']
    for k, v in state_dict.items():
        if 'RDB' in k:
            k2 = k.replace('model.1.sub.', 'RRDB_trunk.')
            if '.0.weight' in k:
                k2 = k2.replace('.0.weight', '.weight')
            elif '.0.bias' in k:
                k2 = k2.replace('.0.bias', '.bias')
            out[k2] = v
    for new, old in _NEW2OLD_FIXED[1:]:
        for p in ('weight', 'bias'):
            out[f'{new}.{p}'] = state_dict[f'{old}.{p}']
    return out


def swa2normal(state_dict):
    """Unwrap a torch.optim.swa_utils.AveragedModel checkpoint: keep only
    'module.module.*' entries, stripped of that prefix (utils.py:701-720)."""
    if 'n_averaged' not in state_dict:
        return state_dict
    print('Attempting to convert a SWA model to a regular model\n')
    out = {}
    for k, v in state_dict.items():
        if 'n_averaged' in k:
            print('n_averaged: {}'.format(v))
        elif 'module.module.' in k:
            out[k.replace('module.module.', '')] = v
    return out
