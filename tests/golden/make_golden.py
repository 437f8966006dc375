#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE (victorca25/iNNfer) itself.

Runs only in the build container, where /root/reference is mounted; the GPU box
never sees the reference.  `cv2` is absent, so an empty stub module is injected
before `utils.utils` is imported -- no cv2-dependent function is
called.  Weights/images come from innfer_amd.synth (integer hash), loaded into
the reference's own nn.Modules with load_state_dict(strict=True).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Fixtures hold inputs' seeds/shapes and expected OUTPUTS only (data, not code).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("INNFER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

from innfer_amd import synth  # noqa: E402
import run as ref_run  # noqa: E402  (reference run.py)
from utils import utils as ref_utils  # noqa: E402
from utils import colors as ref_colors  # noqa: E402
from utils.defaults import get_network_G_config  # noqa: E402
from architectures import get_network  # noqa: E402

torch.set_num_threads(8)


def t_sd(sd_np):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def ref_net(kind, scale, **kw):
    d = dict(type=kind, **kw)
    return get_network(get_network_G_config(d, scale))


def rrdb_ref(nb, scale, seed=0, plus=False):
    sd = synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale, plus=plus), seed)
    net = ref_net("esrgan", scale, nb=nb, plus=plus)
    net.load_state_dict(t_sd(sd), strict=True)
    return net.eval(), sd


# ---------------------------------------------------------------- G1 geometry
def g1():
    out = {}
    for (h, w) in [(128, 128), (200, 200), (250, 330), (1080, 1920), (2160, 3840),
                   (4320, 7680), (201, 640), (150, 250), (540, 960), (400, 400)]:
        ps = min(h, w, 200)
        yy = torch.arange(h, dtype=torch.int16)[:, None].expand(h, w)
        xx = torch.arange(w, dtype=torch.int16)[None, :].expand(h, w)
        img = torch.stack([yy, xx], 0)[None].contiguous()
        p = ref_utils.extract_patches_2d(img, (ps, ps), [0.5, 0.5], batch_first=True).squeeze(0)
        org = p[:, :, 0, 0].numpy().astype(np.int32)          # [n, 2] (y, x)
        assert int(p[0].shape[-1]) == ps
        out[f"org_{h}x{w}"] = org
    save("g1_geometry", **out)


# ------------------------------------------------------------------- G2 blend
def g2():
    out = {}
    for P, scale in [(800, 4), (200, 1), (512, 2), (400, 2), (600, 4), (150, 1)]:
        step = 0.5
        overlap = scale * int(round((1.0 - step) * (P / scale)))
        prof = torch.cat([torch.linspace(0.1, 1.0, overlap), torch.ones(P - 2 * overlap),
                          torch.linspace(1.0, 0.1, overlap)], 0)
        out[f"profile_P{P}_s{scale}"] = prof.numpy()
    for scale in (1, 2, 4):
        h, w = 250, 330
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 100 + scale))
        p = ref_utils.extract_patches_2d(x, (200, 200), [0.5, 0.5], batch_first=True).squeeze(0)
        up = torch.nn.functional.interpolate(p, scale_factor=float(scale), mode="nearest") if scale > 1 else p
        # perturb tiles so that the blend is not an identity: tile k scaled by (1 + k/16)
        k = torch.arange(up.shape[0], dtype=torch.float32)[:, None, None, None]
        tiles = up * (1.0 + k / 16.0)
        r = ref_utils.recompose_tensor(tiles, h, w, step=0.5, scale=scale)
        out[f"blend_s{scale}_sub"] = r[0, :, ::7, ::5].numpy()
        out[f"blend_s{scale}_sum"] = np.float64(r.double().sum().item())
        r_id = ref_utils.recompose_tensor(up, h, w, step=0.5, scale=scale)
        ref_id = torch.nn.functional.interpolate(x, scale_factor=float(scale), mode="nearest") if scale > 1 else x
        out[f"ident_s{scale}_maxerr"] = np.float64((r_id - ref_id).abs().max().item())
    # small-image case (patch = min(H,W) < 200, even)
    h, w = 150, 250
    x = torch.from_numpy(synth.uniform((1, 3, h, w), 77))
    p = ref_utils.extract_patches_2d(x, (150, 150), [0.5, 0.5], batch_first=True).squeeze(0)
    k = torch.arange(p.shape[0], dtype=torch.float32)[:, None, None, None]
    r = ref_utils.recompose_tensor(p * (1.0 + k / 16.0), h, w, step=0.5, scale=1)
    out["blend_150x250"] = r[0].numpy()
    save("g2_blend", **out)


# ------------------------------------------------------- G3/G11 RRDBNet-23 4x
def g3():
    net, _ = rrdb_ref(23, 4)
    x = torch.from_numpy(synth.uniform((1, 3, 32, 32), 3))
    with torch.no_grad():
        y = net(x)
    out = {"out_32": y.numpy()}
    x16 = torch.from_numpy(synth.uniform((1, 3, 16, 16), 4))
    taps = {}
    m = net.model
    hooks = [m[0].register_forward_hook(lambda _m, _i, o: taps.__setitem__("conv_first", o.detach().numpy())),
             m[1].sub[0].register_forward_hook(lambda _m, _i, o: taps.__setitem__("rrdb0", o.detach().numpy())),
             m[1].sub[0].RDB1.register_forward_hook(lambda _m, _i, o: taps.__setitem__("rdb0", o.detach().numpy())),
             m[1].register_forward_hook(lambda _m, _i, o: taps.__setitem__("trunk", o.detach().numpy())),
             m[4].register_forward_hook(lambda _m, _i, o: taps.__setitem__("up0", o.detach().numpy())),
             m[7].register_forward_hook(lambda _m, _i, o: taps.__setitem__("up1", o.detach().numpy()))]
    with torch.no_grad():
        y16 = net(x16)
    for h in hooks:
        h.remove()
    out["out_16"] = y16.numpy()
    for k, v in taps.items():
        out["tap16_" + k] = v
    save("g3_rrdbnet23_x4", **out)
    # G11: the reference's fp16 mode restated on CPU (net.half(), input.half())
    neth = net.half()
    with torch.no_grad():
        yh = neth(x.half())
        yh16 = neth(x16.half())
    save("g11_rrdbnet23_x4_fp16", out_32=yh.float().numpy(), out_16=yh16.float().numpy())


# ------------------------------------------- G4 chop through Model.__call__
def g4():
    tmp = tempfile.mkdtemp()
    out = {}
    for (nb, scale, h, w, tag) in [(2, 4, 250, 330, "x4_250x330"), (1, 2, 201, 640, "x2_201x640"),
                                   (1, 1, 150, 250, "x1_150x250")]:
        sd = synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), 0)
        path = os.path.join(tmp, f"{scale}x_synth_{tag}.pth")
        torch.save(t_sd(sd), path)
        mdl = ref_run.Model(path, arch="infer", scale=None, device="cpu", chop=True)
        assert mdl.scale == scale and mdl.arch == "esrgan"
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 40 + scale))
        y = mdl(x)
        out[f"chop_{tag}_sub"] = y[0, :, ::8, ::8].numpy()
        out[f"chop_{tag}_sum"] = np.float64(y.double().sum().item())
        # crops across seams (tile origins are multiples of 100*scale / ragged)
        s = scale
        out[f"chop_{tag}_crop_a"] = y[0, :, 100 * s - 16:100 * s + 16, 100 * s - 16:100 * s + 16].numpy() \
            if h > 216 else y[0, :, :32, :32].numpy()
        out[f"chop_{tag}_crop_b"] = y[0, :, -32:, -32:].numpy()
        mdl2 = ref_run.Model(path, arch="infer", scale=None, device="cpu", chop=False)
        y2 = mdl2(x)
        out[f"nochop_{tag}_sub"] = y2[0, :, ::8, ::8].numpy()
    save("g4_chop", **out)


# --------------------------------------------------------- G5 scale variants
def g5():
    out = {}
    x = torch.from_numpy(synth.uniform((1, 3, 16, 16), 5))
    for scale in (1, 2, 3, 8):
        net, _ = rrdb_ref(1, scale)
        with torch.no_grad():
            out[f"out_x{scale}"] = net(x).numpy()
    net, _ = rrdb_ref(1, 4, plus=True)
    with torch.no_grad():
        out["out_x4_plus"] = net(x).numpy()
    for fa in ("tanh", "sigmoid"):
        net = ref_net("esrgan", 4, nb=1, finalact=fa)
        net.load_state_dict(t_sd(synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=4), 0)), strict=True)
        with torch.no_grad():
            out[f"out_x4_{fa}"] = net.eval()(x).numpy()
    save("g5_scales", **out)


# ------------------------------------------------------------------ G6 SRGAN
def g6():
    sd = synth.fill_state_dict(synth.srresnet_shapes(nb=16, scale=4), 0)
    net = ref_net("srgan", 4)
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()
    x = torch.from_numpy(synth.uniform((1, 3, 24, 24), 6))
    with torch.no_grad():
        y = net(x)
    ps_in = torch.arange(1 * 8 * 3 * 5, dtype=torch.float32).reshape(1, 8, 3, 5)
    save("g6_srgan", out_24=y.numpy(), ps_in=ps_in.numpy(),
         ps_out=torch.nn.PixelShuffle(2)(ps_in).numpy())


# ------------------------------------------------------------------- G7 UNet
def g7():
    net = ref_net("unet_256", 1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.train()                                          # meval=False (run.py:299-303,98-99)
    xa = torch.from_numpy(synth.uniform((1, 3, 256, 256), 7, -1.0, 1.0))
    with torch.no_grad():
        ya = net(xa)
    keys = np.array(sorted(shapes.keys()))
    save("g7_unet256", out_a=ya.numpy().astype(np.float16), out_a_sub=ya[0, :, ::4, ::4].numpy(),
         keys=keys, shapes=np.array([str(shapes[k]) for k in keys]))


# -------------------------------------------------------------------- G8 PAN
def g8():
    net = ref_net("pan", 4)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()
    out = {"keys": np.array(list(shapes.keys())), "shapes": np.array([str(shapes[k]) for k in shapes])}
    for (h, w, seed) in [(48, 48, 8), (50, 70, 9)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed))
        with torch.no_grad():
            out[f"out_{h}x{w}"] = net(x).numpy()
    save("g8_pan", **out)


# ---------------------------------------------------------------- G13 PPON
def g13():
    net = ref_net("ppon", 4)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()
    out = {"keys": np.array(list(shapes.keys())), "shapes": np.array([str(shapes[k]) for k in shapes])}
    for (h, w, seed) in [(24, 24, 13), (20, 28, 14)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed))
        with torch.no_grad():
            oc, os_, op = net(x)
        out[f"out_c_{h}x{w}"], out[f"out_s_{h}x{w}"], out[f"out_p_{h}x{w}"] = oc.numpy(), os_.numpy(), op.numpy()
    save("g13_ppon", **out)


# ---------------------------------------------------------------- G14 CycleGAN ResNet
def g14():
    net = ref_net("resnet_9blocks", 1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()                                                   # cyglegan_extras: meval=True (run.py:305-309)
    out = {"keys": np.array(list(shapes.keys())), "shapes": np.array([str(shapes[k]) for k in shapes])}
    for (h, w, seed) in [(32, 40, 15), (64, 64, 16)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0))
        with torch.no_grad():
            out[f"out_{h}x{w}"] = net(x).numpy()
    save("g14_resnet9", **out)


# ---------------------------------------------------------------- G15 WBC UNet + guided filter
def g15():
    net = ref_net("wbcunet", 1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()
    out = {"keys": np.array(list(shapes.keys())), "shapes": np.array([str(shapes[k]) for k in shapes])}
    for (h, w, seed) in [(32, 40, 17), (64, 64, 18)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0))
        with torch.no_grad():
            y = net(x)
            out[f"out_{h}x{w}"] = y.numpy()
            out[f"gf_{h}x{w}"] = ref_utils.guided_filter(x, y, r=1, eps=5e-3).numpy()      # run.py:427-429
    net_tf = ref_net("wbcunet_tf", 1)
    net_tf.load_state_dict(t_sd(sd), strict=True)
    net_tf.eval()
    x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 17, -1.0, 1.0))
    with torch.no_grad():
        out["out_tf_32x40"] = net_tf(x).numpy()
    save("g15_wbcunet", **out)


# ---------------------------------------------------------------- G9 convert
def g9():
    ramp = (np.arange(5 * 7 * 3) * 37 % 256).astype(np.uint8).reshape(5, 7, 3)
    t0 = ref_utils.np2tensor(ramp, normalize=False)
    t1 = ref_utils.np2tensor(ramp, normalize=True)
    # values that land exactly on .5 after *255 (0.5/255 -> 0, 1.5/255 -> 2, ...)
    halves = torch.tensor([(k + 0.5) / 255.0 for k in range(0, 12)] + [-0.1, 1.2, 0.999, 0.5],
                          dtype=torch.float32)
    th = halves.reshape(1, 1, 4, 4).repeat(1, 3, 1, 1).contiguous()
    th[0, 1] += 1.0 / 1024
    th[0, 2] -= 1.0 / 1024
    u0 = ref_utils.tensor2np(th, denormalize=False)
    u1 = ref_utils.tensor2np(th * 2 - 1, denormalize=True)
    big = torch.from_numpy(synth.uniform((1, 3, 33, 47), 9, -0.2, 1.2))
    lin = ref_colors.srgb2linear(np.arange(256, dtype=np.uint8))
    back = ref_colors.linear2srgb(np.linspace(-0.1, 1.1, 1001, dtype=np.float32))
    save("g9_convert", ramp=ramp, np2t=t0.numpy(), np2t_norm=t1.numpy(), t2np_in=th.numpy(),
         t2np=u0, t2np_denorm=u1, big_in=big.numpy(), big_u8=ref_utils.tensor2np(big),
         srgb2linear=lin, linear2srgb=back)


def g19():
    """np2tensor / tensor2np beyond the defaults (utils.py:164-248): uint16 images (maxval 65535, utils.py:22-33), bgr2rgb / rgb2bgr off,
    add_batch off, change_range off, (data_range, imtype) = (65535, uint16), 3-D and 2-D tensors."""
    ramp16 = (np.arange(5 * 7 * 3) * 997 % 65536).astype(np.uint16).reshape(5, 7, 3)
    ramp8 = (np.arange(5 * 7 * 3) * 37 % 256).astype(np.uint8).reshape(5, 7, 3)
    ramp4 = (np.arange(4 * 6 * 4) * 1201 % 65536).astype(np.uint16).reshape(4, 6, 4)
    halves = torch.tensor([(k + 0.5) / 65535.0 for k in range(0, 12)] + [-0.1, 1.2, 0.99999, 0.5], dtype=torch.float32)
    th = halves.reshape(1, 1, 4, 4).repeat(1, 3, 1, 1).contiguous()
    th[0, 1] += 1.0 / (1 << 18)
    th[0, 2] -= 1.0 / (1 << 18)
    big = torch.from_numpy(synth.uniform((1, 3, 21, 30), 19, -0.2, 1.2))
    save("g19_convert_flags", ramp16=ramp16, ramp8=ramp8, ramp4=ramp4,
         np2t16=ref_utils.np2tensor(ramp16).numpy(), np2t16_norm=ref_utils.np2tensor(ramp16, normalize=True).numpy(),
         np2t4=ref_utils.np2tensor(ramp4).numpy(),
         np2t8_noflip=ref_utils.np2tensor(ramp8, bgr2rgb=False).numpy(), np2t8_nobatch=ref_utils.np2tensor(ramp8, add_batch=False).numpy(),
         np2t8_norange=ref_utils.np2tensor(ramp8, change_range=False).numpy(),
         t2np_in=th.numpy(), t2np16=ref_utils.tensor2np(th, data_range=65535, imtype=np.uint16),
         t2np16_denorm=ref_utils.tensor2np(th * 2 - 1, denormalize=True, data_range=65535, imtype=np.uint16),
         big_in=big.numpy(), big_u16=ref_utils.tensor2np(big, data_range=65535, imtype=np.uint16),
         big_noflip=ref_utils.tensor2np(big, rgb2bgr=False), big_3d=ref_utils.tensor2np(big[0]), big_2d=ref_utils.tensor2np(big[0, 1]))


def g20():
    """Command-line helpers of run.py (:227-315): scale from the file name, model lookup, the per-architecture presets; modcrop (utils.py:250-264)."""
    import json
    names = ["4x_foo.pth", "/a/b/1x_JPEG.pth", "models/2X_up.pth", "16x.pth", "x4_bar.pth", "8x", "foo.pth", "3x_.pth", "xx.pth", "4xfoo+1x.pth"]
    scales = [ref_run.get_scale_name(n) for n in names]
    with_arg = [ref_run.get_scale_name("4x_foo.pth", 2), ref_run.get_scale_name("foo.pth", 2)]
    extras = {"pix2pix": ref_run.pix2pix_extras, "cyclegan": ref_run.cyglegan_extras, "default": ref_run.default_extras}
    crops = {f"{h}x{w}x{c}@{s}": list(ref_utils.modcrop(np.zeros((h, w, c) if c else (h, w), np.uint8), s).shape)
             for (h, w, c, s) in [(13, 17, 3, 4), (16, 16, 3, 4), (7, 9, 0, 2), (5, 5, 4, 8)]}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "models", "sub"))
    for f in ("4x_alpha.pth", "1x_beta.pth", os.path.join("sub", "2x_gamma.pth")):
        open(os.path.join(tmp, "models", f), "wb").close()
    os.chdir(tmp)
    try:
        chain = {q: [[os.path.relpath(p, tmp) for p in ref_run.parse_models(q)[0]], ref_run.parse_models(q)[1]]
                 for q in ("4x_alpha.pth", "alpha", "4x_alpha.pth+1x_beta.pth", "beta>gamma", "models/1x_beta.pth")}
    finally:
        os.chdir(cwd)
    save("g20_cli", table=np.array(json.dumps({"names": names, "scales": scales, "with_arg": with_arg, "extras": extras, "crops": crops, "chain": chain})))


def g24():
    """The small helpers of utils/utils.py and utils/colors.py that the image path inlines elsewhere: norm / denorm (:136-161), get_box_kernel /
    normalize_kernel2d / compute_padding (:448-481, 538-546), the channel flips (colors.py:5-26), normal2mod (:629-663)."""
    import json
    x = synth.uniform((2, 3, 5, 7), 241, -1.5, 1.5)
    t = torch.from_numpy(x)
    k = torch.from_numpy(synth.uniform((2, 3, 5), 242, -1.0, 1.0))
    rgba = torch.from_numpy(synth.uniform((4, 3, 5), 243))
    pads = {str(ks): ref_utils.compute_padding(ks) for ks in (3, 4, 7, (3, 3), (4, 4), (3, 5), (4, 7), [2, 6], (3, 4, 5))}
    shapes = synth.rrdbnet_shapes(nb=23, scale=4)
    old = {kk: np.zeros((1,), np.float32) for kk in shapes}                      # key mapping only: old-arch keys -> new-arch keys and back
    new_keys = list(ref_utils.normal2mod(dict(old)).keys())
    save("g24_helpers", x=x, norm_t=ref_utils.norm(t).numpy(), norm_np=ref_utils.norm(x), denorm_t=ref_utils.denorm(t).numpy(),
         denorm_np=ref_utils.denorm(x, (-0.5, 1.25)), box5=ref_utils.get_box_kernel(5).numpy(), box37=ref_utils.get_box_kernel([3, 7]).numpy(),
         k=k.numpy(), k_norm=ref_utils.normalize_kernel2d(k).numpy(), rgba=rgba.numpy(), bgr2rgb=ref_colors.bgr_to_rgb(t).numpy(),
         rgb2bgr=ref_colors.rgb_to_bgr(t[0]).numpy(), bgra2rgba=ref_colors.bgra_to_rgba(rgba).numpy(), rgba2bgra=ref_colors.rgba_to_bgra(rgba).numpy(),
         table=np.array(json.dumps({"pads": pads, "old_keys": list(old.keys()), "new_keys": new_keys})))


def g25():
    """filter2D (utils.py:484-535): the four border modes, odd / even kernels, a normalised random kernel."""
    x = torch.from_numpy(synth.uniform((2, 3, 11, 13), 251, -1.0, 1.0))
    out = {"x": x.numpy()}
    k33 = torch.from_numpy(synth.uniform((1, 3, 3), 252, -1.0, 1.0))
    k57 = torch.from_numpy(synth.uniform((1, 5, 7), 253, 0.0, 1.0))
    k44 = torch.from_numpy(synth.uniform((1, 4, 4), 254, -1.0, 1.0))
    out.update(k33=k33.numpy(), k57=k57.numpy(), k44=k44.numpy())
    for b in ("constant", "reflect", "replicate", "circular"):
        out["k33_" + b] = ref_utils.filter2D(x, k33, border_type=b).numpy()
        out["k57n_" + b] = ref_utils.filter2D(x, k57, border_type=b, normalized=True).numpy()
        out["k44_" + b] = ref_utils.filter2D(x, k44, border_type=b).numpy()
    out["box5"] = ref_utils.filter2D(x, ref_utils.get_box_kernel(5).unsqueeze(0)).numpy()
    save("g25_filter2d", **out)


def g26():
    """SRResNet(norm_type, mode) (SRResNet_arch.py:16-27,68-91; block.py:242-254) in eval mode -- the class's own defaults are norm_type='batch',
    mode='NAC' -- and upscale=3 with 'upconv'.  Keys (in state-dict order) are stored too."""
    from architectures.SRResNet_arch import SRResNet as RefSRResNet
    out = {}
    for j, (tag, kw) in enumerate(G26_SR.items()):
        net = RefSRResNet(3, 3, 64, 2, upsample_mode="upconv", **kw).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        norm, mode, sc = bool(kw.get("norm_type", "batch")), kw.get("mode", "NAC"), kw.get("upscale", 4)
        mine = synth.srresnet_shapes(nb=2, scale=sc, upsample_mode="upconv", norm=norm, mode=mode)
        assert list(shapes.items()) == list(mine.items()), tag
        sd = synth.fill_state_dict(mine, 300 + j)
        if norm:
            sd = synth.fill_running_stats(sd, 300 + j)
        net.load_state_dict(t_sd(sd), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 14, 18), 310 + j))
        with torch.no_grad():
            out[tag] = net(x).numpy()
        out[tag + "_keys"] = np.array(list(shapes))
    # RRDBNet(norm_type='batch', mode='NAC' / 'CNAC'): only LR_conv sees the mode (RRDBNet_arch.py:27-29) -- 'NAC' puts its BatchNorm2d in front
    from architectures.RRDBNet_arch import RRDBNet as RefRRDBNet
    for j, mode in enumerate(("NAC", "CNAC")):
        net = RefRRDBNet(3, 3, 64, 2, upscale=2, norm_type="batch", mode=mode).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        mine = synth.rrdbnet_shapes(nb=2, scale=2, norm=True, mode=mode)
        assert list(shapes.items()) == list(mine.items()), mode
        sd = synth.fill_running_stats(synth.fill_state_dict(mine, 320 + j), 320 + j)
        net.load_state_dict(t_sd(sd), strict=True)
        with torch.no_grad():
            out["rrdb_bn_" + mode] = net(torch.from_numpy(synth.uniform((1, 3, 16, 16), 330 + j))).numpy()
        out[f"rrdb_bn_{mode}_keys"] = np.array(list(shapes))
    # PixelShuffle(3) and PixelShuffle(2) on 32 features (block.py:333-346): SRResNet x3, SRResNet nf=32 x4, RRDBNet x3
    for tag, (net, mine, seed) in {
            "ps3_sr": (RefSRResNet(3, 3, 64, 2, upscale=3, norm_type=None, mode="CNA", upsample_mode="pixelshuffle"), synth.srresnet_shapes(nb=2, scale=3, upsample_mode="pixelshuffle"), 360),
            "ps2_sr_nf32": (RefSRResNet(3, 3, 32, 2, upscale=4, norm_type=None, mode="CNA", upsample_mode="pixelshuffle"), synth.srresnet_shapes(nf=32, nb=2, scale=4, upsample_mode="pixelshuffle"), 361),
            "ps3_rrdb": (RefRRDBNet(3, 3, 64, 1, upscale=3, upsample_mode="pixelshuffle"), synth.rrdbnet_shapes(nb=1, scale=3, upsample_mode="pixelshuffle"), 362)}.items():
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert list(shapes.items()) == list(mine.items()), tag
        net.load_state_dict(t_sd(synth.fill_state_dict(mine, seed)), strict=True)
        with torch.no_grad():
            out[tag] = net.eval()(torch.from_numpy(synth.uniform((1, 3, 10, 12), seed + 10))).numpy()
        out[tag + "_keys"] = np.array(list(shapes))
    save("g26_srresnet_modes", **out)


def g27():
    """PPON with upscale 8 (three upconv stages) and 3 (one Upsample(3) stage) (PPON_arch.py:16-63), two-block trunk, all three outputs."""
    from architectures.PPON_arch import PPON as RefPPON
    out = {}
    for j, sc in enumerate((8, 3, 2)):
        net = RefPPON(3, 64, 2, 3, upscale=sc).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(t_sd(synth.fill_state_dict(shapes, 340 + j)), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 10, 12), 350 + j))
        with torch.no_grad():
            oc, os_, op = net(x)
        out[f"x{sc}_c"], out[f"x{sc}_s"], out[f"x{sc}_p"] = oc.numpy(), os_.numpy(), op.numpy()
        out[f"x{sc}_keys"], out[f"x{sc}_shapes"] = np.array(list(shapes)), np.array([str(shapes[k]) for k in shapes])
    save("g27_ppon_scales", **out)


G26_SR = {"nac_bn": dict(), "cna_bn": dict(mode="CNA"), "nac": dict(norm_type=None), "cnac_bn": dict(mode="CNAC"), "cnac": dict(norm_type=None, mode="CNAC"),
          "x3_lrelu": dict(norm_type=None, mode="CNA", upscale=3, act_type="leakyrelu", res_scale=0.5), "nac_bn_x2_lrelu": dict(upscale=2, act_type="leakyrelu", res_scale=0.25)}


def g21():
    """guided_filter beyond r = 1 / 'regular' (utils.py:548-626): a 5x5 and a 7x7 window, and the 'fast' mode on a 2x guidance image."""
    x = torch.from_numpy(synth.uniform((2, 3, 23, 31), 211))
    y = torch.from_numpy(synth.uniform((2, 3, 23, 31), 212))
    xh = torch.from_numpy(synth.uniform((2, 3, 46, 62), 213))
    save("g21_guided", r2=ref_utils.guided_filter(x, y, r=2, eps=5e-3).numpy(), ks7=ref_utils.guided_filter(x, y, ks=7, eps=1e-2).numpy(),
         fast=ref_utils.guided_filter(x, y, x_HR=xh, r=1, eps=5e-3, mode='fast').numpy(),
         fast_r2=ref_utils.guided_filter(x, y, x_HR=xh, r=2, eps=1e-2, mode='fast').numpy(),
         # the forms beyond the fused kernels: a caller's conv_a ('conv' mode), an even window, a precomputed (here: non-uniform) box kernel
         conv_w=_g21_conv_a()[1], conv_b=_g21_conv_a()[2], conv=ref_utils.guided_filter(x, y, x_HR=xh, r=1, mode='conv', conv_a=_g21_conv_a()[0]).detach().numpy(),
         ks4=ref_utils.guided_filter(x, y, ks=4, eps=1e-2).numpy(),
         bk=_g21_bk().numpy(), bk_out=ref_utils.guided_filter(x, y, box_kernel=_g21_bk(), eps=1e-2).numpy())


def _g21_conv_a():
    w, b = synth.uniform((3, 6, 1, 1), 214, -0.5, 0.5), synth.uniform((3,), 215, -0.1, 0.1)
    m = torch.nn.Sequential(torch.nn.Conv2d(6, 3, 1))
    with torch.no_grad():
        m[0].weight.copy_(torch.from_numpy(w)); m[0].bias.copy_(torch.from_numpy(b))
    return m, w, b


def _g21_bk():
    k = torch.from_numpy(synth.uniform((3, 5), 216, 0.5, 1.5))
    return k / k.sum()


def g22():
    """ResnetGenerator(padding_type='zero' / 'replicate', use_dropout=True) in eval mode (ResNet_arch.py:20-146), 2 blocks, 32x40 input."""
    from architectures.ResNet_arch import ResnetGenerator as RefResnet
    out = {}
    for i, (tag, kw) in enumerate({"zero": dict(padding_type="zero"), "replicate": dict(padding_type="replicate"),
                                   "reflect_dropout": dict(padding_type="reflect", use_dropout=True),
                                   "zero_dropout": dict(padding_type="zero", use_dropout=True),
                                   "upconv": dict(upsample_mode="upconv"),
                                   "batch_eval": dict(norm_type="batch"), "batch_train": dict(norm_type="batch", train=True),
                                   "batch_zero_upconv_eval": dict(norm_type="batch", padding_type="zero", upsample_mode="upconv")}.items()):
        kw = dict(kw)
        train = kw.pop("train", False)
        net = RefResnet(3, 3, 64, n_blocks=2, **{"norm_type": "instance", **kw})
        net = net.train() if train else net.eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = synth.fill_state_dict(shapes, 220 + i)
        if kw.get("norm_type") == "batch":
            sd = synth.fill_running_stats(sd, 228 + i)
        net.load_state_dict(t_sd(sd), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 225 + i, -1.0, 1.0))
        with torch.no_grad():
            out[tag] = net(x).numpy()
        out[tag + "_keys"] = np.array(list(shapes.keys()))
        out[tag + "_shapes"] = np.array([str(shapes[k]) for k in shapes])
    save("g22_resnet_variants", **out)


def g23():
    """UnetGenerator(norm_type='instance'), (use_dropout=True, eval) and (upsample_mode='upconv') (UNet_arch.py:20-157): 5 downs, ngf 32, 64x96 input."""
    from architectures.UNet_arch import UnetGenerator as RefUnet
    out = {}
    for i, (tag, kw, ev) in enumerate([("instance", dict(norm_type="instance"), False), ("instance_eval", dict(norm_type="instance"), True),
                                       ("batch_dropout_eval", dict(norm_type="batch", use_dropout=True), True),
                                       ("batch_upconv", dict(norm_type="batch", upsample_mode="upconv"), False),
                                       ("batch_upconv_eval", dict(norm_type="batch", upsample_mode="upconv"), True),
                                       ("instance_upconv", dict(norm_type="instance", upsample_mode="upconv"), False)]):
        net = RefUnet(3, 3, 5, ngf=32, **kw)
        net = net.eval() if ev else net.train()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = synth.fill_running_stats(synth.fill_state_dict(shapes, 230 + i), 235 + i) if "batch" in tag else synth.fill_state_dict(shapes, 230 + i)
        net.load_state_dict(t_sd(sd), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 64, 96), 240 + i, -1.0, 1.0))
        with torch.no_grad():
            out[tag] = net(x).numpy()
        out[tag + "_keys"] = np.array(list(shapes.keys()))
        out[tag + "_shapes"] = np.array([str(shapes[k]) for k in shapes])
    save("g23_unet_variants", **out)


# ----------------------------------------------------------------- G10 loader
def g10():
    tmp = tempfile.mkdtemp()
    rows = []

    def probe(sd, name):
        path = os.path.join(tmp, name)
        torch.save(t_sd(sd), path)
        m = ref_run.Model(path, arch="infer", scale=None, device="cpu", chop=False)
        net = m.model
        first = net.model[0]
        nb = len(net.model[1].sub) - 1
        rows.append((name, m.arch, m.scale, first.out_channels, nb, m.in_nc, m.out_nc,
                     sorted(net.state_dict().keys())))

    for scale in (1, 2, 4, 8):
        probe(synth.fill_state_dict(synth.rrdbnet_shapes(nb=2, scale=scale, nf=48, in_nc=3, out_nc=3), 0),
              f"{scale}x_old.pth")
    probe(synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=4, in_nc=1, out_nc=1), 0), "4x_gray.pth")
    probe(synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=4, plus=True), 0), "4x_plus.pth")
    probe(synth.fill_state_dict(synth.srresnet_shapes(nb=3, scale=4), 0), "4x_srgan.pth")
    # new-arch (MRRDBNet key names) 23-block model -> mod2normal -> old arch
    old = synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4, nf=16), 0)
    new = ref_utils.normal2mod(dict(old))
    probe(new, "4x_newarch.pth")
    new_keys = sorted(new.keys())
    # SWA-wrapped
    swa = {"n_averaged": np.asarray(3, dtype=np.int64)}
    small = synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=2), 0)
    for k, v in small.items():
        swa["module.module." + k] = v
    probe(swa, "2x_swa.pth")
    save("g10_loader",
         names=np.array([r[0] for r in rows]), arch=np.array([r[1] for r in rows]),
         scale=np.array([r[2] for r in rows]), nf=np.array([r[3] for r in rows]),
         nb=np.array([r[4] for r in rows]), in_nc=np.array([r[5] for r in rows]),
         out_nc=np.array([r[6] for r in rows]),
         keys=np.array(["|".join(r[7]) for r in rows]),
         newarch_keys=np.array(new_keys))


# ------------------------------------------------------- G12 default configs
# ---------------------------------------------------------------- G16 MRRDBNet ("new"-arch ESRGAN built directly)
def g16():
    shapes = synth.mrrdbnet_shapes(nb=2)
    sd = synth.fill_state_dict(shapes, 61)
    net = ref_net("mrrdb_net", 4, nb=2)
    assert list(net.state_dict().keys()) == list(shapes.keys())
    net.load_state_dict(t_sd(sd), strict=True)
    net.eval()
    x = torch.from_numpy(synth.uniform((2, 3, 16, 20), 62))
    with torch.no_grad():
        y = net(x)
    save("g16_mrrdb", out=y.numpy(), keys=np.array(list(shapes.keys())), shapes=np.array([str(tuple(v)) for v in shapes.values()]))


def g12():
    import json
    rows = {}
    for kind, scale, extra in [("esrgan", 4, {}), ("esrgan", 1, {}), ("rrdb_net", 2, {"nb": 5}), ("esrgan-lite", 4, {}),
                               ("srgan", 4, {}), ("sr_resnet", 2, {"nf": 32, "nb": 3}), ("srresnet", 8, {}),
                               ("esrgan", 4, {"in_nc": 1, "out_nc": 1, "nf": 48, "nb": 2, "plus": True})]:
        d = dict(type=kind, **extra)
        rows[f"{kind}|{scale}|{json.dumps(extra, sort_keys=True)}"] = get_network_G_config(d, scale)
    for kind in ("p2p_256", "unet_256", "unet_128", "p2p_128", "unet_512"):
        rows[f"str:{kind}|1"] = get_network_G_config(kind, 1)
    for kind in ("wbcunet", "wbcunet_tf"):
        rows[f"str:{kind}|1"] = get_network_G_config(kind, 1)
    for kind in ("resnet_9blocks", "resnet_6blocks", "cg_6", "cg9"):
        rows[f"str:{kind}|1"] = get_network_G_config(kind, 1)
    for kind, scale in (("ppon", 4), ("ppon", 2)):
        rows[f"str:{kind}|{scale}"] = get_network_G_config(kind, scale)
    for kind, scale in (("pan", 4), ("pan_net", 2), ("pan", 1)):
        rows[f"str:{kind}|{scale}"] = get_network_G_config(kind, scale)
    rows["pan|4|" + json.dumps({"nb": 3, "in_nc": 1, "out_nc": 1}, sort_keys=True)] = get_network_G_config(
        dict(type="pan", nb=3, in_nc=1, out_nc=1), 4)
    rows["str:esrgan|4"] = get_network_G_config("esrgan", 4)
    rows["str:mesrgan|4"] = get_network_G_config("mesrgan", 4)
    rows["mrrdb_net|4|" + json.dumps({"nb": 2, "nf": 32}, sort_keys=True)] = get_network_G_config(dict(type="mrrdb_net", nb=2, nf=32), 4)
    rows["which_model_G:srgan|4"] = get_network_G_config({"which_model_G": "srgan"}, 4)
    save("g12_defaults", table=np.array(json.dumps(rows, sort_keys=True)))


# ---------------------------------------------------------------- G17: the reference's own fp16 mode for the normalised generators
def g17():
    """(a) pix2pix UNet_256 and CycleGAN ResNet-9 in the reference's fp16 mode (`net.half()`, run.py:383) on the inputs of G7 / G14, run
    on the CPU: separates "fp16 rounding of a network with BatchNorm / InstanceNorm + tanh" from "kernel bug" the way G11 does for RRDBNet;
    (b) UNet_256 in EVAL mode (Model's default meval=True, run.py:96-97) on non-trivial running statistics, fp32."""
    out = {}
    net = ref_net("unet_256", 1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, 0)
    net.load_state_dict(t_sd(sd), strict=True)
    net.train()
    xa = torch.from_numpy(synth.uniform((1, 3, 256, 256), 7, -1.0, 1.0))
    with torch.no_grad():
        y32 = net(xa)
        yh = net.half()(xa.half()).float()
    out["unet_fp16_out_a"] = yh.numpy().astype(np.float16)
    out["unet_fp16_err_vs_fp32"] = np.array([(yh - y32).abs().max().item(), (yh - y32).abs().mean().item()], np.float32)
    net = ref_net("unet_256", 1)
    sd_ev = synth.fill_running_stats(sd, 17)
    net.load_state_dict(t_sd(sd_ev), strict=True)
    net.eval()
    with torch.no_grad():
        yev = net(xa)
    out["unet_eval_out_a"] = yev.numpy().astype(np.float16)
    out["unet_eval_out_a_sub"] = yev[0, :, ::4, ::4].numpy()
    net = ref_net("resnet_9blocks", 1)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(t_sd(synth.fill_state_dict(shapes, 0)), strict=True)
    net.eval()
    for (h, w, seed) in [(32, 40, 15), (64, 64, 16)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0))
        with torch.no_grad():
            y32 = net.float()(x)
            yh = net.half()(x.half()).float()
        out[f"resnet_fp16_out_{h}x{w}"] = yh.numpy()
        out[f"resnet_fp16_err_vs_fp32_{h}x{w}"] = np.array([(yh - y32).abs().max().item(), (yh - y32).abs().mean().item()], np.float32)
    save("g17_fp16_and_eval", **out)


def g18():
    """Graph-changing constructor arguments of RRDBNet (RRDBNet_arch.py:16-48) built directly from the reference class: nr != 3 (`RDBs.<i>`),
    act_type='relu', mode='NAC' (only LR_conv sees it), upsample_mode='pixelshuffle'.  Keys are stored too: they pin the parameter names."""
    from architectures.RRDBNet_arch import RRDBNet as RefRRDBNet
    out = {}
    cases = {"ps4": (dict(nb=2, upscale=4, upsample_mode="pixelshuffle"), (1, 3, 12, 20)),
             "ps2_relu": (dict(nb=1, upscale=2, upsample_mode="pixelshuffle", act_type="relu"), (1, 3, 16, 16)),
             "nr2_relu_nac": (dict(nb=2, nr=2, upscale=2, act_type="relu", mode="NAC"), (1, 3, 16, 16)),
             "nr4": (dict(nb=1, nr=4, upscale=1), (1, 3, 10, 14))}
    for i, (tag, (kw, shape)) in enumerate(cases.items()):
        net = RefRRDBNet(3, 3, 64, **kw).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        mine = synth.rrdbnet_shapes(nb=kw["nb"], scale=kw["upscale"], nr=kw.get("nr", 3), upsample_mode=kw.get("upsample_mode", "upconv"))
        assert shapes == mine, (tag, set(shapes) ^ set(mine))
        sd = synth.fill_state_dict(mine, 180 + i)
        net.load_state_dict(t_sd(sd), strict=True)
        x = torch.from_numpy(synth.uniform(shape, 190 + i))
        with torch.no_grad():
            out[tag] = net(x).numpy()
        out[tag + "_keys"] = np.array(sorted(shapes))
    # norm_type='batch' under eval(): BatchNorm2d behind every conv of the dense blocks and behind LR_conv
    net = RefRRDBNet(3, 3, 64, 2, upscale=2, norm_type="batch").eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    mine = synth.rrdbnet_shapes(nb=2, scale=2, norm=True)
    assert shapes == mine, set(shapes) ^ set(mine)
    sd = synth.fill_running_stats(synth.fill_state_dict(mine, 184), 184)
    net.load_state_dict(t_sd(sd), strict=True)
    with torch.no_grad():
        out["batchnorm"] = net(torch.from_numpy(synth.uniform((1, 3, 16, 16), 194))).numpy()
    out["batchnorm_keys"] = np.array(sorted(shapes))
    # SRResNet(act_type / res_scale / upsample_mode='upconv') and the `outm` argument of both forwards
    from architectures.SRResNet_arch import SRResNet as RefSRResNet
    for j, (tag, kw) in enumerate({"sr_upconv_lrelu": dict(upscale=2, act_type="leakyrelu", upsample_mode="upconv", res_scale=0.5),
                                   "sr_ps_scale": dict(upscale=4, act_type="relu", upsample_mode="pixelshuffle", res_scale=0.25)}.items()):
        net = RefSRResNet(3, 3, 64, 2, norm_type=None, mode="CNA", **kw).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        mine = synth.srresnet_shapes(nb=2, scale=kw["upscale"], upsample_mode=kw["upsample_mode"])
        assert shapes == mine, (tag, set(shapes) ^ set(mine))
        net.load_state_dict(t_sd(synth.fill_state_dict(mine, 186 + j)), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 14, 18), 196 + j))
        with torch.no_grad():
            out[tag] = net(x).numpy()
            for om in ("scaltanh", "tanh", "sigmoid", "clamp"):
                out[f"{tag}_{om}"] = net(x, outm=om).numpy()
        out[tag + "_keys"] = np.array(sorted(shapes))
    net, _ = rrdb_ref(1, 2)
    x = torch.from_numpy(synth.uniform((1, 3, 12, 12), 198))
    with torch.no_grad():
        for om in ("scaltanh", "clamp"):
            out[f"rrdb_{om}"] = net(x, outm=om).numpy()
    save("g18_rrdb_variants", **out)
    # PAN(self_attention=False), PAN(double_scpa=True) (PAN_arch.py:115-141,193-203)
    from architectures.PAN_arch import PAN as RefPAN
    out = {}
    for i, (tag, kw) in enumerate({"noattn": dict(self_attention=False), "double": dict(double_scpa=True),
                                   "double_noattn_x2": dict(double_scpa=True, self_attention=False, scale=2),
                                   "bilinear": dict(ups_inter_mode="bilinear"),
                                   "bilinear_noattn_x2": dict(ups_inter_mode="bilinear", self_attention=False, scale=2),
                                   "x3": dict(scale=3), "bilinear_x3": dict(ups_inter_mode="bilinear", scale=3, self_attention=False)}.items()):
        net = RefPAN(3, 3, 40, 24, 3, **kw).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = synth.fill_state_dict(shapes, 185 + i)
        net.load_state_dict(t_sd(sd), strict=True)
        x = torch.from_numpy(synth.uniform((1, 3, 20, 28), 195 + i))
        with torch.no_grad():
            out[tag] = net(x).numpy()
        out[tag + "_keys"] = np.array(list(shapes.keys()))
        out[tag + "_shapes"] = np.array([str(shapes[k]) for k in shapes])
    save("g18_pan_variants", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22", "g23", "g24", "g25", "g26", "g27"]
    for g in which:
        globals()[g]()
