"""The fp32-accurate forward (innfer_net_set_precision(1); `-no_fp16` of the reference's command line, run.py:345,421-422) against the fp32
golden vectors generated from the reference.  Needs an MI355X: `pytest -m gpu`.

Tolerance: fp32 path vs G3 / G4 max-abs <= 1e-4 on [0,1]-scaled outputs.  The engine keeps every value as a pair of fp16
slabs (22 significant bits) and multiplies on the fp16 matrix cores with fp32 accumulation, so the expected error is ~1e-6; the single-conv
tests below hold the kernel to 3e-6 of the output range against float64 convolutions of the same fp32 operands.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _sd(shapes, seed=0):
    from innfer_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed).items()}


def _split_slab(dev, x, groups, Hs, Ws):
    """fp32 NCHW (cpu) -> (tensor holding the hi slab [groups,N,H,W,32] followed by its lo twin, group stride, lo distance) on the GPU."""
    import innfer_amd.lib as L
    N, Cc = x.shape[:2]
    g = N * Hs * Ws * 32
    lo = groups * g
    buf = torch.full((2, groups, N, Hs, Ws, 32), 7.0, dtype=torch.float16, device=dev)       # junk in unused groups
    L.check(L.lib.innfer_nchw_to_slab_split(x.to(dev).contiguous().data_ptr(), buf.data_ptr(), g, lo, 0, N, Cc, Hs, Ws, None))
    return buf, g, lo


def _run_split_conv(dev, x, w, b, K, act=0, up=False, res1=None, s1=1.0, res2=None, s2=1.0, in_extra=0, rows=None, out_off=0, out_channels=None):
    """x [N,C,Hs,Ws] fp32 (cpu), w [K,C,3,3] fp32: the split conv through the C ABI, result NCHW fp32 on the cpu."""
    import innfer_amd.lib as L
    N, Cc, Hs, Ws = x.shape
    H, W = (2 * Hs, 2 * Ws) if up else (Hs, Ws)
    xin, g_in, lo_in = _split_slab(dev, x, (Cc + in_extra) // 32, Hs, Ws)
    packed = np.zeros(3 * L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
    wc = np.ascontiguousarray(w.numpy())
    L.check(L.lib.innfer_pack_conv3x3_split(wc.ctypes.data, K, Cc, packed.ctypes.data))
    d_packed, d_bias = torch.from_numpy(packed).to(dev), b.float().to(dev)
    og = (out_channels or max(K, 32)) // 32
    g_out = N * H * W * 32
    out = torch.full((2, og, N, H, W, 32), -3.0, dtype=torch.float16, device=dev)
    a = L.ConvArgs()
    a.d_in, a.in_group_stride, a.C = xin.data_ptr(), g_in, Cc
    a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
    a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g_out, out_off, K
    a.N, a.H, a.W, a.act, a.upsample2x = N, H, W, act, int(up)
    a.split, a.in_lo, a.out_lo = 1, lo_in, og * g_out
    keep = [xin, d_packed, d_bias]
    for name, r, sc in (("1", res1, s1), ("2", res2, s2)):
        if r is not None:
            rs, g_r, lo_r = _split_slab(dev, r, max(K, 32) // 32, H, W)
            setattr(a, f"d_res{name}", rs.data_ptr()); setattr(a, f"res{name}_group_stride", g_r); setattr(a, f"res{name}_scale", sc)
            setattr(a, f"res{name}_lo", lo_r)
            keep.append(rs)
    if rows:
        a.row_begin, a.row_end = rows
    L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
    torch.cuda.synchronize()
    res = torch.empty((N, K, H, W), dtype=torch.float32, device=dev)
    L.check(L.lib.innfer_slab_split_to_nchw(out.data_ptr(), g_out, og * g_out, out_off, res.data_ptr(), N, K, H, W, None))
    torch.cuda.synchronize()
    return res.cpu()


def _ref64(x, w, b, act=0, up=False, res1=None, s1=1.0, res2=None, s2=1.0):
    xx = x.double()
    if up:
        xx = F.interpolate(xx, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xx, w.double(), b.double(), padding=1)
    if act == 1:
        y = F.leaky_relu(y, 0.2)
    elif act == 2:
        y = F.relu(y)
    if res1 is not None:
        y = y * s1 + res1.double()
    if res2 is not None:
        y = y * s2 + res2.double()
    return y


def test_split_slab_round_trip_keeps_22_bits(dev):
    """fp32 -> (hi, lo) slabs -> fp32: relative error <= 2^-22 (values in fp16's normal range), exact zeros, fp16-representable values exact."""
    import innfer_amd.lib as L
    from innfer_amd import synth
    x = torch.from_numpy(synth.uniform((2, 40, 9, 11), 7, -4, 4))
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, -0.5, 1.0009765625])
    buf, g, lo = _split_slab(dev, x, 2, 9, 11)
    back = torch.empty_like(x, device=dev)
    L.check(L.lib.innfer_slab_split_to_nchw(buf.data_ptr(), g, lo, 0, back.data_ptr(), 2, 40, 9, 11, None))
    torch.cuda.synchronize()
    back = back.cpu()
    assert torch.equal(back[0, 0, 0, :4], x[0, 0, 0, :4])
    rel = ((back - x).abs() / x.abs().clamp_min(1e-3)).max().item()
    assert rel <= 2.0 ** -22, rel


@pytest.mark.parametrize("Cc,K,H,W,N,act", [
    (64, 32, 16, 32, 1, 1), (96, 32, 37, 45, 2, 1), (160, 32, 33, 33, 1, 2), (192, 64, 21, 50, 2, 0), (64, 64, 24, 40, 3, 1),
    (32, 32, 5, 3, 1, 0), (64, 64, 1, 1, 1, 1), (128, 32, 50, 70, 5, 1),
])
def test_split_conv_vs_float64_conv2d(dev, Cc, K, H, W, N, act):
    """The split conv (hi/lo operands, three virtual chunk passes, one accumulator set) against F.conv2d in float64 on the same fp32 operands.
    N > 1 takes the image-canvas form where that saves tiles."""
    from innfer_amd import synth
    x = torch.from_numpy(synth.uniform((N, Cc, H, W), 1, -1, 1))
    w = torch.from_numpy(synth.uniform((K, Cc, 3, 3), 2, -1, 1)) / np.sqrt(9 * Cc)
    b = torch.from_numpy(synth.uniform((K,), 3, -1, 1))
    got = _run_split_conv(dev, x, w, b, K, act=act, in_extra=32)
    ref = _ref64(x, w, b, act=act)
    err = (got.double() - ref).abs().max().item()
    print(f"split conv {Cc}->{K} {N}x{H}x{W}: max|err| {err:.2e}")
    assert err < 3e-6, err


def test_split_conv_epilogues_upsample_rows(dev):
    """Residual epilogues (x5 * 0.2 + x, then * 0.2 + x: RRDBNet_arch.py:98,165), the nearest-2x read (block.py:358), a row range and an output
    channel offset in split mode."""
    from innfer_amd import synth
    x = torch.from_numpy(synth.uniform((2, 192, 30, 41), 11, -1, 1))
    w = torch.from_numpy(synth.uniform((64, 192, 3, 3), 12, -1, 1)) / np.sqrt(9 * 192)
    b = torch.from_numpy(synth.uniform((64,), 13, -1, 1))
    r1 = torch.from_numpy(synth.uniform((2, 64, 30, 41), 14, -2, 2))
    r2 = torch.from_numpy(synth.uniform((2, 64, 30, 41), 15, -2, 2))
    got = _run_split_conv(dev, x, w, b, 64, res1=r1, s1=0.2, res2=r2, s2=0.2)
    assert (got.double() - _ref64(x, w, b, res1=r1, s1=0.2, res2=r2, s2=0.2)).abs().max().item() < 3e-6
    got = _run_split_conv(dev, x, w, b, 64, act=1, res1=r1, s1=1.0)
    assert (got.double() - _ref64(x, w, b, act=1, res1=r1, s1=1.0)).abs().max().item() < 3e-6
    xs = torch.from_numpy(synth.uniform((1, 64, 17, 23), 16, -1, 1))
    ws = torch.from_numpy(synth.uniform((64, 64, 3, 3), 17, -1, 1)) / np.sqrt(9 * 64)
    got = _run_split_conv(dev, xs, ws, b, 64, act=1, up=True)
    assert (got.double() - _ref64(xs, ws, b, act=1, up=True)).abs().max().item() < 3e-6
    w32 = torch.from_numpy(synth.uniform((32, 64, 3, 3), 18, -1, 1)) / np.sqrt(9 * 64)
    full = _ref64(xs, w32, b[:32], act=1)
    got = _run_split_conv(dev, xs, w32, b[:32], 32, act=1, rows=(5, 12), out_off=32, out_channels=64)
    assert (got[:, :, 5:12].double() - full[:, :, 5:12]).abs().max().item() < 3e-6


def _rrdb(dev, nb, scale, seed=0, **kw):
    from innfer_amd import synth
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    net = RRDBNet(3, 3, 64, nb, upscale=scale, **kw)
    net.load_state_dict(_sd(synth.rrdbnet_shapes(nb=nb, scale=scale, **({"plus": True} if kw.get("plus") else {})), seed), strict=True)
    return net.to(dev).eval()


def test_rrdbnet_fp32_mode_vs_golden_g3(dev, golden):
    """RRDBNet-23 4x on a float32 tensor = the fp32-accurate engine: vs golden G3 (the reference's fp32 forward) <= 1e-4; the same
    module on x.half() still runs the fp16 engine (error of the fp16 size, <= 1e-2); the fp32 result does not depend on which mode ran before; a
    batch equals its batch-1 forwards bit for bit."""
    from innfer_amd import synth
    g3 = golden("g3_rrdbnet23_x4")
    net = _rrdb(dev, 23, 4)
    for tag, shape, seed in (("out_32", (1, 3, 32, 32), 3), ("out_16", (1, 3, 16, 16), 4)):
        x = torch.from_numpy(synth.uniform(shape, seed)).to(dev)
        y32 = net(x)
        assert y32.dtype == torch.float32
        err = np.abs(y32.cpu().numpy() - g3[tag])
        print(f"RRDBNet-23 fp32 mode vs G3 {tag}: max {err.max():.2e} mean {err.mean():.2e}")
        assert err.max() < FP32_TOL, (tag, err.max())
        y16 = net(x.half())
        e16 = np.abs(y16.float().cpu().numpy() - g3[tag]).max()
        assert y16.dtype == torch.float16 and 10 * err.max() < e16 < 1e-2, (e16, err.max())
        assert torch.equal(net(x), y32)
    xb = torch.from_numpy(synth.uniform((1, 3, 16, 16), 77)).to(dev)
    y2 = net(torch.cat([x, xb], 0))
    assert torch.equal(y2[0:1], y32) and torch.equal(y2[1:2], net(xb))


def test_fp32_mode_scales_plus_srresnet(dev, golden):
    """fp32-accurate forwards of the other net.hip graphs against their fp32 goldens: RRDBNet 1x / 2x / 3x / 8x, finalact and ESRGAN+ (G5), SRResNet
    4x with PixelShuffle (G6: the shuffle runs as a gather over both slabs of a pair), <= 1e-4."""
    from innfer_amd import synth
    from innfer_amd.architectures.SRResNet_arch import SRResNet
    g = golden("g5_scales")
    x = torch.from_numpy(synth.uniform((1, 3, 16, 16), 5)).to(dev)
    for scale in (1, 2, 3, 8):
        y = _rrdb(dev, 1, scale)(x).cpu().numpy()
        err = np.abs(y - g[f"out_x{scale}"]).max()
        print(f"RRDBNet {scale}x fp32 mode: max err {err:.2e}")
        assert y.shape == g[f"out_x{scale}"].shape and err < FP32_TOL, (scale, err)
    for fa in ("tanh", "sigmoid"):
        assert np.abs(_rrdb(dev, 1, 4, finalact=fa)(x).cpu().numpy() - g[f"out_x4_{fa}"]).max() < FP32_TOL, fa
    err = np.abs(_rrdb(dev, 1, 4, plus=True)(x).cpu().numpy() - g["out_x4_plus"]).max()
    print(f"ESRGAN+ fp32 mode: max err {err:.2e}")
    assert err < FP32_TOL, err
    g6 = golden("g6_srgan")
    net = SRResNet(3, 3, 64, 16, upscale=4, norm_type=None, act_type='relu', mode='CNA', upsample_mode='pixelshuffle')
    net.load_state_dict(_sd(synth.srresnet_shapes(nb=16, scale=4)), strict=True)
    y = net.to(dev).eval()(torch.from_numpy(synth.uniform((1, 3, 24, 24), 6)).to(dev)).cpu().numpy()
    err = np.abs(y - g6["out_24"]).max()
    print(f"SRResNet 4x fp32 mode: max err {err:.2e}")
    assert err < FP32_TOL, err


def test_model_chop_fp32_mode_vs_golden_g4(dev, golden, tmp_path):
    """Model.__call__ (chop and un-tiled) on float32 tensors vs golden G4 (the reference's Model in fp32): <= 1e-4 -- tile gather, canvas batches
    through the fp32-accurate engine, fp32 blend (bit-exact by itself)."""
    from innfer_amd import synth
    from innfer_amd.run import Model
    g = golden("g4_chop")
    for (nb, scale, h, w, tag) in [(2, 4, 250, 330, "x4_250x330"), (1, 2, 201, 640, "x2_201x640"), (1, 1, 150, 250, "x1_150x250")]:
        path = str(tmp_path / f"{scale}x_{tag}.pth")
        torch.save(_sd(synth.rrdbnet_shapes(nb=nb, scale=scale)), path)
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 40 + scale)).to(dev)
        y = Model(path, arch="infer", scale=None, device="cuda", chop=True, tile_batch=4)(x)
        assert y.dtype == torch.float32 and tuple(y.shape) == (1, 3, h * scale, w * scale)
        y = y.cpu()
        e1 = np.abs(y[0, :, ::8, ::8].numpy() - g[f"chop_{tag}_sub"]).max()
        e2 = np.abs(y[0, :, -32:, -32:].numpy() - g[f"chop_{tag}_crop_b"]).max()
        y2 = Model(path, arch="infer", device="cuda", chop=False)(x).cpu()
        e3 = np.abs(y2[0, :, ::8, ::8].numpy() - g[f"nochop_{tag}_sub"]).max()
        print(f"Model fp32 mode {tag}: chop {e1:.2e} / {e2:.2e}, un-tiled {e3:.2e}")
        assert max(e1, e2, e3) < FP32_TOL, (tag, e1, e2, e3)


def test_unet_fp32_mode_vs_goldens(dev, golden):
    """pix2pix UnetGenerator on float32 tensors = the reference's -no_fp16 mode (run.py:345,421-422): innfer_unet_set_precision(1), every conv / norm / activation in
    fp32 (csrc/f32ops.hip).  Golden G7 (train-mode BatchNorm, UNet_256), G17 (eval mode on running statistics), G23 (instance norm, dropout under eval, upconv):
    <= 1e-4 on the tanh output; a batch equals its images' own forwards; the fp16 engine still answers float16 tensors from the same module."""
    import ast
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.architectures.UNet_arch import UnetGenerator
    from innfer_amd.utils.defaults import get_network_G_config
    from test_oracle_golden import G23_CASES, _g23_state
    g, g17 = golden("g7_unet256"), golden("g17_fp16_and_eval")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    net = get_network(get_network_G_config("p2p_256", 1))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}, strict=True)
    net = net.to(dev).train()
    xa = torch.from_numpy(synth.uniform((1, 3, 256, 256), 7, -1.0, 1.0)).to(dev)
    xb = torch.from_numpy(synth.uniform((1, 3, 256, 256), 8, -1.0, 1.0)).to(dev)
    ya = net(xa)
    assert ya.dtype == torch.float32
    e = ya.cpu().numpy()[0, :, ::4, ::4] - g["out_a_sub"]        # (the fixture's fp32 part: every fourth pixel; `out_a` is stored as fp16)
    print(f"unet256 fp32 mode, train-mode BN vs G7: max {np.abs(e).max():.2e} mean {np.abs(e).mean():.2e}")
    assert np.abs(e).max() < FP32_TOL, np.abs(e).max()
    assert np.abs(ya.cpu().numpy() - g["out_a"].astype(np.float32)).max() < 2.0 ** -11           # ... and the whole image to the fp16 fixture's own rounding
    yab = net(torch.cat([xa, xb], 0))
    assert torch.equal(yab[0:1], ya) and torch.equal(yab[1:2], net(xb))
    y16 = net(xa.half())                                   # the dtype selects the engine per call
    assert y16.dtype == torch.float16 and (y16.float() - ya).abs().max().item() < 1e-2
    assert torch.equal(net(xa), ya)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_running_stats(synth.fill_state_dict(shapes, 0), 17).items()}, strict=True)
    net.eval()
    e = net(xa).cpu().numpy()[0, :, ::4, ::4] - g17["unet_eval_out_a_sub"]
    print(f"unet256 fp32 mode, eval-mode BN vs G17: max {np.abs(e).max():.2e}")
    assert np.abs(e).max() < FP32_TOL, np.abs(e).max()
    g23 = golden("g23_unet_variants")
    for i, (tag, kw, ev) in enumerate(G23_CASES):
        v = UnetGenerator(3, 3, 5, ngf=32, **kw)
        v.load_state_dict(_g23_state(g23, tag, i), strict=True)
        v = v.to(dev)
        v = v.eval() if ev else v.train()
        x = torch.from_numpy(synth.uniform((1, 3, 64, 96), 240 + i, -1.0, 1.0)).to(dev)
        e = np.abs(v(x).cpu().numpy() - g23[tag]).max()
        print(f"unet variant {tag} fp32 mode vs G23: max {e:.2e}")
        assert e < FP32_TOL, (tag, e)


def test_pan_fp32_mode_vs_goldens(dev, golden):
    """PAN on float32 tensors = the reference's -no_fp16 mode: innfer_pan_set_precision(1), PAN.forward in fp32 (csrc/f32ops.hip; FSA attention on the fp32 kernel).
    Golden G8 (PAN 4x, 16 SCPA blocks, self attention), G18 (self_attention=False, double_scpa=True, 2x) and the oracle at scales 1 / 2 / 3 with ragged sizes,
    grey input, bilinear up-blocks: <= 1e-4; batches equal their images' own forwards."""
    import ast
    import oracle
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.architectures.PAN_arch import PAN
    from innfer_amd.utils.defaults import get_network_G_config
    from test_oracle_golden import G18_PAN
    g = golden("g8_pan")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    net = get_network(get_network_G_config("pan", 4))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}, strict=True)
    net = net.to(dev).eval()
    for (h, w, seed) in [(48, 48, 8), (50, 70, 9)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed)).to(dev)
        y = net(x)
        assert y.dtype == torch.float32
        e = np.abs(y.cpu().numpy() - g[f"out_{h}x{w}"].astype(np.float32))
        print(f"PAN fp32 mode {h}x{w} vs G8: max {e.max():.2e} mean {e.mean():.2e}")
        assert e.max() < FP32_TOL * max(1.0, np.abs(g[f'out_{h}x{w}']).max()), e.max()
        y16 = net(x.half())
        assert y16.dtype == torch.float16 and (y16.float() - y).abs().max().item() < 1e-2
    xa = torch.from_numpy(synth.uniform((1, 3, 48, 48), 8)).to(dev)
    xb = torch.from_numpy(synth.uniform((1, 3, 48, 48), 21)).to(dev)
    yab = net(torch.cat([xa, xb], 0))
    assert torch.equal(yab[0:1], net(xa)) and torch.equal(yab[1:2], net(xb))
    g18 = golden("g18_pan_variants")
    for i, (tag, kw) in enumerate(G18_PAN.items()):
        shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g18[tag + "_keys"], g18[tag + "_shapes"])}
        v = PAN(3, 3, 40, 24, 3, **kw)
        v.load_state_dict({k: torch.from_numpy(a) for k, a in synth.fill_state_dict(shapes, 185 + i).items()}, strict=True)
        v = v.to(dev).eval()
        x = torch.from_numpy(synth.uniform((1, 3, 20, 28), 195 + i)).to(dev)
        e = np.abs(v(x).cpu().numpy() - g18[tag].astype(np.float32)).max()
        print(f"PAN variant {tag} fp32 mode vs G18: max {e:.2e}")
        assert e < FP32_TOL * max(1.0, np.abs(g18[tag]).max()), (tag, e)
    # (100 x 133 / 96 x 128: many 8 x 32 tiles of the LDS-tiled conv with ragged right / bottom tiles -- rows that are not 16-byte aligned take its one-pixel-per-lane
    #  epilogue, 96 x 128 the four-pixel one on every layer)
    for scale, in_nc, h, w, mode in [(2, 3, 37, 21, "nearest"), (1, 1, 24, 33, "nearest"), (3, 3, 17, 23, "nearest"), (4, 3, 20, 28, "bilinear"), (4, 3, 100, 133, "nearest"), (2, 3, 96, 128, "nearest")]:
        cfg = get_network_G_config({"type": "pan", "nb": 3, "in_nc": in_nc, "out_nc": in_nc}, scale)
        v = PAN(in_nc, in_nc, 40, 24, 3, scale=scale, ups_inter_mode=mode) if mode != "nearest" else get_network(cfg)
        sd = {k: torch.from_numpy(a) for k, a in synth.fill_state_dict({k: tuple(t.shape) for k, t in v.state_dict().items()}, 50 + scale).items()}
        v.load_state_dict(sd, strict=True)
        v = v.to(dev).eval()
        x = torch.from_numpy(synth.uniform((2, in_nc, h, w), 30 + scale))
        with torch.no_grad():
            ref = oracle.pan_forward(sd, x, nb=3, scale=scale, ups_inter_mode=mode)
        e = (v(x.to(dev)).cpu() - ref).abs().max().item()
        print(f"PAN x{scale} {mode} fp32 mode vs oracle: max {e:.2e}")
        assert e < FP32_TOL * max(1.0, ref.abs().max().item()), (scale, mode, e)


def test_pan_fp32_split_forms_vs_generic_and_oracle(dev):
    """The split-operand forms of PAN's fp32 mode (round 6: csrc/pan_scpa_split.hip -- an SCPA block as one launch on (hi, lo) fp16 pairs, 8 x 32 tiles --, the HR side on
    conv3x3_pc SPLIT with the PA block as the up-conv's epilogue, the attention on pan_attention_mfma<true>) against the generic fp32 kernels they replace
    (innfer_pan_set_fused_scpa(pan, 0)) and against the oracle: frames of exactly one tile, one pixel more than a tile in both directions, several tile rows / XCD runs,
    border-only frames, batches (an image of a batch == its own forward, bit for bit).  <= 1e-4 of the oracle, <= 1e-5 between the two engines."""
    import oracle
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    net = get_network(get_network_G_config("pan", 4))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    try:
        for shape in [(1, 3, 8, 32), (1, 3, 9, 33), (2, 3, 37, 45), (1, 3, 4, 5), (3, 3, 16, 64), (1, 3, 70, 130)]:
            x = torch.from_numpy(synth.uniform(shape, 60 + shape[2], 0, 1))
            with torch.no_grad():
                ref = oracle.pan_forward(sd, x, nb=16, scale=4)
            y = {}
            for mode in (1, 0, 5):
                net.fused_scpa = mode
                y[mode] = net(x.to(dev))
                assert y[mode].dtype == torch.float32
            lim = max(1.0, ref.abs().max().item())
            e1, e0 = (y[1].cpu() - ref).abs().max().item(), (y[0].cpu() - ref).abs().max().item()
            d10, d15 = (y[1] - y[0]).abs().max().item(), (y[1] - y[5]).abs().max().item()
            print(f"PAN fp32 {shape}: split forms vs oracle {e1:.2e}, generic vs oracle {e0:.2e}, split vs generic {d10:.2e}, PA epilogue vs PA launch {d15:.2e}")
            assert e1 < FP32_TOL * lim and e0 < FP32_TOL * lim, (shape, e1, e0)
            assert d10 < 1e-5 * lim and d15 < 1e-5 * lim, (shape, d10, d15)
            net.fused_scpa = 1
            if shape[0] > 1:
                xs = x.to(dev)
                for i in range(shape[0]):
                    assert torch.equal(y[1][i:i + 1], net(xs[i:i + 1])), (shape, i)
        # a frame of many tiles per workgroup (270 x 480: 34 x 15 tiles of 8 x 32 on 256 workgroups), engines against each other
        x = torch.from_numpy(synth.uniform((1, 3, 270, 480), 77, 0, 1)).to(dev)
        net.fused_scpa = 1
        a = net(x)
        net.fused_scpa = 0
        b = net(x)
        d = (a - b).abs().max().item()
        print(f"PAN fp32 270x480: split forms vs generic {d:.2e}")
        assert d < 1e-5 * max(1.0, b.abs().max().item()), d
    finally:
        net.fused_scpa = 1


def test_ppon_fp32_mode_vs_golden(dev, golden):
    """PPON on float32 tensors (innfer_ppon_set_precision(1): the eight dilated convs as tap tables of the generic fp32 conv, running sums in fp32) against golden
    G13 (all three outputs) and the oracle at scales 2 / 3 / 8: <= 1e-4 of the output range."""
    import ast
    import oracle
    from innfer_amd import synth
    from innfer_amd.architectures.PPON_arch import PPON
    g = golden("g13_ppon")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    nb = max(int(k.split(".")[3]) for k in shapes if k.startswith("CFEM.1.sub."))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}
    net = PPON(3, 64, nb, 3, upscale=4)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    for (h, w, seed) in [(24, 24, 13), (20, 28, 14)]:                 # the golden's own inputs
        outs = net(torch.from_numpy(synth.uniform((1, 3, h, w), seed)).to(dev))
        for name, y in zip("csp", outs):
            ref = g[f"out_{name}_{h}x{w}"].astype(np.float32)
            e = np.abs(y.cpu().numpy() - ref).max()
            print(f"PPON fp32 mode out_{name} {h}x{w} vs G13: max {e:.2e} (range {np.abs(ref).max():.2f})")
            assert e < FP32_TOL * max(1.0, np.abs(ref).max()), (name, h, w, e)
    for scale in (4, 2, 3, 8):
        net = PPON(3, 64, nb, 3, upscale=scale)
        sdn = sd if scale == 4 else {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(t.shape) for k, t in net.state_dict().items()}, 60 + scale).items()}
        net.load_state_dict(sdn, strict=True)
        net = net.to(dev).eval()
        x = torch.from_numpy(synth.uniform((2, 3, 12, 10) if scale == 8 else (2, 3, 24, 20), 70 + scale))
        with torch.no_grad():
            ref = oracle.ppon_forward(sdn, x, nb=nb, scale=scale)
        ys = net(x.to(dev))
        for name, y, r in zip(("out_c", "out_s", "out_p"), ys, ref):
            e = (y.cpu() - r).abs().max().item()
            print(f"PPON x{scale} fp32 mode {name} vs oracle: max {e:.2e} (range {r.abs().max().item():.2f})")
            assert y.dtype == torch.float32 and e < FP32_TOL * max(1.0, r.abs().max().item()), (scale, name, e)
        assert torch.equal(net(x[1:2].to(dev))[2], ys[2][1:2])
        y16 = net(x.to(dev).half())[2]
        assert y16.dtype == torch.float16 and (y16.float() - ys[2]).abs().max().item() < 1e-2 * max(1.0, ref[2].abs().max().item())


def test_cyclegan_resnet_fp32_mode_vs_goldens(dev, golden):
    """CycleGAN ResnetGenerator on float32 tensors (innfer_resnet_set_precision(1)): reflection / replication / zero padding as the generic fp32 conv's padding modes, 7x7
    convs as 49-entry tap tables, ConvTranspose2d(3, 2, 1, 1) as four phase launches, instance / batch norm in fp32.  Goldens G14 (9 blocks) and G22 (paddings, dropout
    under eval, upconv, norm_type 'batch' in both modes): <= 1e-4 on the tanh output."""
    import ast
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.architectures.ResNet_arch import ResnetGenerator
    from innfer_amd.utils.defaults import get_network_G_config
    from test_oracle_golden import G22_CASES, _g22_state
    g = golden("g14_resnet9")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    net = get_network(get_network_G_config("resnet_9blocks", 1))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}, strict=True)
    net = net.to(dev).eval()
    for (h, w, seed) in [(32, 40, 15), (64, 64, 16)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0)).to(dev)
        y = net(x)
        e = np.abs(y.cpu().numpy() - g[f"out_{h}x{w}"].astype(np.float32)).max()
        print(f"resnet9 fp32 mode {h}x{w} vs G14: max {e:.2e}")
        assert y.dtype == torch.float32 and e < FP32_TOL, (h, w, e)
        assert (net(x.half()).float() - y).abs().max().item() < 1e-2
    xa = torch.from_numpy(synth.uniform((2, 3, 32, 40), 15, -1.0, 1.0)).to(dev)
    yab = net(xa)
    assert torch.equal(yab[0:1], net(xa[0:1])) and torch.equal(yab[1:2], net(xa[1:2]))
    g22 = golden("g22_resnet_variants")
    for i, (tag, kw) in enumerate(G22_CASES.items()):
        shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g22[tag + "_keys"], g22[tag + "_shapes"])}
        kw = dict(kw)
        train = kw.pop("train", False)
        v = ResnetGenerator(3, 3, 64, n_blocks=2, **{"norm_type": "instance", **kw})
        v.load_state_dict(_g22_state(shapes, kw, i), strict=True)
        v = v.to(dev)
        v = v.train() if train else v.eval()
        x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 225 + i, -1.0, 1.0)).to(dev)
        e = np.abs(v(x).cpu().numpy() - g22[tag]).max()
        print(f"resnet variant {tag} fp32 mode vs G22: max {e:.2e}")
        assert e < FP32_TOL, (tag, e)


def test_wbcunet_fp32_mode_vs_golden(dev, golden):
    """White-box-Cartoonization UNet on float32 tensors (innfer_wbc_set_precision(1)), 'pt' and 'tf' variants, and the run.py sequence network -> guided filter in fp32
    against golden G15: <= 1e-4."""
    import ast
    from innfer_amd import synth
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    from innfer_amd.utils import utils as U
    g = golden("g15_wbcunet")
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}
    net = get_network(get_network_G_config("wbcunet", 1))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    for (h, w, seed) in [(32, 40, 17), (64, 64, 18)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0)).to(dev)
        y = net(x)
        e = np.abs(y.cpu().numpy() - g[f"out_{h}x{w}"].astype(np.float32)).max()
        gf = U.guided_filter(x, y, r=1, eps=5e-3).cpu().numpy()
        eg = np.abs(gf - g[f"gf_{h}x{w}"].astype(np.float32)).max()
        print(f"wbcunet fp32 mode {h}x{w} vs G15: network max {e:.2e}, + guided filter {eg:.2e}")
        assert y.dtype == torch.float32 and e < FP32_TOL and eg < FP32_TOL, (h, w, e, eg)
    xa = torch.from_numpy(synth.uniform((2, 3, 32, 40), 17, -1.0, 1.0)).to(dev)
    yab = net(xa)
    assert torch.equal(yab[0:1], net(xa[0:1])) and torch.equal(yab[1:2], net(xa[1:2]))
    net_tf = get_network(get_network_G_config("wbcunet_tf", 1))
    net_tf.load_state_dict(sd, strict=True)
    net_tf = net_tf.to(dev).eval()
    x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 17, -1.0, 1.0)).to(dev)
    e = np.abs(net_tf(x).cpu().numpy() - g["out_tf_32x40"]).max()
    print(f"wbcunet_tf fp32 mode vs G15: max {e:.2e}")
    assert e < FP32_TOL, e


def test_every_generator_answers_float32_tensors(dev):
    """Since round 4 every generator has an fp32 mode (RRDBNet / SRResNet: the fp32-accurate engine on (hi, lo) fp16 pairs; PAN, UNet, PPON, CycleGAN ResNet, WBC UNet: fp32
    tensors on the fp32 matrix instruction): a float32 tensor returns float32 -- never fp16 accuracy behind fp32 I/O, never a refusal (the reference's -no_fp16 runs all)."""
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    for arch, scale, shape in (("esrgan", 4, (1, 3, 16, 16)), ("srgan", 4, (1, 3, 16, 16)), ("pan", 4, (1, 3, 16, 16)), ("unet_256", 1, (1, 3, 256, 256)),
                               ("ppon", 4, (1, 3, 16, 16)), ("resnet_9blocks", 1, (1, 3, 32, 32)), ("wbcunet", 1, (1, 3, 32, 32))):
        net = get_network(get_network_G_config(arch, scale)).to(dev)
        y = net(torch.zeros(shape, device=dev))
        y = y[-1] if isinstance(y, tuple) else y
        assert y.dtype == torch.float32 and torch.isfinite(y).all(), arch


def test_split_conv_random_shapes_fuzz(dev):
    """Seeded sweep over what the split kernels are parameterised on -- channel counts, image sizes around the tile sizes (24 x 32, 16 x 32), batches
    (N > 1 takes the image-canvas form when that saves tiles: cells, gutters, tiles straddling images), activation, one or two residuals, nearest-2x
    input -- against float64 F.conv2d on the same fp32 operands; bound 3e-6 of O(1) outputs (measured <= 1.8e-6)."""
    from innfer_amd import synth
    rng = np.random.RandomState(20260303)
    worst = 0.0
    for case in range(40):
        K = int(rng.choice([32, 64]))
        Cc = int(rng.choice([32, 64, 96, 128, 160, 192]))
        N = int(rng.choice([1, 1, 2, 3, 5]))
        H, W = int(rng.randint(1, 70)), int(rng.randint(1, 90))
        if case % 6 == 0:
            H, W = int(rng.choice([24, 48, 16, 32, 25, 47])), int(rng.choice([32, 64, 33, 63, 31]))      # whole tiles and one off
        act = int(rng.choice([0, 1, 2]))
        up = bool(rng.rand() < 0.15) and N == 1
        use_r1 = rng.rand() < 0.4
        use_r2 = use_r1 and rng.rand() < 0.5
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        x = torch.from_numpy(synth.uniform((N, Cc, H, W), 7000 + case, -1, 1))
        w = torch.from_numpy(synth.uniform((K, Cc, 3, 3), 7100 + case, -1, 1)) / np.sqrt(9 * Cc)
        b = torch.from_numpy(synth.uniform((K,), 7200 + case, -1, 1))
        r1 = torch.from_numpy(synth.uniform((N, K, Ho, Wo), 7300 + case, -1, 1)) if use_r1 else None
        r2 = torch.from_numpy(synth.uniform((N, K, Ho, Wo), 7400 + case, -1, 1)) if use_r2 else None
        got = _run_split_conv(dev, x, w, b, K, act=act, up=up, res1=r1, s1=0.2, res2=r2, s2=0.2)
        ref = _ref64(x, w, b, act=act, up=up, res1=r1, s1=0.2, res2=r2, s2=0.2)
        err = (got.double() - ref).abs().max().item()
        worst = max(worst, err)
        assert err < 3e-6, (case, Cc, K, N, H, W, act, up, use_r1, use_r2, err)
    print(f"split conv fuzz: worst max|err| {worst:.2e}")
