"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from oracle import tiles as otiles
from innfer_amd import synth


def _sd(shapes, seed=0):
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed).items()}


def test_g1_geometry(golden):
    g = golden("g1_geometry")
    for key in g.files:
        h, w = map(int, key[4:].split("x"))
        ps, ys, xs = oracle.chop_geometry(h, w)
        org = np.array([(y, x) for y in ys for x in xs], dtype=np.int32)
        assert np.array_equal(org, g[key]), key
    ps, ys, xs = oracle.chop_geometry(250, 330)
    assert [(y, x) for y in ys for x in xs] == [(0, 0), (0, 100), (0, 130), (50, 0), (50, 100), (50, 130)]
    for (h, w, n) in [(1080, 1920, 190), (2160, 3840, 798), (4320, 7680, 3268), (128, 128, 1)]:
        _, ys, xs = oracle.chop_geometry(h, w)
        assert len(ys) * len(xs) == n


def test_g2_blend(golden):
    g = golden("g2_blend")
    for key in g.files:
        if key.startswith("profile_"):
            P, s = key[8:].split("_")
            prof = oracle.blend_profile(int(P[1:]), 0.5, int(s[1:]))
            assert np.array_equal(prof.numpy(), g[key]), key
    for scale in (1, 2, 4):
        h, w = 250, 330
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 100 + scale))
        p = oracle.extract_patches_2d(x, (200, 200), [0.5, 0.5], batch_first=True).squeeze(0)
        up = torch.nn.functional.interpolate(p, scale_factor=float(scale), mode="nearest") if scale > 1 else p
        k = torch.arange(up.shape[0], dtype=torch.float32)[:, None, None, None]
        r = oracle.recompose_tensor(up * (1.0 + k / 16.0), h, w, step=0.5, scale=scale)
        assert np.array_equal(r[0, :, ::7, ::5].numpy(), g[f"blend_s{scale}_sub"])      # bit-exact
        assert r.double().sum().item() == g[f"blend_s{scale}_sum"]
        r_id = oracle.recompose_tensor(up, h, w, step=0.5, scale=scale)
        ref = torch.nn.functional.interpolate(x, scale_factor=float(scale), mode="nearest") if scale > 1 else x
        assert (r_id - ref).abs().max().item() <= 3e-7
    x = torch.from_numpy(synth.uniform((1, 3, 150, 250), 77))
    p = oracle.extract_patches_2d(x, (150, 150), [0.5, 0.5], batch_first=True).squeeze(0)
    k = torch.arange(p.shape[0], dtype=torch.float32)[:, None, None, None]
    r = oracle.recompose_tensor(p * (1.0 + k / 16.0), 150, 250, step=0.5, scale=1)
    assert np.array_equal(r[0].numpy(), g["blend_150x250"])


def test_g3_rrdbnet23(golden):
    g = golden("g3_rrdbnet23_x4")
    sd = _sd(synth.rrdbnet_shapes(nb=23, scale=4))
    with torch.no_grad():
        x16 = torch.from_numpy(synth.uniform((1, 3, 16, 16), 4))
        taps = {}
        y16 = oracle.rrdbnet_forward(sd, x16, nb=23, scale=4, taps=taps)
        np.testing.assert_allclose(y16.numpy(), g["out_16"], atol=2e-6, rtol=0)
        for k in ("conv_first", "rrdb0", "trunk", "up0", "up1"):
            np.testing.assert_allclose(taps[k].numpy(), g["tap16_" + k], atol=2e-6, rtol=0, err_msg=k)
        x = torch.from_numpy(synth.uniform((1, 3, 32, 32), 3))
        y = oracle.rrdbnet_forward(sd, x, nb=23, scale=4)
        np.testing.assert_allclose(y.numpy(), g["out_32"], atol=2e-6, rtol=0)
        rdb = oracle.rdb_forward(sd, "model.1.sub.0.RDB1.", taps["conv_first"])
        np.testing.assert_allclose(rdb.numpy(), g["tap16_rdb0"], atol=2e-6, rtol=0)


def test_g4_chop(golden):
    g = golden("g4_chop")
    for (nb, scale, h, w, tag) in [(2, 4, 250, 330, "x4_250x330"), (1, 2, 201, 640, "x2_201x640"),
                                   (1, 1, 150, 250, "x1_150x250")]:
        sd = _sd(synth.rrdbnet_shapes(nb=nb, scale=scale))
        x = torch.from_numpy(synth.uniform((1, 3, h, w), 40 + scale))
        with torch.no_grad():
            fn = lambda t: oracle.rrdbnet_forward(sd, t, nb=nb, scale=scale)
            y = oracle.chop_forward(fn, x, scale)
            y2 = fn(x)
        np.testing.assert_allclose(y[0, :, ::8, ::8].numpy(), g[f"chop_{tag}_sub"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(y[0, :, -32:, -32:].numpy(), g[f"chop_{tag}_crop_b"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(y2[0, :, ::8, ::8].numpy(), g[f"nochop_{tag}_sub"], atol=2e-6, rtol=0)
        assert abs(y.double().sum().item() - g[f"chop_{tag}_sum"]) < 1e-2 * max(1, abs(g[f"chop_{tag}_sum"]) * 1e-4)


def test_g5_scales(golden):
    g = golden("g5_scales")
    x = torch.from_numpy(synth.uniform((1, 3, 16, 16), 5))
    with torch.no_grad():
        for scale in (1, 2, 3, 8):
            sd = _sd(synth.rrdbnet_shapes(nb=1, scale=scale))
            y = oracle.rrdbnet_forward(sd, x, nb=1, scale=scale)
            np.testing.assert_allclose(y.numpy(), g[f"out_x{scale}"], atol=2e-6, rtol=0)
        sd = _sd(synth.rrdbnet_shapes(nb=1, scale=4, plus=True))
        y = oracle.rrdbnet_forward(sd, x, nb=1, scale=4, plus=True)
        np.testing.assert_allclose(y.numpy(), g["out_x4_plus"], atol=2e-6, rtol=0)
        sd = _sd(synth.rrdbnet_shapes(nb=1, scale=4))
        for fa in ("tanh", "sigmoid"):
            y = oracle.rrdbnet_forward(sd, x, nb=1, scale=4, finalact=fa)
            np.testing.assert_allclose(y.numpy(), g[f"out_x4_{fa}"], atol=2e-6, rtol=0)


G18_CASES = {"ps4": (dict(nb=2, scale=4, upsample_mode="pixelshuffle"), (1, 3, 12, 20), 190),
             "ps2_relu": (dict(nb=1, scale=2, upsample_mode="pixelshuffle", act_type="relu"), (1, 3, 16, 16), 191),
             "nr2_relu_nac": (dict(nb=2, nr=2, scale=2, act_type="relu"), (1, 3, 16, 16), 192),
             "nr4": (dict(nb=1, nr=4, scale=1), (1, 3, 10, 14), 193)}


def test_g18_rrdbnet_constructor_variants(golden):
    """nr != 3, act_type='relu', mode='NAC', upsample_mode='pixelshuffle' (RRDBNet_arch.py:16-48): key names and outputs of the reference."""
    g = golden("g18_rrdb_variants")
    for i, (tag, (kw, shape, xseed)) in enumerate(G18_CASES.items()):
        shapes = synth.rrdbnet_shapes(nb=kw["nb"], scale=kw["scale"], nr=kw.get("nr", 3), upsample_mode=kw.get("upsample_mode", "upconv"))
        assert sorted(shapes) == list(g[tag + "_keys"])
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 180 + i).items()}
        x = torch.from_numpy(synth.uniform(shape, xseed))
        with torch.no_grad():
            y = oracle.rrdbnet_forward(sd, x, **kw).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 2e-6, tag
    # norm_type='batch' under eval(): the BatchNorm2d layers behind the dense blocks' convs and behind LR_conv
    shapes = synth.rrdbnet_shapes(nb=2, scale=2, norm=True)
    assert sorted(shapes) == list(g["batchnorm_keys"])
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_running_stats(synth.fill_state_dict(shapes, 184), 184).items()}
    with torch.no_grad():
        y = oracle.rrdbnet_forward(sd, torch.from_numpy(synth.uniform((1, 3, 16, 16), 194)), nb=2, scale=2).numpy()
    assert np.abs(y - g["batchnorm"]).max() < 5e-6


G18_SR = {"sr_upconv_lrelu": dict(scale=2, act_type="leakyrelu", upsample_mode="upconv", res_scale=0.5),
          "sr_ps_scale": dict(scale=4, act_type="relu", upsample_mode="pixelshuffle", res_scale=0.25)}
OUTMS = (None, "scaltanh", "tanh", "sigmoid", "clamp")


def test_g18_srresnet_variants_and_outm(golden):
    """SRResNet(act_type, res_scale, upsample_mode='upconv') (SRResNet_arch.py:16-91) and the `outm` argument of SRResNet.forward /
    RRDBNet.forward (RRDBNet_arch.py:50-62) against the reference."""
    g = golden("g18_rrdb_variants")
    for j, (tag, kw) in enumerate(G18_SR.items()):
        shapes = synth.srresnet_shapes(nb=2, scale=kw["scale"], upsample_mode=kw["upsample_mode"])
        assert sorted(shapes) == list(g[tag + "_keys"])
        sd = _sd(shapes, 186 + j)
        x = torch.from_numpy(synth.uniform((1, 3, 14, 18), 196 + j))
        for om in OUTMS:
            with torch.no_grad():
                y = oracle.srresnet_forward(sd, x, nb=2, outm=om, **kw).numpy()
            assert np.abs(y - g[tag + ("_" + om if om else "")]).max() < 2e-6, (tag, om)
    sd = _sd(synth.rrdbnet_shapes(nb=1, scale=2))
    x = torch.from_numpy(synth.uniform((1, 3, 12, 12), 198))
    for om in ("scaltanh", "clamp"):
        with torch.no_grad():
            assert np.abs(oracle.rrdbnet_forward(sd, x, nb=1, scale=2, outm=om).numpy() - g["rrdb_" + om]).max() < 2e-6


G26_SR = {"nac_bn": dict(), "cna_bn": dict(mode="CNA"), "nac": dict(norm_type=None), "cnac_bn": dict(mode="CNAC"), "cnac": dict(norm_type=None, mode="CNAC"),
          "x3_lrelu": dict(norm_type=None, mode="CNA", upscale=3, act_type="leakyrelu", res_scale=0.5), "nac_bn_x2_lrelu": dict(upscale=2, act_type="leakyrelu", res_scale=0.25)}


def g26_case(j, tag):
    """(constructor kwargs, state dict as numpy, input) of golden G26's case `tag` -- as tests/golden/make_golden.py g26 builds them."""
    kw = G26_SR[tag]
    norm, mode, sc = bool(kw.get("norm_type", "batch")), kw.get("mode", "NAC"), kw.get("upscale", 4)
    shapes = synth.srresnet_shapes(nb=2, scale=sc, upsample_mode="upconv", norm=norm, mode=mode)
    sd = synth.fill_state_dict(shapes, 300 + j)
    if norm:
        sd = synth.fill_running_stats(sd, 300 + j)
    return dict(norm=norm, mode=mode, scale=sc, act_type=kw.get("act_type", "relu"), res_scale=kw.get("res_scale", 1)), shapes, sd, synth.uniform((1, 3, 14, 18), 310 + j)


def test_g26_srresnet_norm_and_mode(golden):
    """SRResNet(norm_type='batch' / None, mode='NAC' / 'CNA' / 'CNAC') in eval mode and upscale=3 (SRResNet_arch.py:16-27,68-91; block.py:242-254)
    against the reference; the parameter names in state-dict order too."""
    g = golden("g26_srresnet_modes")
    for j, tag in enumerate(G26_SR):
        c, shapes, sd, x = g26_case(j, tag)
        assert list(shapes) == list(g[tag + "_keys"]), tag
        tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        with torch.no_grad():
            y = oracle.srresnet_forward(tsd, torch.from_numpy(x), nb=2, scale=c["scale"], act_type=c["act_type"], res_scale=c["res_scale"],
                                        upsample_mode="upconv", norm_type="batch" if c["norm"] else None, mode=c["mode"]).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 2e-6, tag
    for j, mode in enumerate(("NAC", "CNAC")):              # RRDBNet(norm_type='batch', mode): LR_conv = norm, conv under 'NAC'
        shapes = synth.rrdbnet_shapes(nb=2, scale=2, norm=True, mode=mode)
        assert list(shapes) == list(g[f"rrdb_bn_{mode}_keys"])
        sd = synth.fill_running_stats(synth.fill_state_dict(shapes, 320 + j), 320 + j)
        tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        with torch.no_grad():
            y = oracle.rrdbnet_forward(tsd, torch.from_numpy(synth.uniform((1, 3, 16, 16), 330 + j)), nb=2, scale=2).numpy()
        assert np.abs(y - g["rrdb_bn_" + mode]).max() < 2e-6, mode
    for tag, (shapes, seed, fwd) in G26_PS.items():        # PixelShuffle(3), PixelShuffle(2) on 32 features
        assert list(shapes()) == list(g[tag + "_keys"]), tag
        with torch.no_grad():
            y = fwd(_sd(shapes(), seed), torch.from_numpy(synth.uniform((1, 3, 10, 12), seed + 10))).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 2e-6, tag


G26_PS = {"ps3_sr": (lambda: synth.srresnet_shapes(nb=2, scale=3, upsample_mode="pixelshuffle"), 360,
                     lambda sd, x: oracle.srresnet_forward(sd, x, nb=2, scale=3, upsample_mode="pixelshuffle")),
          "ps2_sr_nf32": (lambda: synth.srresnet_shapes(nf=32, nb=2, scale=4, upsample_mode="pixelshuffle"), 361,
                          lambda sd, x: oracle.srresnet_forward(sd, x, nb=2, scale=4, upsample_mode="pixelshuffle")),
          "ps3_rrdb": (lambda: synth.rrdbnet_shapes(nb=1, scale=3, upsample_mode="pixelshuffle"), 362,
                       lambda sd, x: oracle.rrdbnet_forward(sd, x, nb=1, scale=3, upsample_mode="pixelshuffle"))}


G18_PAN = {"noattn": dict(self_attention=False), "double": dict(double_scpa=True),
           "double_noattn_x2": dict(double_scpa=True, self_attention=False, scale=2),
           "bilinear": dict(ups_inter_mode="bilinear"), "bilinear_noattn_x2": dict(ups_inter_mode="bilinear", self_attention=False, scale=2),
           "x3": dict(scale=3), "bilinear_x3": dict(ups_inter_mode="bilinear", scale=3, self_attention=False)}


def test_g18_pan_constructor_variants(golden):
    """PAN(self_attention=False / double_scpa=True) (PAN_arch.py:115-141,193-203) against the reference."""
    import ast
    g = golden("g18_pan_variants")
    for i, (tag, kw) in enumerate(G18_PAN.items()):
        shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g[tag + "_keys"], g[tag + "_shapes"])}
        sd = _sd(shapes, 185 + i)
        x = torch.from_numpy(synth.uniform((1, 3, 20, 28), 195 + i))
        with torch.no_grad():
            y = oracle.pan_forward(sd, x, nb=3, **kw).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 5e-6, tag


def test_g6_srgan(golden):
    g = golden("g6_srgan")
    sd = _sd(synth.srresnet_shapes(nb=16, scale=4))
    x = torch.from_numpy(synth.uniform((1, 3, 24, 24), 6))
    with torch.no_grad():
        y = oracle.srresnet_forward(sd, x, nb=16, scale=4)
    np.testing.assert_allclose(y.numpy(), g["out_24"], atol=2e-6, rtol=0)
    ps = torch.nn.functional.pixel_shuffle(torch.from_numpy(g["ps_in"]), 2)
    assert np.array_equal(ps.numpy(), g["ps_out"])


def test_g7_unet(golden):
    g = golden("g7_unet256")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = _sd(shapes)
    x = torch.from_numpy(synth.uniform((1, 3, 256, 256), 7, -1.0, 1.0))
    with torch.no_grad():
        y = oracle.unet_forward(sd, x)
    np.testing.assert_allclose(y[0, :, ::4, ::4].numpy(), g["out_a_sub"], atol=2e-5, rtol=0)


def test_g17_unet_eval_mode(golden):
    """The oracle's eval-mode BatchNorm (running statistics) against the reference UNet_256 in eval mode (golden G17)."""
    g, g17 = golden("g7_unet256"), golden("g17_fp16_and_eval")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_running_stats(synth.fill_state_dict(shapes, 0), 17).items()}
    x = torch.from_numpy(synth.uniform((1, 3, 256, 256), 7, -1.0, 1.0))
    with torch.no_grad():
        y = oracle.unet_forward(sd, x, training=False)
    np.testing.assert_allclose(y[0, :, ::4, ::4].numpy(), g17["unet_eval_out_a_sub"], atol=2e-5, rtol=0)
    # and the fixture's own statement of what fp16 costs the reference on these networks stays what the tests quote
    assert 5e-3 < g17["unet_fp16_err_vs_fp32"][0] < 1e-2 and 3e-3 < g17["resnet_fp16_err_vs_fp32_32x40"][0] < 6e-3


def test_g8_pan(golden):
    g = golden("g8_pan")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = _sd(shapes)
    for (h, w, seed) in [(48, 48, 8), (50, 70, 9)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed))
        with torch.no_grad():
            y = oracle.pan_forward(sd, x, nb=16, scale=4)
        np.testing.assert_allclose(y.numpy(), g[f"out_{h}x{w}"], atol=3e-6, rtol=0)


def test_g13_ppon(golden):
    g = golden("g13_ppon")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = _sd(shapes)
    for (h, w, seed) in [(24, 24, 13), (20, 28, 14)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed))
        with torch.no_grad():
            oc, os_, op = oracle.ppon_forward(sd, x, nb=24, scale=4)
        for name, t in (("c", oc), ("s", os_), ("p", op)):
            np.testing.assert_allclose(t.numpy(), g[f"out_{name}_{h}x{w}"], atol=5e-6, rtol=0)


def test_g27_ppon_scales(golden):
    """PPON with upscale 8 / 3 / 2 (PPON_arch.py:16-63: log2 upconv stages, ONE Upsample(3) stage for 3) against the reference."""
    g = golden("g27_ppon_scales")
    import ast
    for j, sc in enumerate((8, 3, 2)):
        shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g[f"x{sc}_keys"], g[f"x{sc}_shapes"])}
        sd = _sd(shapes, 340 + j)
        with torch.no_grad():
            outs = oracle.ppon_forward(sd, torch.from_numpy(synth.uniform((1, 3, 10, 12), 350 + j)), nb=2, scale=sc)
        for name, t in zip("csp", outs):
            np.testing.assert_allclose(t.numpy(), g[f"x{sc}_{name}"], atol=5e-6, rtol=0)


def test_g14_resnet9(golden):
    g = golden("g14_resnet9")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = _sd(shapes)
    for (h, w, seed) in [(32, 40, 15), (64, 64, 16)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0))
        with torch.no_grad():
            y = oracle.resnet_forward(sd, x, n_blocks=9)
        np.testing.assert_allclose(y.numpy(), g[f"out_{h}x{w}"], atol=1e-5, rtol=0)


def test_g15_wbcunet_and_guided_filter(golden):
    from oracle.guided import guided_filter
    g = golden("g15_wbcunet")
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    sd = _sd(shapes)
    for (h, w, seed) in [(32, 40, 17), (64, 64, 18)]:
        x = torch.from_numpy(synth.uniform((1, 3, h, w), seed, -1.0, 1.0))
        with torch.no_grad():
            y = oracle.wbcunet_forward(sd, x)
            gf = guided_filter(x, y, eps=5e-3)
        np.testing.assert_allclose(y.numpy(), g[f"out_{h}x{w}"], atol=5e-6, rtol=0)
        np.testing.assert_allclose(gf.numpy(), g[f"gf_{h}x{w}"], atol=2e-5, rtol=0)
    x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 17, -1.0, 1.0))
    with torch.no_grad():
        np.testing.assert_allclose(oracle.wbcunet_forward(sd, x, mode="tf").numpy(), g["out_tf_32x40"], atol=5e-6, rtol=0)


def test_g9_convert(golden):
    g = golden("g9_convert")
    assert np.array_equal(oracle.np2tensor(g["ramp"]).numpy(), g["np2t"])
    assert np.array_equal(oracle.np2tensor(g["ramp"], normalize=True).numpy(), g["np2t_norm"])
    th = torch.from_numpy(g["t2np_in"])
    assert np.array_equal(oracle.tensor2np(th), g["t2np"])
    assert np.array_equal(oracle.tensor2np(th * 2 - 1, denormalize=True), g["t2np_denorm"])
    assert np.array_equal(oracle.tensor2np(torch.from_numpy(g["big_in"])), g["big_u8"])
    assert np.array_equal(oracle.srgb2linear(np.arange(256, dtype=np.uint8)), g["srgb2linear"])
    assert np.array_equal(oracle.linear2srgb(np.linspace(-0.1, 1.1, 1001, dtype=np.float32)), g["linear2srgb"])
    # round trip is exact for uint8
    img = synth.image_u8(33, 47, 3, 1)
    assert np.array_equal(oracle.tensor2np(oracle.np2tensor(img)), img)


def _convert_flag_cases(g, np2tensor, tensor2np, to_t=lambda a: torch.from_numpy(a)):
    """(name, result, expected) of every G19 case for one implementation of np2tensor / tensor2np."""
    th, big = to_t(g["t2np_in"]), to_t(g["big_in"])
    return [("np2t16", np2tensor(g["ramp16"]), g["np2t16"]), ("np2t16_norm", np2tensor(g["ramp16"], normalize=True), g["np2t16_norm"]),
            ("np2t4", np2tensor(g["ramp4"]), g["np2t4"]), ("np2t8_noflip", np2tensor(g["ramp8"], bgr2rgb=False), g["np2t8_noflip"]),
            ("np2t8_nobatch", np2tensor(g["ramp8"], add_batch=False), g["np2t8_nobatch"]),
            ("np2t8_norange", np2tensor(g["ramp8"], change_range=False), g["np2t8_norange"]),
            ("t2np16", tensor2np(th, data_range=65535, imtype=np.uint16), g["t2np16"]),
            ("t2np16_denorm", tensor2np(th * 2 - 1, denormalize=True, data_range=65535, imtype=np.uint16), g["t2np16_denorm"]),
            ("big_u16", tensor2np(big, data_range=65535, imtype=np.uint16), g["big_u16"]),
            ("big_noflip", tensor2np(big, rgb2bgr=False), g["big_noflip"]), ("big_3d", tensor2np(big[0]), g["big_3d"]),
            ("big_2d", tensor2np(big[0, 1]), g["big_2d"])]


def test_g19_convert_flags(golden):
    """uint16 images and the non-default flags of np2tensor / tensor2np (utils.py:22-33,164-248) against the reference: bit-exact."""
    for name, got, want in _convert_flag_cases(golden("g19_convert_flags"), oracle.np2tensor, oracle.tensor2np):
        got = got.numpy() if isinstance(got, torch.Tensor) else got
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), name


def _guided_cases():
    x = torch.from_numpy(synth.uniform((2, 3, 23, 31), 211))
    y = torch.from_numpy(synth.uniform((2, 3, 23, 31), 212))
    xh = torch.from_numpy(synth.uniform((2, 3, 46, 62), 213))
    return x, y, xh, {"r2": dict(ks=5, eps=5e-3), "ks7": dict(ks=7, eps=1e-2), "fast": dict(ks=3, eps=5e-3, hr=True), "fast_r2": dict(ks=5, eps=1e-2, hr=True)}


def test_g21_guided_filter_windows_and_fast_mode(golden):
    """guided_filter with 5x5 / 7x7 windows and in 'fast' mode (utils.py:548-626) against the reference."""
    from oracle.guided import guided_filter_ex
    g = golden("g21_guided")
    x, y, xh, cases = _guided_cases()
    for tag, kw in cases.items():
        got = guided_filter_ex(x, y, ks=kw["ks"], eps=kw["eps"], x_hr=xh if kw.get("hr") else None).numpy()
        assert got.shape == g[tag].shape and np.abs(got - g[tag]).max() < 2e-5, tag


G22_CASES = {"zero": dict(padding_type="zero"), "replicate": dict(padding_type="replicate"),
             "reflect_dropout": dict(padding_type="reflect", use_dropout=True), "zero_dropout": dict(padding_type="zero", use_dropout=True),
             "upconv": dict(upsample_mode="upconv"),
             "batch_eval": dict(norm_type="batch"), "batch_train": dict(norm_type="batch", train=True),
             "batch_zero_upconv_eval": dict(norm_type="batch", padding_type="zero", upsample_mode="upconv")}


def _g22_state(shapes, kw, i):
    sd = synth.fill_state_dict(shapes, 220 + i)
    if kw.get("norm_type") == "batch":
        sd = synth.fill_running_stats(sd, 228 + i)
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_g22_resnet_padding_and_dropout_variants(golden):
    """ResnetGenerator(padding_type=zero / replicate, use_dropout=True) in eval mode (ResNet_arch.py:104-146) against the reference."""
    import ast
    g = golden("g22_resnet_variants")
    for i, (tag, kw) in enumerate(G22_CASES.items()):
        shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g[tag + "_keys"], g[tag + "_shapes"])}
        x = torch.from_numpy(synth.uniform((1, 3, 32, 40), 225 + i, -1.0, 1.0))
        with torch.no_grad():
            y = oracle.resnet_forward(_g22_state(shapes, kw, i), x, n_blocks=2, training=kw.get("train", False),
                                      **{k: v for k, v in kw.items() if k not in ("upsample_mode", "train")}).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 2e-5, tag


G23_CASES = [("instance", dict(norm_type="instance"), False), ("instance_eval", dict(norm_type="instance"), True),
             ("batch_dropout_eval", dict(norm_type="batch", use_dropout=True), True),
             ("batch_upconv", dict(norm_type="batch", upsample_mode="upconv"), False),
             ("batch_upconv_eval", dict(norm_type="batch", upsample_mode="upconv"), True),
             ("instance_upconv", dict(norm_type="instance", upsample_mode="upconv"), False)]


def _g23_state(g, tag, i):
    import ast
    shapes = {str(k): ast.literal_eval(str(s)) for k, s in zip(g[tag + "_keys"], g[tag + "_shapes"])}
    sd = synth.fill_state_dict(shapes, 230 + i)
    if "batch" in tag:
        sd = synth.fill_running_stats(sd, 235 + i)
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_g23_unet_instance_norm_and_dropout_variants(golden):
    """UnetGenerator(norm_type='instance') (train / eval: the same) and (use_dropout=True) under eval() (UNet_arch.py:20-157) against the reference."""
    g = golden("g23_unet_variants")
    for i, (tag, kw, ev) in enumerate(G23_CASES):
        x = torch.from_numpy(synth.uniform((1, 3, 64, 96), 240 + i, -1.0, 1.0))
        with torch.no_grad():
            y = oracle.unet_forward(_g23_state(g, tag, i), x, num_downs=5, training=not ev, norm_type=kw["norm_type"],
                                    upsample_mode=kw.get("upsample_mode", "deconv")).numpy()
        assert y.shape == g[tag].shape and np.abs(y - g[tag]).max() < 2e-5, tag


def test_g16_mrrdbnet(golden):
    """MRRDBNet (new-arch ESRGAN built directly, RRDBNet_arch.py:173-231) on its own key names."""
    g = golden("g16_mrrdb")
    shapes = synth.mrrdbnet_shapes(nb=2)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 61).items()}
    x = torch.from_numpy(synth.uniform((2, 3, 16, 20), 62))
    with torch.no_grad():
        y = oracle.mrrdbnet_forward(sd, x, nb=2)
    np.testing.assert_allclose(y.numpy(), g["out"], atol=2e-6, rtol=0)


def test_chop_forward_window_is_the_full_chop_on_its_window():
    """oracle.chop_forward_window (the restricted chop_forward the full-size GPU tests check BASELINE configs 3 / 4 with) == oracle.chop_forward -- itself pinned to the
    reference's by G2 / G4 -- on the window, bit for bit: corners, a four-tile seam, a ragged last row / column, a window across many tiles; one and two stages."""
    import torch.nn.functional as F
    x = torch.from_numpy(synth.uniform((1, 3, 431, 615), 71))
    w = torch.from_numpy(synth.uniform((3, 3, 3, 3), 72, -0.3, 0.3))
    fn = lambda t: F.interpolate(F.conv2d(t, w, padding=1), scale_factor=2.0, mode="bilinear")
    crop = lambda a, b, c, d: x[:, :, a:b, c:d]
    full = oracle.chop_forward(fn, x, 2)
    for win in [(0, 64, 0, 64), (862 - 50, 862, 1230 - 70, 1230), (390, 470, 380, 460), (100, 700, 50, 60)]:
        r = oracle.chop_forward_window(fn, crop, 431, 615, 2, win)
        assert torch.equal(r, full[:, :, win[0]:win[1], win[2]:win[3]]), win
    f1 = lambda t: F.conv2d(t, w, padding=1)
    mid_full = oracle.chop_forward(f1, x, 1)
    two = oracle.chop_forward(fn, mid_full, 2)
    cache = {}
    mid = lambda a, b, c, d: oracle.chop_forward_window(f1, crop, 431, 615, 1, (a, b, c, d), cache=cache)
    win = (862 - 40, 862, 1230 - 40, 1230)
    assert torch.equal(oracle.chop_forward_window(fn, mid, 431, 615, 2, win), two[:, :, win[0]:win[1], win[2]:win[3]])
