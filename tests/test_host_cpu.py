"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host-side logic
(tile geometry, blend profile, weight panels, loader, default configs, tile sharding) matches the
golden vectors generated from the reference, and nothing silently falls back to the CPU."""
import ast
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import innfer_amd.lib as L
from innfer_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "innfer_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(innfer_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 24
    for n in sorted(names):
        assert hasattr(L.lib, n), f"libinnfer_amd.so does not export {n}"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in innfer_amd/lib.py"
    assert set(L.SIGNATURES) <= names, "lib.py binds symbols the header does not declare"
    assert L.lib.innfer_version() == L.ABI_VERSION == int(re.search(r"#define INNFER_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "innfer_amd.h")).read()).group(1))


def test_chop_plan_matches_reference_geometry(golden):
    g = golden("g1_geometry")
    for key in g.files:
        h, w = map(int, key[4:].split("x"))
        ps, ys, xs = L.chop_plan(h, w)
        assert ps == min(h, w, 200)
        org = np.array([(y, x) for y in ys for x in xs], dtype=np.int32)
        assert np.array_equal(org, g[key]), key
    for (h, w, n) in [(1080, 1920, 190), (2160, 3840, 798), (4320, 7680, 3268), (128, 128, 1)]:
        _, ys, xs = L.chop_plan(h, w)
        assert len(ys) * len(xs) == n
    with pytest.raises(ValueError):
        L.chop_plan(0, 10)


def test_blend_profile_is_torch_linspace_bit_exact(golden):
    g = golden("g2_blend")
    for key in g.files:
        if key.startswith("profile_"):
            P, s = key[8:].split("_")
            assert np.array_equal(L.blend_profile(int(P[1:]), 0.5, int(s[1:])), g[key]), key
    with pytest.raises(ValueError):          # odd small patch: the reference raises too (utils.py:415)
        L.blend_profile(151, 0.5, 1)


def _pack_reference(w, K, Cc):
    """numpy restatement of the documented panel layout (csrc/conv3x3.hip conv_pack)."""
    nt = 4 if K >= 64 else (2 if K >= 32 else 1)
    rows = 16 * nt
    groups = (K + rows - 1) // rows
    out = np.zeros((groups, Cc // 32, 9, rows, 4, 8), dtype=np.float16)
    for g in range(groups):
        for R in range(rows):
            t, rho = R >> 4, R & 15
            oc = g * rows + 4 * nt * (rho >> 2) + 4 * t + (rho & 3)
            if oc >= K:
                continue
            for sg in range(4):
                cg = sg ^ (((R >> 2) & 1) << 1)
                for c in range(Cc // 32):
                    ic = c * 32 + cg * 8
                    out[g, c, :, R, sg, :] = w[oc, ic:ic + 8].reshape(8, 9).T
    return out


@pytest.mark.parametrize("K,Cc", [(32, 64), (64, 96), (16, 32), (3, 64), (256, 64)])
def test_weight_panel_layout(K, Cc):
    w = synth.uniform((K, Cc, 3, 3), 7, -1, 1)
    n = L.lib.innfer_conv3x3_packed_bytes(K, Cc)
    buf = np.zeros(n, dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, K, Cc, buf.ctypes.data))
    ref = _pack_reference(w, K, Cc)
    assert n == ref.nbytes
    assert np.array_equal(buf.view(np.float16), ref.reshape(-1))
    assert L.lib.innfer_conv3x3_packed_bytes(32, 48) == 0
    assert L.lib.innfer_pack_conv3x3(w.ctypes.data, K, 48, buf.ctypes.data) == L.ERR_INVALID


def _new_arch(sd_old):
    fixed = {"model.0": "conv_first", "model.1.sub.23": "trunk_conv", "model.3": "upconv1", "model.6": "upconv2",
             "model.8": "HRconv", "model.10": "conv_last"}
    out = {}
    for k, v in sd_old.items():
        base, leaf = k.rsplit(".", 1)
        if base in fixed:
            out[f"{fixed[base]}.{leaf}"] = v
        else:
            out[base.replace("model.1.sub.", "RRDB_trunk.")[:-2] + "." + leaf] = v
    return out


def test_loader_inference_matches_reference(golden):
    from innfer_amd.run import infer_from_state_dict
    g = golden("g10_loader")
    cases = {}
    for scale in (1, 2, 4, 8):
        cases[f"{scale}x_old.pth"] = synth.fill_state_dict(synth.rrdbnet_shapes(nb=2, scale=scale, nf=48), 0)
    cases["4x_gray.pth"] = synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=4, in_nc=1, out_nc=1), 0)
    cases["4x_plus.pth"] = synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=4, plus=True), 0)
    cases["4x_srgan.pth"] = synth.fill_state_dict(synth.srresnet_shapes(nb=3, scale=4), 0)
    old = synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4, nf=16), 0)
    cases["4x_newarch.pth"] = _new_arch(old)
    assert sorted(cases["4x_newarch.pth"]) == list(g["newarch_keys"])
    swa = {"n_averaged": np.asarray(3)}
    swa.update({"module.module." + k: v for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=2), 0).items()})
    cases["2x_swa.pth"] = swa
    for i, name in enumerate(g["names"]):
        info = infer_from_state_dict(dict(cases[str(name)]))
        assert info["arch"] == str(g["arch"][i]), name
        assert info["scale"] == int(g["scale"][i]), name
        assert (info["nf"], info["nb"], info["in_nc"], info["out_nc"]) == \
            (int(g["nf"][i]), int(g["nb"][i]), int(g["in_nc"][i]), int(g["out_nc"][i])), name
        assert "|".join(sorted(info["state_dict"].keys())) == str(g["keys"][i]), name
        assert info["plus"] == (str(name) == "4x_plus.pth")
    with pytest.raises(Exception, match="Could not infer"):
        infer_from_state_dict({"foo.weight": np.zeros(1)})
    pan = infer_from_state_dict({"SCPA_trunk.0.conv1_a.weight": np.zeros(1), "upsample.1.weight": np.zeros(1),
                                 "upsample.6.weight": np.zeros(1)})
    assert pan["arch"] == "pan" and pan["scale"] == 4 and pan["net_params"]["type"] == "pan_net"
    assert infer_from_state_dict({"SCPA_trunk.0.conv1_a.weight": np.zeros(1)}, scale=2)["net_params"]["scale"] == 2
    ppon = infer_from_state_dict({"CFEM.0.weight": np.zeros(1), "CRM.1.weight": 0, "CRM.4.weight": 0, "CRM.6.weight": 0, "CRM.8.weight": 0})
    assert ppon["arch"] == "ppon" and ppon["scale"] == 4 and ppon["net_params"]["type"] == "ppon" and ppon["net_params"]["nb"] == 24
    wbc = infer_from_state_dict({"conv_9.weight": np.zeros((3, 32, 7, 7)), "conv.weight": np.zeros((32, 3, 7, 7))})
    assert wbc["arch"] == "wbcunet" and wbc["scale"] == 1 and wbc["net_params"] == {"type": "wbcunet_net", "nf": 32, "mode": "pt"}


def test_default_configs_match_reference(golden):
    from innfer_amd.utils.defaults import get_network_G_config
    table = json.loads(str(golden("g12_defaults")["table"]))
    for key, ref in table.items():
        if key.startswith("str:"):
            kind, scale = key[4:].split("|")
            got = get_network_G_config(kind, int(scale))
        elif key.startswith("which_model_G:"):
            kind, scale = key[len("which_model_G:"):].split("|")
            got = get_network_G_config({"which_model_G": kind}, int(scale))
        else:
            kind, scale, extra = key.split("|", 2)
            got = get_network_G_config(dict(type=kind, **json.loads(extra)), int(scale))
        assert got == ref, key
    with pytest.raises(NotImplementedError):
        get_network_G_config("nope", 4)


def test_module_shells_carry_reference_keys_and_refuse_cpu():
    from innfer_amd.architectures import get_network
    from innfer_amd.utils.defaults import get_network_G_config
    net = get_network(get_network_G_config({"type": "esrgan", "nb": 2}, 4))
    shapes = synth.rrdbnet_shapes(nb=2, scale=4)
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, 0).items()}, strict=True)
    bad = dict(sd); bad.pop("model.0.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 3, 8, 8))
    srg = get_network(get_network_G_config({"type": "srgan", "nb": 2}, 4))
    assert list(srg.state_dict().keys()) == list(synth.srresnet_shapes(nb=2, scale=4).keys())
    # the class's own defaults (norm_type='batch', mode='NAC': SRResNet_arch.py:16-17) and act() spellings (block.py:86-90), golden G26 pins the keys
    from innfer_amd.architectures.SRResNet_arch import SRResNet
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    assert list(SRResNet(3, 3, 64, 2).state_dict()) == list(synth.srresnet_shapes(nb=2, scale=4, upsample_mode="upconv", norm=True, mode="NAC"))
    assert SRResNet(3, 3, 64, 1, act_type="LRelu").trunk_act == RRDBNet(3, 3, 64, 1, act_type="lrelu").trunk_act == 1
    plus = get_network(get_network_G_config({"type": "esrgan", "plus": True, "nb": 1}, 4))
    assert list(plus.state_dict().keys()) == list(synth.rrdbnet_shapes(nb=1, scale=4, plus=True).keys())
    m = get_network(get_network_G_config({"type": "mesrgan", "nb": 2}, 4))          # new-arch ESRGAN built directly
    assert list(m.state_dict().keys()) == list(synth.mrrdbnet_shapes(nb=2).keys())
    from innfer_amd.architectures.keys import mrrdb_key_of
    from innfer_amd.utils.utils import mod2normal
    old = mod2normal({k: k for k in synth.mrrdbnet_shapes(nb=23)})            # the reference's own conversion (nb 23)
    assert all(mrrdb_key_of(k_old.rsplit(".", 1)[0], 23) + "." + k_old.rsplit(".", 1)[1] == k_new for k_old, k_new in old.items())
    assert get_network(get_network_G_config("wbcunet_tf", 1)).mode == "tf"
    g15 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g15_wbcunet.npz"))
    wb = get_network(get_network_G_config("wbcunet", 1))
    assert {k: tuple(v.shape) for k, v in wb.state_dict().items()} == {str(k): tuple(ast.literal_eval(str(v))) for k, v in zip(g15["keys"], g15["shapes"])}
    g14 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_resnet9.npz"))
    cg = get_network(get_network_G_config("resnet_9blocks", 1))
    assert {k: tuple(v.shape) for k, v in cg.state_dict().items()} == {str(k): tuple(ast.literal_eval(str(v))) for k, v in zip(g14["keys"], g14["shapes"])}
    bn = get_network(get_network_G_config({"type": "resnet_9blocks", "norm_type": "batch"}, 1))        # BatchNorm2d: no conv biases, running statistics
    assert "model.2.running_var" in bn.state_dict() and "model.1.bias" not in bn.state_dict() and "model.26.bias" in bn.state_dict()
    with pytest.raises(NameError):
        get_network(get_network_G_config({"type": "resnet_9blocks", "norm_type": "group"}, 1))
    g13 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_ppon.npz"))
    ppon = get_network(get_network_G_config("ppon", 4))
    assert {k: tuple(v.shape) for k, v in ppon.state_dict().items()} == {str(k): tuple(ast.literal_eval(str(v))) for k, v in zip(g13["keys"], g13["shapes"])}
    g8 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_pan.npz"))
    pan = get_network(get_network_G_config("pan", 4))
    psd = pan.state_dict()
    assert {k: tuple(v.shape) for k, v in psd.items()} == {str(k): tuple(ast.literal_eval(str(v))) for k, v in zip(g8["keys"], g8["shapes"])}
    noattn = get_network(get_network_G_config({"type": "pan", "self_attention": False}, 4))
    assert not any(k.startswith("FSA") for k in noattn.state_dict())
    with pytest.raises(NotImplementedError):
        get_network(get_network_G_config({"type": "pan", "ups_inter_mode": "bicubic"}, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        pan(torch.zeros(1, 3, 8, 8))
    unet = get_network(get_network_G_config("p2p_256", 1))
    assert len(unet.state_dict()) == 82 and "model.model.1.model.2.running_mean" in unet.state_dict()
    with pytest.raises(RuntimeError, match="no CPU path"):
        unet(torch.zeros(1, 3, 256, 256))
    from innfer_amd.run import Model
    with pytest.raises(RuntimeError):
        Model("nowhere.pth", arch="infer", device="cpu")


def test_entry_points_report_errors_without_a_gpu():
    a = L.ConvArgs()
    assert L.lib.innfer_conv3x3_f16(C.byref(a), None) == L.ERR_INVALID
    assert "null" in L.last_error()
    h = C.c_void_p()
    assert L.lib.innfer_rrdbnet_create(C.byref(h), 3, 3, 64, 1, 32, 5, 0) == L.ERR_UNSUPPORTED      # scale 5
    L.check(L.lib.innfer_rrdbnet_create(C.byref(h), 3, 3, 64, 1, 32, 3, 0))                           # scale 3: ONE up stage (factor 3)
    assert L.lib.innfer_net_num_convs(h) == 1 + 3 * 5 + 1 + 1 + 2 and L.lib.innfer_net_scale(h) == 3
    L.lib.innfer_net_destroy(h)
    L.check(L.lib.innfer_rrdbnet_create(C.byref(h), 3, 3, 64, 1, 32, 4, 1))                           # ESRGAN+
    assert L.lib.innfer_net_num_convs(h) == 1 + 3 * 6 + 1 + 2 + 2
    L.lib.innfer_net_destroy(h)
    L.check(L.lib.innfer_rrdbnet_create(C.byref(h), 3, 3, 64, 23, 32, 4, 0))
    assert L.lib.innfer_net_num_convs(h) == 351                                                      # 351 convs
    key = C.create_string_buffer(128); K = C.c_int(); Cc = C.c_int()
    L.check(L.lib.innfer_net_conv_info(h, 5, key, 128, C.byref(K), C.byref(Cc)))
    assert (key.value.decode(), K.value, Cc.value) == ("model.1.sub.0.RDB1.conv5.0", 64, 192)
    # 35 853 696 FLOP per input pixel
    assert L.lib.innfer_net_flops(h, 1, 1080, 1920) == pytest.approx(35853696.0 * 1080 * 1920, rel=1e-12)
    assert L.lib.innfer_net_workspace_bytes(h, 1, 1080, 1920) > 0
    assert L.lib.innfer_net_forward(h, None, 0, None, 0, 1, 8, 8, None, 0, None) == L.ERR_INVALID
    # input maps ('NAC' conv blocks): argument checks come before any device call
    assert L.lib.innfer_net_set_conv_input_map(h, 999, None, None, 1) == L.ERR_INVALID
    assert L.lib.innfer_net_set_conv_input_map(h, 1, None, None, 3) == L.ERR_INVALID                 # act: 0, 1, 2
    assert L.lib.innfer_net_set_conv_input_map(h, 1, None, None, 1) == L.ERR_UNSUPPORTED             # a dense-block conv: no map in front of it
    assert "model.1.sub.0.RDB1.conv1.0" in L.last_error()
    L.check(L.lib.innfer_net_set_conv_input_map(h, 1, None, None, 0))                                # removing an absent map is fine
    L.lib.innfer_net_destroy(h)
    assert L.lib.innfer_srresnet_create_ex(C.byref(h), 3, 3, 32, 2, 3, 2, 1.0, 0) == L.ERR_UNSUPPORTED   # PixelShuffle(3) on 32 features
    L.check(L.lib.innfer_srresnet_create_ex(C.byref(h), 3, 3, 64, 2, 3, 2, 1.0, 0))                      # ... on 64: conv 64 -> 576
    L.check(L.lib.innfer_net_conv_info(h, 1 + 2 * 2 + 1, key, 128, C.byref(K), C.byref(Cc)))
    assert (key.value.decode(), K.value, Cc.value) == ("model.2", 576, 64) and L.lib.innfer_net_scale(h) == 3
    L.lib.innfer_net_destroy(h)


def test_tile_row_sharding():
    from innfer_amd.parallel import shard_tile_rows
    for rows, cols, world in [(43, 76, 8), (21, 38, 8), (10, 19, 4), (3, 5, 8), (1, 1, 2)]:
        spans = [shard_tile_rows(rows, cols, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == rows * cols
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        counts = [c // cols for _, c in spans]
        assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    assert [shard_tile_rows(43, 76, 8, r)[1] // 76 for r in range(8)] == [6, 6, 6, 5, 5, 5, 5, 5]


def test_cli_helpers_match_reference(golden, tmp_path, monkeypatch):
    """run.py:227-315 + utils.py:250-264 against the reference's own functions (golden G20): scale from the file name, model lookup in ./models
    (exact, partial, chains with + and >), the per-architecture presets, modcrop; and the flag surface of the parser (run.py:320-331)."""
    from innfer_amd import run as R
    from innfer_amd.utils import utils as U
    t = json.loads(str(golden("g20_cli")["table"]))
    assert [R.get_scale_name(n) for n in t["names"]] == t["scales"]
    assert [R.get_scale_name("4x_foo.pth", 2), R.get_scale_name("foo.pth", 2)] == t["with_arg"]
    assert {"pix2pix": R.pix2pix_extras, "cyclegan": R.cyglegan_extras, "default": R.default_extras} == t["extras"]
    for key, shape in t["crops"].items():
        dims, s = key.split("@")
        h, w, c = map(int, dims.split("x"))
        assert list(U.modcrop(np.zeros((h, w, c) if c else (h, w), np.uint8), int(s)).shape) == shape, key
    (tmp_path / "models" / "sub").mkdir(parents=True)
    for f in ("4x_alpha.pth", "1x_beta.pth", os.path.join("sub", "2x_gamma.pth")):
        (tmp_path / "models" / f).write_bytes(b"")
    monkeypatch.chdir(tmp_path)
    for q, (paths, scales) in t["chain"].items():
        got_p, got_s = R.parse_models(q)
        assert [os.path.relpath(p, tmp_path) for p in got_p] == paths and got_s == scales, q
    with pytest.raises(ValueError):
        R.parse_models("a")                       # 'a' matches alpha, beta and gamma
    with pytest.raises(ValueError):
        R.parse_models("nope.pth")
    a = R.build_parser().parse_args(["-m", "4x.pth", "-a", "p2p_256", "-i", "in", "-o", "out", "-cf", "-comp", "-no_fp16", "-norm", "-s", "4"])
    assert vars(a) == {"models": "4x.pth", "arch": "p2p_256", "input": "in", "output": "out", "scale": "4", "cf": True, "comp": True,
                       "no_gpu": True, "no_fp16": False, "norm": True}
    assert vars(R.build_parser().parse_args(["-models", "m.pth", "-cpu"]))["no_gpu"] is False
    with pytest.raises(RuntimeError, match="MI355X only"):
        R.main(["-m", "m.pth", "-cpu"])            # no silent CPU path


def test_image_files_round_trip_in_opencv_channel_order(tmp_path):
    """read_img / save_img (utils.py:68-96) through PIL when OpenCV is absent: BGR, BGRA, gray, 16-bit gray come back bit for bit, an
    unreadable file is None, merge_imgs enlarges the smaller image like cv2.resize(INTER_NEAREST)."""
    from innfer_amd.utils import utils as U
    bgr = synth.image_u8(9, 13, 3, 1)
    bgra = synth.image_u8(7, 5, 4, 2)
    gray = synth.image_u8(6, 8, 1, 3)[:, :, 0]
    g16 = (synth.image_u8(6, 8, 1, 4)[:, :, 0].astype(np.uint16) * 257)
    for name, im in (("a.png", bgr), ("b.png", bgra), ("c.png", gray), ("d.png", g16)):
        p = str(tmp_path / name)
        U.save_img(im, p)
        back = U.read_img(p)
        assert back.dtype == im.dtype and np.array_equal(back, im), name
    if not U.cv2_available:
        from PIL import Image
        rgb = np.asarray(Image.open(str(tmp_path / "a.png")))
        assert np.array_equal(rgb[:, :, ::-1], bgr)                  # the file holds RGB: what was saved was BGR
    (tmp_path / "junk.png").write_bytes(b"not an image")
    assert U.read_img(str(tmp_path / "junk.png")) is None
    assert U.get_images_paths(str(tmp_path)) == sorted(str(tmp_path / n) for n in ("a.png", "b.png", "c.png", "d.png", "junk.png"))
    with pytest.raises(AssertionError):
        U.get_images_paths(str(tmp_path / "a.png"))
    m = U.merge_imgs([bgr, np.repeat(np.repeat(bgr, 2, 0), 2, 1)])
    assert m.shape == (18, 52, 3) and np.array_equal(m[:, :26], m[:, 26:])


def _kernel_metadata(so_path):
    """(kernel name, private_segment_fixed_size, vgpr_count, sgpr_spill, vgpr_spill) of every gfx950 kernel in the shared library: the
    uncompressed clang offload bundles of its .hip_fatbin section -> the gfx950 ELF code objects -> the msgpack AMDGPU metadata note."""
    import struct
    import msgpack
    blob = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    for m in re.finditer(magic, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + 24)
        p = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" not in triple or size == 0:
                continue
            elf = blob[base + off:base + off + size]
            assert elf[:4] == b"\x7fELF"
            shoff, = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
            for i in range(shnum):
                sh = shoff + i * shentsize
                sh_type, = struct.unpack_from("<I", elf, sh + 4)
                sh_offset, sh_size = struct.unpack_from("<QQ", elf, sh + 0x18)
                if sh_type != 7:                                  # SHT_NOTE
                    continue
                q, end = sh_offset, sh_offset + sh_size
                while q + 12 <= end:
                    namesz, descsz, ntype = struct.unpack_from("<III", elf, q)
                    d0 = q + 12 + (namesz + 3) // 4 * 4
                    if ntype == 32:                               # NT_AMDGPU_METADATA
                        md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                        for k in md.get("amdhsa.kernels", []):
                            out.append((k[".name"], k[".private_segment_fixed_size"], k[".vgpr_count"], k.get(".sgpr_spill_count", 0), k.get(".vgpr_spill_count", 0)))
                    q = d0 + (descsz + 3) // 4 * 4
    return out


def test_no_shipped_kernel_uses_scratch_memory():
    """A kernel whose per-lane arrays land in scratch memory stays correct and gets slow without a trace in any parity test (the 7x7 instantiation
    of conv3x3_pc ran 11 x slower for a while because a loader lambda was no longer inlined).  Every kernel of the shipped library must have a
    private segment of 0 bytes and no spilled vector registers; the one exception is a fallback instantiation no launch path selects."""
    ks = _kernel_metadata(L.LIB_PATH)
    assert len(ks) >= 60, len(ks)
    allowed = ("conv3x3_mfmaILi4ELi4ELi0E",)          # 16-row x 64-channel two-workgroup form: diagnostic builds only (INNFER_PC=0, INNFER_RPW64=4)
    bad = [(n, priv, vs) for (n, priv, _v, _ss, vs) in ks if (priv or vs) and not any(a in n for a in allowed)]
    assert not bad, bad


def test_chop_plan_equals_oracle_geometry_on_random_sizes():
    """innfer_chop_plan (C, host) against the oracle's restatement of run.py:176-181 / utils.py:350-365 for 400 seeded image sizes, patch sizes and
    steps -- the oracle itself is pinned on the reference's unfold geometry by golden G1."""
    from oracle.tiles import chop_geometry
    rng = np.random.RandomState(7)
    for _ in range(400):
        h, w = int(rng.randint(2, 1300)), int(rng.randint(2, 1300))
        patch = int(rng.choice([200, 200, 200, 64, 150, 333, 17]))
        step = float(rng.choice([0.5, 0.5, 0.75, 1.0]))
        ps, ys, xs = chop_geometry(h, w, patch, step)
        if int(ps * step) <= 0:
            continue
        assert L.chop_plan(h, w, patch, step) == (ps, ys, xs), (h, w, patch, step)
        first, count = C.c_int(), C.c_int()
        n, tot = len(ys) * len(xs), 0
        for world in (1, 3, 8):
            tot = 0
            for r in range(world):
                L.check(L.lib.innfer_shard_tiles(n, world, r, C.byref(first), C.byref(count)))
                assert first.value == tot and 0 <= count.value <= -(-n // world)
                tot += count.value
            assert tot == n


def test_small_helpers_golden():
    """norm / denorm, the box-kernel helpers, the channel flips and normal2mod against the reference's own functions (golden G24)."""
    import json
    from innfer_amd.utils import utils as U
    from innfer_amd.utils import colors as Cc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g24_helpers.npz"))
    x, t = g["x"], torch.from_numpy(g["x"])
    assert np.array_equal(U.norm(t).numpy(), g["norm_t"]) and np.array_equal(U.norm(x), g["norm_np"])
    assert np.array_equal(U.denorm(t).numpy(), g["denorm_t"]) and np.array_equal(U.denorm(x, (-0.5, 1.25)), g["denorm_np"])
    with pytest.raises(TypeError):
        U.norm([0.5])
    assert np.array_equal(U.get_box_kernel(5).numpy(), g["box5"]) and np.array_equal(U.get_box_kernel([3, 7]).numpy(), g["box37"])
    assert np.allclose(U.normalize_kernel2d(torch.from_numpy(g["k"])).numpy(), g["k_norm"], atol=1e-7, rtol=0)
    rgba = torch.from_numpy(g["rgba"])
    assert np.array_equal(Cc.bgr_to_rgb(t).numpy(), g["bgr2rgb"]) and np.array_equal(Cc.rgb_to_bgr(t[0]).numpy(), g["rgb2bgr"])
    assert np.array_equal(Cc.bgra_to_rgba(rgba).numpy(), g["bgra2rgba"]) and np.array_equal(Cc.rgba_to_bgra(rgba).numpy(), g["rgba2bgra"])
    tab = json.loads(str(g["table"]))
    for ks, want in tab["pads"].items():
        assert U.compute_padding(eval(ks)) == want, ks
    old = {k: np.zeros((1,), np.float32) for k in tab["old_keys"]}
    new = U.normal2mod(dict(old))
    assert list(new.keys()) == tab["new_keys"]
    assert list(U.mod2normal(dict(new)).keys()) != [] and set(U.mod2normal(dict(new)).keys()) == set(old.keys())      # and back
    assert U.normal2mod({"a": 1}) == {"a": 1}


def test_image_loop_is_pipelined_and_keeps_the_serial_semantics(tmp_path, monkeypatch, capsys):
    """run.py main (reference run.py:404-442) over a folder with the model replaced by a stand-in (no GPU here): the reader / writer threads must
    produce one PNG per readable image, holding that image's result, report and skip a file that is no image, and surface a writer's exception."""
    from innfer_amd import run as R, synth
    from innfer_amd.utils import utils as U

    class Doubler:                                     # stands in for Model: nearest 2x of the uint8 image
        def __init__(self, *a, **k):
            pass

        def run_u8(self, img, normalize=False, fp16=True):
            return np.repeat(np.repeat(img, 2, 0), 2, 1)

    monkeypatch.setattr(R, "Model", Doubler)
    (tmp_path / "models").mkdir(); (tmp_path / "in").mkdir()
    torch.save({}, str(tmp_path / "models" / "2x_standin.pth"))
    imgs = {f"img{i:02d}": synth.image_u8(8 + i, 11 + 2 * i, 3, 700 + i) for i in range(21)}          # more images than writer threads + queue
    for k, im in imgs.items():
        U.save_img(im, str(tmp_path / "in" / f"{k}.png"))
    (tmp_path / "in" / "img05b.png").write_bytes(b"no image")
    monkeypatch.chdir(tmp_path)
    assert R.main(["-m", "2x_standin", "-i", "in", "-o", "out"]) == 0
    out = capsys.readouterr().out
    assert out.count("Error reading image") == 1 and "img05b" in out
    assert sorted(os.listdir(tmp_path / "out")) == sorted(f"{k}.png" for k in imgs)
    for k, im in imgs.items():
        assert np.array_equal(U.read_img(str(tmp_path / "out" / f"{k}.png")), np.repeat(np.repeat(im, 2, 0), 2, 1)), k
    assert R.main(["-m", "2x_standin", "-i", "in", "-o", "out_comp", "-comp"]) == 0                  # LR | SR side by side (save_img_comp)
    assert U.read_img(str(tmp_path / "out_comp" / "img00.png")).shape == (16, 44, 3)

    # inputs in sub-folders that share a stem are written to ONE output path; the reference's serial loop keeps the last one, and so must the
    # writer pool (writes to one path are chained, never concurrent).  A slow first write would otherwise win the race.
    import time
    (tmp_path / "in2" / "a").mkdir(parents=True); (tmp_path / "in2" / "b").mkdir()
    first, last = synth.image_u8(64, 64, 3, 801), synth.image_u8(9, 9, 3, 802)
    U.save_img(first, str(tmp_path / "in2" / "a" / "same.png")); U.save_img(last, str(tmp_path / "in2" / "b" / "same.png"))
    order = [p for p in U.get_images_paths(str(tmp_path / "in2"))]
    real_save = U.save_img

    def slow_first(img, path, *a, **k):
        if img.shape[0] == 128:
            time.sleep(0.5)                              # the 64 x 64 input's (earlier) write is slow
        return real_save(img, path, *a, **k)
    monkeypatch.setattr(U, "save_img", slow_first)
    assert R.main(["-m", "2x_standin", "-i", "in2", "-o", "out_same"]) == 0
    want = last if order[-1].endswith(os.path.join("b", "same.png")) else first
    assert np.array_equal(U.read_img(str(tmp_path / "out_same" / "same.png")), np.repeat(np.repeat(want, 2, 0), 2, 1))

    def failing_save(*a, **k):
        raise OSError("disk full")
    monkeypatch.setattr(U, "save_img", failing_save)
    with pytest.raises(OSError, match="disk full"):
        R.main(["-m", "2x_standin", "-i", "in", "-o", "out2"])



@pytest.mark.parametrize("K,Cc", [(32, 64), (64, 96)])
def test_split_panels(K, Cc):
    """Host packers of round 3 (no GPU needed).  innfer_pack_conv3x3_split: the panels of the conv over 3 C virtual input channels -- (w - wh) * 2^11, wh, wh
    -- so that wh + wl * 2^-11 carries w to 2^-22."""
    w = np.ascontiguousarray((synth.uniform((K, Cc, 3, 3), 11, -1, 1) / np.sqrt(9 * Cc)).astype(np.float32))
    n = L.lib.innfer_conv3x3_packed_bytes(K, Cc)
    buf = np.zeros(3 * n, dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3_split(w.ctypes.data, K, Cc, buf.ctypes.data))
    wh = w.astype(np.float16).astype(np.float32)
    wl = ((w - wh) * 2048.0).astype(np.float16).astype(np.float32)
    virt = np.concatenate([wl, wh, wh], axis=1)                       # [K, 3C, 3, 3]: exactly representable, so the reference packer's cast is the identity
    assert np.array_equal(buf.view(np.float16), _pack_reference(virt, K, 3 * Cc).reshape(-1))
    assert np.abs(wh + wl / 2048.0 - w).max() <= 2.0 ** -22 * np.abs(w).max()


def test_scpa_panel_layout_is_bank_conflict_free():
    """csrc/pan_scpa_layout.h: the permuted tap blocks of the SCPA kernels' 20 -> 20 convs.  Re-derives the lane constants (kfrag_t0 / kfrag_t1) from the header's sigma tables and
    checks what its comment claims: every real (row, octet) fragment has a slot of its own inside the 62-slot block, lanes of rows beyond the real ones read a real row of
    their own ds_read_b128 lane group, lanes of k-octet 3 read the block's zero slots, and in each of the four lane groups ({0-3, 12-15, 20-27}, ...) the distinct addresses
    of a read fall on distinct 16-byte slots mod 256 B -- for every tap (block base parity) -- so a fragment read is free of bank conflicts."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "innfer_amd", "csrc", "pan_scpa_layout.h")).read()
    sig0 = [int(v) for v in re.search(r"SIG0\[12\] = \{([^}]*)\}", src).group(1).split(",")]
    sig1 = [int(v) for v in re.search(r"SIG1\[8\] = \{([^}]*)\}", src).group(1).split(",")]
    assert sorted(sig0) == list(range(12)) and sorted(sig1) == list(range(8))
    K_T1, K_Z0, K_Z1, K_TAP = 36, 60, 61, 62                                  # slots (16 B)
    assert all(f"{name} = {v} * 16" in src for name, v in (("K_T1", K_T1), ("K_Z0", K_Z0), ("K_Z1", K_Z1), ("K_TAP", K_TAP)))

    def t0(li, lg):
        r = li if li < 12 else li - 12
        return 3 * sig0[r] + lg if lg < 3 else K_Z0

    def t1(li, lg):
        r = li if li < 8 else (li - 4 if li < 12 else li - 12)
        return K_T1 + 3 * sig1[r] + lg if lg < 3 else K_Z1

    # real fragments: one slot each, inside the block, clear of the zero slots
    real = [t0(li, lg) for li in range(12) for lg in range(3)] + [t1(li, lg) for li in range(8) for lg in range(3)]
    assert len(set(real)) == 60 and max(real) < K_Z0 and min(real) >= 0
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for f, nreal in ((t0, 12), (t1, 8)):
        for g in groups:
            addr = {}
            for lane in g:
                li, lg = lane & 15, lane >> 4
                a = f(li, lg)
                if lg < 3 and li >= nreal:                                      # a row beyond the real ones: the fragment of a real row whose lane sits in the same group
                    src_lanes = [l for l in g if (l >> 4) == lg and (l & 15) < nreal and f(l & 15, lg) == a]
                    assert src_lanes, (f.__name__, lane)
                addr[lane] = a
            for tap in range(9):                                                # the block's base moves by 62 slots per tap: every parity mod 16
                slots = {}
                for a in set(addr.values()):
                    s = (tap * K_TAP + a) % 16
                    assert s not in slots, (f.__name__, g, tap, a, slots[s])
                    slots[s] = a
