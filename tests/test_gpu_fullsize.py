"""Full-size parity seams: BASELINE configs 3 and 4 at their TRUE sizes, value-checked.

  * config 3's blend at scale 4: 3268 tiles of 800 x 800 -> 17280 x 30720 x 3 (fp32: a 25 GB tile buffer into a 6.4 GB frame, fp16: 12.5 GB into 3.2 GB --
    byte offsets far beyond 2 GiB in the gather-form blend) through the model-free property SURVEY 4 gives for it: the blend of nearest-x4 tiles IS the
    nearest-x4 image;
  * windows of the REAL chop8k / chain4k outputs (RRDBNet-23, the bench's own input and objects) against the oracle's chop_forward restricted to the few tiles
    that cover the window (oracle.chop_forward_window: same origins, profile and += order; bit-identical to the full oracle on its window, tests/test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _sd(shapes, seed=0):
    from innfer_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed).items()}


def _rrdb(dev, nb, scale, seed=0):
    from innfer_amd import synth
    from innfer_amd.architectures.RRDBNet_arch import RRDBNet
    sd = _sd(synth.rrdbnet_shapes(nb=nb, scale=scale), seed)
    net = RRDBNet(3, 3, 64, nb, upscale=scale)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def _nearest(t, s):
    return t.repeat_interleave(s, -2).repeat_interleave(s, -1)


def test_blend_scale4_at_config3_size_beyond_2gib(dev):
    """recompose_tensor(nearest_x4(tiles), scale=4) == nearest_x4(image) at config 3's size (utils.py:372-445; SURVEY 4's model-free property).
    fp32 tiles: out = (sum_k w_k x) / (sum_k w_k) over the <= 4 tiles covering a pixel, every x the same value: one rounding per product, per addition of either
    sum and for the quotient -> |out - x| <= 8 half-ulps of a value below 1 = 4.8e-7; where ONE tile covers (the four 400 x 400 corners): fl(fl(x w) / w), two
    roundings -> <= 1.2e-7.  fp16 tiles: the fp32 blend of equal fp16 values v is v (1 +- 4.8e-7), which rounds back to v: bit-exact.
    Checked on EVERY pixel (row bands), the last band holding the largest addresses of the 25 GB / 6.4 GB buffers."""
    from innfer_amd.utils import utils as U
    h, w, s = 4320, 7680, 4
    x = torch.rand((1, 3, h, w), device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
    tiles = U.extract_patches_2d(x, (200, 200), [0.5, 0.5], batch_first=True).squeeze(0)
    n = tiles.shape[0]
    assert n == 3268
    for dt, bound in ((torch.float32, 4.8e-7), (torch.float16, 0.0)):
        xs = x.to(dt)
        up = torch.empty((n, 3, 800, 800), dtype=dt, device=dev)
        for i in range(0, n, 128):
            up[i:i + 128] = _nearest(tiles[i:i + 128].to(dt), s)
        assert up.numel() * up.element_size() > 2 ** 33          # > 8 GiB: 32-bit element AND byte offsets overflow
        r = U.recompose_tensor(up, h, w, step=0.5, scale=s)
        del up
        assert tuple(r.shape) == (1, 3, s * h, s * w) and r.dtype == dt
        worst = 0.0
        for y0 in range(0, h, 270):                              # 16 bands of 1080 HR rows
            e = (r[:, :, s * y0:s * (y0 + 270)].float() - _nearest(xs[:, :, y0:y0 + 270], s).float()).abs().max().item()
            worst = max(worst, e)
        assert worst <= bound, (dt, worst)
        if dt == torch.float32:                                  # one covering tile: the first 400 x 400 pixels (tile pitch 400) and the last 80 x 80 (the ragged last
            # tile row / column start at 16480 / 29920, their predecessors end at 17200 / 30400)
            for lr, hr in ((slice(0, 100), slice(0, 400)), (slice(-20, None), slice(-80, None))):
                e = (r[:, :, hr, hr] - _nearest(xs[:, :, lr, lr], s)).abs().max().item()
                assert e <= 1.2e-7, e
        del r
        torch.cuda.empty_cache()


def _oracle_rrdb(sd, scale):
    import oracle
    def fn(t):
        with torch.no_grad():
            return oracle.rrdbnet_forward(sd, t, nb=23, scale=scale)
    return fn


def _check_window(dev, y, ref, what):
    """SURVEY 8c for the fp16 engine against the fp32 oracle: max-abs <= 1e-2 ON [0, 1]-SCALED OUTPUTS and >= 99 % of the uint8 codes within +-1.  The synthetic networks'
    outputs are not confined to [0, 1] (the 1x RRDBNet-23 reaches several units): the absolute bounds scale with the window's largest reference magnitude (as the
    PPON / PAN tests do), and the magnitude is printed."""
    from innfer_amd.utils import utils as U
    err = (y - ref).abs()
    lim = max(1.0, ref.abs().max().item())
    print(f"{what}: max {err.max().item():.2e} mean {err.mean().item():.2e} (reference magnitude up to {lim:.2f})")
    assert err.max().item() < 1e-2 * lim and err.mean().item() < 1e-3 * lim, (what, err.max().item(), err.mean().item(), lim)
    a = U.tensor2np((y / lim).to(dev)).astype(np.int32)          # (the uint8 codes of the [0, 1]-scaled outputs)
    b = U.tensor2np((ref / lim).to(dev)).astype(np.int32)
    assert (np.abs(a - b) <= 1).mean() >= 0.99, what


def test_chop8k_real_output_windows_vs_oracle(dev):
    """BASELINE config 3 with the bench's own objects (bench.py chop_setup: synth.uniform(.., 2), RRDBNet-23 4x, ChopRunner): 4320 x 7680 -> 17280 x 30720 through
    3268 tiles.  An interior 48 x 48 window where FOUR tiles meet (tile rows 20, 21 x columns 37, 38: origins 2000 / 2100, 3700 / 3800) and the frame's last
    48 x 48 pixels (one tile; the largest addresses of the 12.5 GB tile buffer and of the 3.2 GB frame) against the oracle on exactly those tiles."""
    import oracle
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner
    H, W = 4320, 7680
    net, sd = _rrdb(dev, 23, 4)
    x = torch.from_numpy(synth.uniform((1, 3, H, W), 2))
    y = ChopRunner(net, scale=4)(x.to(dev).half())
    assert tuple(y.shape) == (1, 3, 4 * H, 4 * W) and bool(torch.isfinite(y[:, :, ::16, ::16]).all())
    fn, cache = _oracle_rrdb(sd, 4), {}
    for name, win in (("four-tile seam", (4 * 2150 - 24, 4 * 2150 + 24, 4 * 3850 - 24, 4 * 3850 + 24)),
                      ("last pixels", (4 * H - 48, 4 * H, 4 * W - 48, 4 * W))):
        ref = oracle.chop_forward_window(fn, lambda a, b, c, d: x[:, :, a:b, c:d], H, W, 4, win, cache=cache)
        _check_window(dev, y[:, :, win[0]:win[1], win[2]:win[3]].float().cpu(), ref, f"chop8k {name}")
    assert len(cache) == 5
    del y
    net.release_workspace()
    torch.cuda.empty_cache()


def test_chain4k_real_output_window_vs_oracle(dev):
    """BASELINE config 4 with the bench's own objects: the model chain RRDBNet-23 1x + RRDBNet-23 4x on 2160 x 3840 (798 tiles per stage, run.py:424-426).  The
    frame's last 64 x 64 output pixels lie in ONE stage-2 tile (origin 1960, 3640: the ragged last row / column of utils.py:354-362); its 200 x 200 input is the
    stage-1 blend over nine stage-1 tiles (origins 1800 / 1900 / 1960 x 3500 / 3600 / 3640).  Oracle: ten fp32 forwards, the two blends restricted to those tiles.
    SURVEY 8c's bound (fp16 engine vs fp32 oracle <= 1e-2) is a bound per MODEL: each stage is held to it on its own input -- stage 1 on the frame, stage 2 on the
    intermediate the chain actually fed it.  End to end the second 23-block network amplifies the first one's fp16 error (the reference's own fp16 chain does the
    same: every stage rounds to fp16, run.py:421-426): measured 2.7e-2 max / 4.6e-3 mean against the all-fp32 oracle -- asserted <= 5e-2 / 1e-2 and reported."""
    import oracle
    from innfer_amd import synth
    from innfer_amd.parallel import ChopRunner, run_chain
    H, W = 2160, 3840
    net1, sd1 = _rrdb(dev, 23, 1)
    net4, sd4 = _rrdb(dev, 23, 4)
    x = torch.from_numpy(synth.uniform((1, 3, H, W), 2))
    y = run_chain([ChopRunner(net1, 1), ChopRunner(net4, 4)], x.to(dev).half())
    assert tuple(y.shape) == (1, 3, 4 * H, 4 * W)
    y1 = ChopRunner(net1, 1)(x.to(dev).half())                   # the intermediate of the chain (the chain is the two runners back to back: asserted)
    assert torch.equal(ChopRunner(net4, 4)(y1), y)
    f1, f4, c1 = _oracle_rrdb(sd1, 1), _oracle_rrdb(sd4, 4), {}
    mid = lambda a, b, c, d: oracle.chop_forward_window(f1, lambda p, q, r, s_: x[:, :, p:q, r:s_], H, W, 1, (a, b, c, d), cache=c1)
    win = (4 * H - 64, 4 * H, 4 * W - 64, 4 * W)
    ref = oracle.chop_forward_window(f4, mid, H, W, 4, win)      # the all-fp32 chain
    assert len(c1) == 9
    ty, tx = H - 200, W - 200                                    # the last stage-2 tile's input region
    # stage 1 alone: the chain's intermediate on that region against the oracle's blend of the nine stage-1 tiles
    _check_window(dev, y1[:, :, ty:, tx:].float().cpu(), mid(ty, H, tx, W), "chain4k stage 1 (1x), last 200 x 200")
    # stage 2 alone: the oracle's 4x model on the intermediate the chain fed it
    y1c = y1.float().cpu()
    ref2 = oracle.chop_forward_window(f4, lambda a, b, c, d: y1c[:, :, a:b, c:d], H, W, 4, win)
    got = y[:, :, win[0]:win[1], win[2]:win[3]].float().cpu()
    _check_window(dev, got, ref2, "chain4k stage 2 (4x) on the chain's intermediate, last pixels")
    err = (got - ref).abs()
    lim = max(1.0, ref.abs().max().item())
    print(f"chain4k end to end vs the all-fp32 oracle: max {err.max().item():.2e} mean {err.mean().item():.2e} (reference magnitude up to {lim:.2f})")
    assert err.max().item() < 5e-2 * lim and err.mean().item() < 1e-2 * lim, (err.max().item(), err.mean().item(), lim)
