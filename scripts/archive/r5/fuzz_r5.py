#!/usr/bin/env python3
"""Round-5 fuzz: random batches / ragged sizes through every generator family in BOTH precisions (float16 tensors: the fp16 engines; float32 tensors: the split engine of
RRDBNet / SRResNet and csrc/f32ops.hip's LDS-tiled conv for the rest) against the CPU oracle, plus poisoned-workspace repeats.  Aimed at this round's new code: the row-walking
first conv (widths around 16 / 64 boundaries, 1..8 input channels), the PixelShuffle store, f32conv_tiled's tile / chunk / image-batch choices, unet_deep_post's lane forms
(UNets of 5..8 levels), pan_fsa_combine's strips (widths that are / are not multiples of 4).  Usage: fuzz_r5.py [seconds] [seed]; prints BAD lines, exit code 1 if any.
FUZZ_CROSS=1: no oracle (30 s a case on the box's host cores) -- the fp16 engine is held to the fp32 engine of the same module instead (two independent kernel sets)."""
import os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
while not os.path.isdir(os.path.join(ROOT, "innfer_amd")):          # (the script moved under scripts/archive/ in round 6)
    ROOT = os.path.dirname(ROOT)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import numpy as np, torch
import oracle
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.architectures.RRDBNet_arch import RRDBNet
from innfer_amd.architectures.SRResNet_arch import SRResNet
from innfer_amd.architectures.UNet_arch import UnetGenerator
from innfer_amd.architectures.ResNet_arch import ResnetGenerator
from innfer_amd.utils.defaults import get_network_G_config
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
dev = torch.device("cuda:0")
torch.set_num_threads(max(1, os.cpu_count() or 1))
bad = 0; done = 0; t0 = time.time()
CROSS = os.environ.get("FUZZ_CROSS") == "1"
if CROSS:
    class _NoOracle:
        def __getattr__(self, name): return lambda *a, **k: None
    oracle = _NoOracle()

def load(net, seed, bn=False):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed)
    if bn: sd = synth.fill_running_stats(sd, seed)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    net.load_state_dict(sd, strict=True)
    return sd

def check(tag, net, x, ref, tol16, tol32):
    global bad, done
    if CROSS:
        ref = net(x.to(dev).float())
        ref = tuple(r.float().cpu() for r in ref) if isinstance(ref, (tuple, list)) else ref.float().cpu()
        tol32 = 1e-30 if False else tol32
    for dt, tol in ((torch.float16, tol16), (torch.float32, tol32)):
        y = net(x.to(dev).to(dt))
        ys = y if isinstance(y, (tuple, list)) else (y,)
        rs = ref if isinstance(ref, (tuple, list)) else (ref,)
        for yi, ri in zip(ys, rs):
            e = (yi.float().cpu() - ri).abs().max().item()
            lim = tol * max(1.0, ri.abs().max().item())
            if yi.shape != ri.shape or not (e < lim):
                bad += 1; print("BAD", tag, str(dt), tuple(x.shape), f"err {e:.3e} limit {lim:.3e}", flush=True)
        if hasattr(net, "_ws") and net._ws is not None:
            net._ws.fill_(0xFF)
            y2 = net(x.to(dev).to(dt))
            y2s = y2 if isinstance(y2, (tuple, list)) else (y2,)
            if not all(torch.equal(a, b) for a, b in zip(ys, y2s)):
                bad += 1; print("BAD", tag, str(dt), tuple(x.shape), "result depends on the workspace's old contents", flush=True)
    done += 1

def family(i):
    k = i % 7
    if k == 0:      # RRDBNet: first conv widths / channel counts, up-conv phases
        in_nc = int(rng.choice([1, 2, 3, 4, 5, 8])); nf = int(rng.choice([32, 64])); scale = int(rng.choice([1, 2, 4]))
        net = RRDBNet(in_nc, 3, nf, 1, upscale=scale); sd = load(net, 900 + i)
        n, h, w = int(rng.randint(1, 4)), int(rng.randint(1, 40)), int(rng.choice([1, 15, 16, 17, 63, 64, 65, 79, 80, 81, int(rng.randint(1, 130))]))
        x = torch.from_numpy(synth.uniform((n, in_nc, h, w), 1000 + i))
        with torch.no_grad(): ref = oracle.rrdbnet_forward(sd, x, nb=1, scale=scale)
        return f"rrdb in{in_nc} nf{nf} x{scale}", net.to(dev).eval(), x, ref, 1e-2, 1e-4
    if k == 1:      # SRResNet: PixelShuffle store
        scale = int(rng.choice([2, 4])); nf = int(rng.choice([32, 64]))
        net = SRResNet(3, 3, nf, 2, upscale=scale, norm_type=None, mode="CNA", upsample_mode="pixelshuffle"); sd = load(net, 1100 + i)
        n, h, w = int(rng.randint(1, 4)), int(rng.randint(3, 60)), int(rng.randint(3, 90))
        x = torch.from_numpy(synth.uniform((n, 3, h, w), 1200 + i))
        with torch.no_grad(): ref = oracle.srresnet_forward(sd, x, nb=2, scale=scale, upsample_mode="pixelshuffle")
        return f"srresnet nf{nf} ps x{scale}", net.to(dev).eval(), x, ref, 1e-2, 1e-4
    if k == 2:      # UNet, train-mode BatchNorm per image: every deep_post lane form, f32conv image batches
        nd = int(rng.choice([5, 6, 7, 8])); ngf = int(rng.choice([32, 64])) if nd < 8 else 64
        net = UnetGenerator(3, 3, nd, ngf=ngf); sd = load(net, 1300 + i)
        m = 1 << nd
        n, h, w = int(rng.randint(1, 6)), m * int(rng.randint(1, max(2, 256 // m + 1))), m * int(rng.randint(1, max(2, 256 // m + 1)))
        x = torch.from_numpy(synth.uniform((n, 3, h, w), 1400 + i, -1.0, 1.0))
        with torch.no_grad(): ref = None if CROSS else torch.cat([oracle.unet_forward(sd, x[j:j + 1], num_downs=nd) for j in range(n)], 0)
        return f"unet d{nd} ngf{ngf}", net.to(dev).train(), x, ref, 3e-2, 1e-4
    if k == 3:      # PAN with self attention: fsa strips for widths % 4 == 0, scalar form otherwise
        scale = int(rng.choice([1, 2, 3, 4]))
        net = get_network(get_network_G_config({"type": "pan", "nb": 2}, scale)); sd = load(net, 1500 + i)
        n, h, w = int(rng.randint(1, 3)), int(rng.randint(8, 70)), int(rng.choice([8, 12, 36, 64, 100, int(rng.randint(8, 110))]))
        x = torch.from_numpy(synth.uniform((n, 3, h, w), 1600 + i))
        with torch.no_grad(): ref = oracle.pan_forward(sd, x, nb=2, scale=scale)
        return f"pan x{scale}", net.to(dev).eval(), x, ref, 1e-2, 1e-4
    if k == 4:      # CycleGAN ResNet
        nb = int(rng.choice([1, 2]))
        net = ResnetGenerator(3, 3, 64, norm_type="instance", n_blocks=nb); sd = load(net, 1700 + i)
        n, h, w = int(rng.randint(1, 3)), 4 * int(rng.randint(4, 40)), 4 * int(rng.randint(4, 40))
        x = torch.from_numpy(synth.uniform((n, 3, h, w), 1800 + i, -1.0, 1.0))
        with torch.no_grad(): ref = None if CROSS else torch.cat([oracle.resnet_forward(sd, x[j:j + 1], n_blocks=nb) for j in range(n)], 0)
        return f"resnet b{nb}", net.to(dev).eval(), x, ref, 1e-2, 1e-4
    if k == 5:      # PPON
        net = get_network(get_network_G_config({"type": "ppon", "nb": 2}, 4)); sd = load(net, 1900 + i)
        n, h, w = int(rng.randint(1, 3)), int(rng.randint(9, 50)), int(rng.randint(9, 50))
        x = torch.from_numpy(synth.uniform((n, 3, h, w), 2000 + i))
        with torch.no_grad(): ref = oracle.ppon_forward(sd, x, nb=2, scale=4)
        return "ppon", net.to(dev).eval(), x, ref, 2e-2, 1e-4
    net = get_network(get_network_G_config("wbcunet", 1)); sd = load(net, 2100 + i)
    n, h, w = 1, 16 * int(rng.randint(2, 12)), 16 * int(rng.randint(2, 12))
    x = torch.from_numpy(synth.uniform((n, 3, h, w), 2200 + i, -1.0, 1.0))
    with torch.no_grad(): ref = oracle.wbcunet_forward(sd, x, mode="pt")
    return "wbcunet", net.to(dev).eval(), x, ref, 2e-2, 1e-4

i = 0
while time.time() - t0 < budget:
    try:
        tag, net, x, ref, t16, t32 = family(i)
        check(tag, net, x, ref, t16, t32)
    except Exception as e:      # a refusal is a finding too
        bad += 1; print("BAD", i % 7, "exception", repr(e)[:300], flush=True)
    i += 1
    if i % 10 == 0: print(f"[{time.time() - t0:6.0f} s] {done} cases, {bad} bad", flush=True)
print(f"fuzz done: {done} cases in {time.time() - t0:.0f} s, bad = {bad}")
sys.exit(1 if bad else 0)
