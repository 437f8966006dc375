#!/usr/bin/env python3
"""HR_conv0 -> conv_last fused against the two-launch form on random frames of whole 16 x 32 HR tiles: 1..6 x 1..5 tiles, batches of 1..3,
scale 4 and 2, LeakyReLU / ReLU features, fp16 output tensors and uint8 images.  Agreement to the last rounding (tests/_assert_same_to_the_last_rounding)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
import numpy as np, torch
import test_gpu_parity as T
from innfer_amd import synth
from innfer_amd.architectures.RRDBNet_arch import RRDBNet
dev = torch.device('cuda:0')
rng = np.random.RandomState(123)
bad = 0
for it in range(40):
    scale = 4 if it % 3 else 2
    ty, tx, N = int(rng.randint(1, 7)), int(rng.randint(1, 6)), int(rng.randint(1, 4))
    h, w = 16 * ty // scale, 32 * tx // scale
    act = "relu" if it % 5 == 0 else "leakyrelu"
    sd = T._sd(synth.rrdbnet_shapes(nb=1, scale=scale), 300 + it)
    net = RRDBNet(3, 3, 64, 1, upscale=scale, act_type=act)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.uniform((N, 3, h, w), 900 + it)).to(dev).half()
    try:
        yf = net(x); net.fused_tail = False; y2 = net(x); net.fused_tail = True
        T._assert_same_to_the_last_rounding(yf, y2, (it, scale, N, h, w, act))
        img = torch.from_numpy(np.stack([synth.image_u8(h, w, 3, 50 + it + i) for i in range(N)])).to(dev)
        uf = net.forward_u8(img); net.fused_tail = False; u2 = net.forward_u8(img); net.fused_tail = True
        dd = (uf.int() - u2.int()).abs()
        assert dd.max().item() <= 1 and (dd > 0).float().mean().item() < 0.01, (dd.max().item(), (dd > 0).float().mean().item())
    except AssertionError as e:
        bad += 1; print("BAD", it, scale, N, h, w, act, e)
print("fuzz done, bad =", bad)
