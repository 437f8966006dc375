#!/usr/bin/env python3
"""Round-6 fuzz of the chained HR tail (csrc/hr_chain.hip): random RRDBNet shapes on which the chain applies (whole 16 x 32 HR tiles, last up stage wider than 16 pixels) --
scale 2 / 4 / 8, batches, LeakyReLU / ReLU trunks, fp16 and uint8 boundaries, poisoned workspace -- chained == unchained BIT FOR BIT, and a sample of them against the oracle
(<= 1e-2 on the output range).  Prints one line per case and a summary; exit code = number of bad cases.   python3 scripts/r6/fuzz_chain.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from innfer_amd import synth
from innfer_amd.architectures.RRDBNet_arch import RRDBNet

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
dev = torch.device("cuda:0")
bad = 0
for i in range(cases):
    scale = int(rng.choice([2, 4, 4, 4, 8]))
    act = str(rng.choice(["leakyrelu", "leakyrelu", "relu"]))
    # HR = scale * (h, w) must be whole 16 x 32 tiles; the last up stage's input (HR / 2) wider than 16
    uy, ux = 16 // np.gcd(16, scale), 32 // np.gcd(32, scale)
    h = int(uy * rng.randint(1, max(2, 240 // (uy * scale) + 1)))
    w = int(ux * rng.randint(1, max(2, 320 // (ux * scale) + 1)))
    if scale * w // 2 <= 16:
        w += ux
    n = int(rng.choice([1, 1, 1, 2, 3]))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=1, scale=scale), 600 + i).items()}
    net = RRDBNet(3, 3, 64, 1, upscale=scale, act_type=act)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.uniform((n, 3, h, w), 700 + i)).to(dev).half()
    net.hr_chain = True
    y1 = net(x)
    net._ws.fill_(0xFF)
    y1b = net(x)
    net.hr_chain = False
    y0 = net(x)
    ok = torch.equal(y1, y0) and torch.equal(y1, y1b)
    msg = ""
    if i % 6 == 0:
        with torch.no_grad():
            ref = oracle.rrdbnet_forward(sd, x.float().cpu(), nb=1, scale=scale, act_type=act)
        e = (y1.float().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        ok = ok and e < 1e-2
        msg = f" vs oracle {e:.1e}"
    if i % 5 == 0:
        img = torch.from_numpy((synth.uniform((h, w, 3), 800 + i) * 255).astype(np.uint8)).to(dev)
        net.hr_chain = True
        u1 = net.forward_u8(img)
        net.hr_chain = False
        ok = ok and torch.equal(u1, net.forward_u8(img))
        msg += " u8"
    bad += 0 if ok else 1
    print(f"case {i:3d} x{scale} {act:9s} {n}x3x{h}x{w} -> {scale * h}x{scale * w}: {'ok' if ok else 'BAD'}{msg}", flush=True)
    del net
print(f"fuzz_chain: {cases} cases, {bad} bad")
sys.exit(bad)
