#!/usr/bin/env python3
"""Every launch of one fp16 forward in order (the library's HIP-event launch timer), grouped runs collapsed: launches.py <arch> [N H W]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
CASES = {"pan": (4, (1, 3, 540, 960), (0, 1), False), "p2p_256": (1, (64, 3, 256, 256), (-1, 1), True), "ppon": (4, (8, 3, 200, 200), (0, 1), False),
         "resnet_9blocks": (1, (16, 3, 256, 256), (-1, 1), False), "wbcunet": (1, (1, 3, 1080, 1920), (-1, 1), False), "srgan": (4, (1, 3, 1080, 1920), (0, 1), False),
         "esrgan": (4, (1, 3, 1080, 1920), (0, 1), False)}
arch = sys.argv[1]
scale, shape, rng, train = CASES[arch]
if len(sys.argv) > 4: shape = (int(sys.argv[2]), 3, int(sys.argv[3]), int(sys.argv[4]))
dev = torch.device("cuda:0")
net = get_network(get_network_G_config(arch, scale))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
net = net.train() if train else net.eval()
x = torch.from_numpy(synth.uniform(shape, 3, *rng)).to(dev).half()
for _ in range(4): net(x)
torch.cuda.synchronize()
runs = [L.timed_launches(lambda: net(x)) for _ in range(5)]
rows = []
for i in range(len(runs[0])):
    ms = sorted(r[i][1] for r in runs)[2]
    name, _, fl, by = runs[0][i]
    rows.append((name, ms, fl, by))
tot = sum(r[1] for r in rows)
print(f"== {arch} {shape}: {len(rows)} launches, sum {tot * 1e3:.1f} us")
i = 0
while i < len(rows):
    j = i
    while j + 1 < len(rows) and rows[j + 1][0] == rows[i][0] and abs(rows[j + 1][2] - rows[i][2]) < 1 and abs(rows[j + 1][3] - rows[i][3]) < 1: j += 1
    n = j - i + 1
    ms = sum(r[1] for r in rows[i:j + 1]) / n
    name, _, fl, by = rows[i]
    print(f"{i:4d} x{n:<3d} {name[:64]:64s} {ms * 1e3:8.1f} us  {fl / 1e9:8.2f} GFLOP {by / 1e6:8.1f} MB  {fl / ms / 1e9 if ms else 0:7.1f} TF/s {by / ms / 1e6 if ms else 0:7.1f} GB/s")
    i = j + 1
