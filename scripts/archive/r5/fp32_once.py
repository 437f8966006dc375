#!/usr/bin/env python3
"""A few fp32-mode forwards of one generator (the program behind `rocprofv3 --kernel-trace --stats -- python3 scripts/r5/fp32_once.py <arch>`)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
CASES = {"pan": (4, (1, 3, 540, 960), (0, 1), False), "p2p_256": (1, (64, 3, 256, 256), (-1, 1), True), "ppon": (4, (8, 3, 200, 200), (0, 1), False),
         "resnet_9blocks": (1, (16, 3, 256, 256), (-1, 1), False), "wbcunet": (1, (1, 3, 1080, 1920), (-1, 1), False), "srgan": (4, (1, 3, 1080, 1920), (0, 1), False), "esrgan": (4, (1, 3, 1080, 1920), (0, 1), False)}
arch = sys.argv[1] if len(sys.argv) > 1 else "p2p_256"
half = len(sys.argv) > 2 and sys.argv[2] == "fp16"
scale, shape, rng, train = CASES[arch]
dev = torch.device("cuda:0")
net = get_network(get_network_G_config(arch, scale))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
net = net.train() if train else net.eval()
x = torch.from_numpy(synth.uniform(shape, 3, *rng)).to(dev)
if half:
    x = x.half()
for _ in range(6):
    net(x)
torch.cuda.synchronize()
