#!/usr/bin/env python3
"""Shader clock / package power the GPU holds under the fp32-mode UNet (bench.power_probe around net(x)): is the fp32 matrix pipe clock-limited?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("p2p_256", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).train()
for dt in (torch.float32, torch.float16):
    x = torch.from_numpy(synth.uniform((64, 3, 256, 256), 3, -1, 1)).to(dev).to(dt)
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    print(dt, bench.power_probe(lambda: net(x), seconds=4.0), flush=True)
