#!/usr/bin/env python3
"""Median forward ms of the generators whose last conv ends in tanh (UNet_256 x64, CycleGAN ResNet-9 x16, WBC UNet 1080p), library from INNFER_LIB."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
out = []
for arch, shape, train in (("p2p_256", (64, 3, 256, 256), True), ("resnet_9blocks", (16, 3, 256, 256), False), ("wbcunet", (1, 3, 1080, 1920), False)):
    net = get_network(get_network_G_config(arch, 1))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev); net = net.train() if train else net.eval()
    x = torch.from_numpy(synth.uniform(shape, 3, -1, 1)).to(dev).half()
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    win = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x)
        e1.record(); torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / 20)
    out.append(f"{arch} {sorted(win)[2]:.4f} ms")
    del net, x
print(" | ".join(out))
