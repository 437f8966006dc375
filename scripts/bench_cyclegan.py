#!/usr/bin/env python3
"""CycleGAN ResnetGenerator (9 blocks, ngf 64) on 256x256 images and 200x200 chop tiles."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("resnet_9blocks", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
mac = 3 * 64 * 49 + 64 * 128 * 9 / 4 + 128 * 256 * 9 / 16 + 18 * 256 * 256 * 9 / 16 + 256 * 128 * 9 / 16 + 128 * 64 * 9 / 4 + 64 * 3 * 49   # per input pixel
for (N, H, W) in ((1, 256, 256), (16, 256, 256), (16, 200, 200)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3, -1, 1)).to(dev).half()
    for _ in range(2): y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): y = net(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"CycleGAN resnet_9blocks N={N:2d} {H}x{W}: {ms:8.3f} ms  {N / ms * 1e3:8.1f} img/s  {2 * mac * N * H * W / ms / 1e9:7.2f} TFLOP/s", flush=True)
