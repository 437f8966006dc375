#!/usr/bin/env python3
"""Timing of the fp32 modes (float32 tensors / -no_fp16) beside the fp16 engines: median of 5 windows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")


def timed(net, x, reps):
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    win = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net(x)
        e1.record(); torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / reps)
    return sorted(win)[2]


for arch, scale, shape, rng, train in (("pan", 4, (1, 3, 540, 960), (0, 1), False), ("pan", 4, (16, 3, 200, 200), (0, 1), False), ("p2p_256", 1, (64, 3, 256, 256), (-1, 1), True),
                                       ("ppon", 4, (8, 3, 200, 200), (0, 1), False), ("resnet_9blocks", 1, (16, 3, 256, 256), (-1, 1), False), ("wbcunet", 1, (1, 3, 1080, 1920), (-1, 1), False)):
    net = get_network(get_network_G_config(arch, scale))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    net = net.train() if train else net.eval()
    x = torch.from_numpy(synth.uniform(shape, 3, *rng)).to(dev)
    t16 = timed(net, x.half(), 5)
    t32 = timed(net, x, 2)
    print(f"{arch:16s} {str(shape):22s} fp16 engine {t16:9.3f} ms   fp32 mode {t32:9.3f} ms   ({t32 / t16:5.1f} x)", flush=True)
    del net, x
    torch.cuda.empty_cache()
