#!/usr/bin/env python3
"""The first conv's launch time inside a 1080p forward (the library's HIP-event launch timer): SRResNet (one output slab) and RRDBNet (two)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
for arch in ("srgan", "esrgan"):
    net = get_network(get_network_G_config(arch, 4))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.uniform((1, 3, 1080, 1920), 3)).to(dev).half()
    for _ in range(3): net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): net(x)
    e1.record(); torch.cuda.synchronize()
    launches = L.timed_launches(lambda: net(x))
    first = [(n, ms) for n, ms, fl, by in launches if "first" in n]
    print(f"{arch}: {e0.elapsed_time(e1) / 5:.3f} ms per forward; first conv {first}")
    del net, x
    torch.cuda.empty_cache()
