#!/usr/bin/env python3
"""Per-launch breakdown of the fp32 modes (float32 tensors / -no_fp16) from the library's launch timer: kernel family, launches, ms, algorithmic TFLOP/s / GB/s.
Usage: fp32_breakdown.py [arch ...]   (default: p2p_256 pan)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
CASES = {"pan": (4, (1, 3, 540, 960), (0, 1), False), "p2p_256": (1, (64, 3, 256, 256), (-1, 1), True), "ppon": (4, (8, 3, 200, 200), (0, 1), False),
         "resnet_9blocks": (1, (16, 3, 256, 256), (-1, 1), False), "wbcunet": (1, (1, 3, 1080, 1920), (-1, 1), False)}
detail = os.environ.get("DETAIL") == "1"
for arch in (sys.argv[1:] or ["p2p_256", "pan"]):
    scale, shape, rng, train = CASES[arch]
    net = get_network(get_network_G_config(arch, scale))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    net = net.train() if train else net.eval()
    x = torch.from_numpy(synth.uniform(shape, 3, *rng)).to(dev)
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net(x)
    e1.record(); torch.cuda.synchronize()
    print(f"== {arch} {shape} fp32 mode: {e0.elapsed_time(e1) / 5:.3f} ms per forward")
    launches = L.timed_launches(lambda: net(x))
    agg = {}
    for name, ms, fl, by in launches:
        key = (name, round(fl / 1e6), round(by / 1e3)) if detail else name
        a = agg.setdefault(key, [0.0, 0.0, 0.0, 0])
        a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
    tot = sum(a[0] for a in agg.values())
    for key, (ms, fl, by, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"  {str(key):70s} n={n:4d} {ms:9.4f} ms ({100 * ms / tot:5.1f} %)  {fl / ms / 1e9 if ms else 0:8.2f} TFLOP/s  {by / ms / 1e6 if ms else 0:8.1f} GB/s")
    print(f"  sum of launches {tot:.3f} ms")
    del net, x
    torch.cuda.empty_cache()
