#!/usr/bin/env python3
"""Per-launch breakdown (library launch timer) of PAN 4x at 540 x 960 and 16 x 200^2, fp16 and fp32 mode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
for shape in [(1, 3, 540, 960), (16, 3, 200, 200)]:
    for dt in (torch.float16, torch.float32):
        x = torch.from_numpy(synth.uniform(shape, 3, 0, 1)).cuda().to(dt)
        for _ in range(5): net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x)
        e1.record(); torch.cuda.synchronize()
        print(f"== {shape} {dt}: {e0.elapsed_time(e1) / 20:.3f} ms per forward")
        agg = {}
        for name, ms, fl, by in L.timed_launches(lambda: net(x)):
            a = agg.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
        for k, (m, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:64s} n={n:3d} {m:8.4f} ms")
