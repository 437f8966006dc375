#!/usr/bin/env python3
"""Device-resident uint8 frame in, uint8 frame out (EngineModule.forward_u8: np2tensor in the first conv, tensor2np in the last conv's epilogue) on the
1080p bench frame, beside the fp16 tensor forward: what the uint8 epilogue of the planar last conv costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from innfer_amd import synth
dev = torch.device("cuda:0")
net, _ = bench.build_net(dev)
img = torch.from_numpy(synth.image_u8(1080, 1920, 3, 2)).to(dev)
x = torch.from_numpy(synth.uniform((1, 3, 1080, 1920), 2)).to(dev).half()
out = torch.empty((4320, 7680, 3), dtype=torch.uint8, device=dev)
for name, fn in (("fp16 tensor forward", lambda: net(x)), ("uint8 image forward ", lambda: net.forward_u8(img, out=out))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms / frame", flush=True)
