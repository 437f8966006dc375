#!/usr/bin/env python3
"""PCIe-inclusive frame time of the frame1080 workload through the reference-shaped API: host uint8 HWC BGR
frame -> np2tensor (H2D of the uint8 frame + one kernel) -> RRDBNet-23 4x -> tensor2np (uint8 on the GPU, one
D2H copy to pageable numpy memory).  Reported beside the HBM-resident number of bench.py (DESIGN.md
section 4); never used as the bench value."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from innfer_amd import synth
from innfer_amd.utils import utils as U
dev = torch.device("cuda:0")
net, _ = bench.build_net(dev)
img = synth.image_u8(1080, 1920, 3, 2)
def frame():
    x = U.np2tensor(img, device=dev).half()
    y = net(x)
    return U.tensor2np(y)
for _ in range(2): out = frame()
assert out.shape == (4320, 7680, 3) and out.dtype == np.uint8
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); frame(); ts.append(time.perf_counter() - t0)
ms = 1e3 * float(np.median(ts))
print(f"PCIe-inclusive: {ms:.2f} ms/frame = {4320 * 7680 / ms / 1e3:.1f} output MPix/s (median of 5; 6.2 MB in, 99.5 MB out as uint8)")

# the same 1080p frames through the overlapped image loop (innfer_amd/pipeline.py): uint8 in / out over pinned
# buffers on side streams
from innfer_amd.pipeline import FramePipeline
frames = [img] * 12
pipe = FramePipeline(net, scale=4, device=dev, depth=3)
for _ in pipe(frames[:3]): pass
torch.cuda.synchronize(); t0 = time.perf_counter()
n = sum(1 for _ in pipe(frames))
dt = time.perf_counter() - t0
print(f"PCIe-inclusive, overlapped loop: {1e3 * dt / n:.2f} ms/frame = {n * 4320 * 7680 / dt / 1e6:.1f} output MPix/s ({n} frames)")
