#!/usr/bin/env python3
"""White-box-cartoonization UNet (wbcunet) + guided filter on a 1080p frame and a 720x1280 frame."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
from innfer_amd.utils.utils import guided_filter
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("wbcunet", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
mac = 3 * 32 * 49 + 32 * 32 * 9 / 4 + 32 * 64 * 9 / 4 + 64 * 64 * 9 / 16 + 64 * 128 * 9 / 16 + 8 * 128 * 128 * 9 / 16 + 128 * 64 * 9 / 16 \
      + 64 * 64 * 9 / 4 + 64 * 32 * 9 / 4 + 32 * 32 * 9 + 32 * 3 * 49          # per input pixel
for (N, H, W) in ((1, 720, 1280), (1, 1080, 1920)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3, -1, 1)).to(dev).half()
    for _ in range(2): y = guided_filter(x, net(x), r=1, eps=5e-3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): y = guided_filter(x, net(x), r=1, eps=5e-3)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"WBC UNet + guided filter N={N} {H}x{W}: {ms:8.3f} ms  {N * H * W / ms / 1e3:8.1f} MPix/s  {2 * mac * N * H * W / ms / 1e9:7.2f} TFLOP/s", flush=True)
