#!/usr/bin/env python3
"""PAN 4x with the library named by INNFER_LIB: median ms of 16x200^2 and 540x960, and a checksum of each output (same-box A/B of library builds: the fused
bilinear skip of conv_last and the strip form of pan_fsa_combine must leave the bits alone)."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
out = []
for (N, H, W) in ((16, 200, 200), (1, 540, 960), (1, 101, 135)):
    x = torch.from_numpy(synth.uniform((N, 3, H, W), 3)).to(dev).half()
    for _ in range(5): y = net(x)
    torch.cuda.synchronize()
    win = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = net(x)
        e1.record(); torch.cuda.synchronize()
        win.append(e0.elapsed_time(e1) / 20)
    out.append(f"{N}x{H}x{W} {sorted(win)[2]:.3f} ms {hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]}")
print(" | ".join(out))
