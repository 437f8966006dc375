#!/usr/bin/env python3
"""Where a tile of the fused SCPA kernel spends its time (diagnostic library `make ablate`): INNFER_SCPA_ABL bits skip 1 P1 (conv1), 2 P2a (k1), 4 P2b
(k3 * sigmoid(k2)), 8 P3's 3x3 (k4), 16 the X fetch, 32 the stores.  Results are wrong by construction; only the per-launch time means anything."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["INNFER_LIB"] = os.path.join(REPO, "innfer_amd", "lib", "libinnfer_amd_ablate.so")
import torch
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
import innfer_amd.lib as L
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
x = torch.from_numpy(synth.uniform((1, 3, 540, 960), 3)).to(dev).half()
for abl in (0, 1, 2, 4, 8, 16, 32, 15, 31, 63, 47):
    os.environ["INNFER_SCPA_ABL"] = str(abl)
    for _ in range(3):
        net(x)
    launches = L.timed_launches(lambda: net(x))
    t = [ms for name, ms, _, _ in launches if name.startswith("pan_scpa")]
    print(f"abl={abl:2d}: {len(t)} fused SCPA launches, avg {sum(t) / len(t) * 1e3:7.1f} us", flush=True)
