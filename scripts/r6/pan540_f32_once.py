"""PAN 4x on the bench's 540 x 960 frame as a float32 tensor (the -no_fp16 mode), a few forwards: the program behind `scripts/r6/pmc_generic.sh panf32 scripts/r6/pan540_f32_once.py`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
x = torch.from_numpy(synth.uniform((1, 3, 540, 960), 3)).cuda()
for _ in range(4): net(x)
torch.cuda.synchronize()
