import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
net = get_network(get_network_G_config("resnet_9blocks", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)          # (zero weights would pick PAN's one-tap kernels and draw less power than real data)
net = net.cuda().eval()
x = (torch.rand(16,3,256,256,device="cuda")*2-1).half()
for _ in range(3): net(x)
torch.cuda.synchronize()
