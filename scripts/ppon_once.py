import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
net = get_network(get_network_G_config("ppon", 4)).cuda().eval()
x = torch.rand(1,3,540,960,device="cuda").half()
for _ in range(3): net(x)
torch.cuda.synchronize()
