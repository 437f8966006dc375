import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures.UNet_arch import UnetGenerator
mode = sys.argv[1]
net = UnetGenerator(3,3,8,ngf=64,upsample_mode=mode)
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)          # (zero weights would pick PAN's one-tap kernels and draw less power than real data)
net = net.cuda().train()
x = (torch.rand(64,3,256,256,device="cuda")*2-1).half()
for _ in range(2): net(x)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): net(x)
e1.record(); torch.cuda.synchronize()
print(mode, e0.elapsed_time(e1)/5, "ms / 64 images", flush=True)
