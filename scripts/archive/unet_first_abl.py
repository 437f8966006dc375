"""Diagnostic build only (make ablate): time of unet_first_mfma with its loads / stores removed (INNFER_FIRST_ABL 1 / 2 / 3)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(%r))
from innfer_amd.architectures.UNet_arch import UnetGenerator
net = UnetGenerator(3,3,5,ngf=64).cuda().train()
x = (torch.rand(64,3,256,256,device="cuda")*2-1).half()
for _ in range(2): net(x)
torch.cuda.synchronize()
import torch.profiler as P
with P.profile(activities=[P.ProfilerActivity.CUDA]) as prof:
    for _ in range(5): net(x)
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "unet_first" in e.key: print(os.environ.get("INNFER_FIRST_ABL"), e.key[:40], e.device_time_total / e.count, "us")
''' % here
for abl in ("0", "1", "2", "3"):
    env = dict(os.environ, INNFER_FIRST_ABL=abl, INNFER_LIB=os.path.join(os.path.dirname(here), "innfer_amd/lib/libinnfer_amd_ablate.so"))
    subprocess.run([sys.executable, "-c", code], env=env)
