"""SRResNet-16 4x on a 1080p frame: CNA without norm against the class defaults (BatchNorm, NAC: an input-map pass per block and before LR_conv)."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures.SRResNet_arch import SRResNet
dev = torch.device('cuda:0')
for kw in (dict(norm_type=None, mode='CNA'), dict()):
    net = SRResNet(3, 3, 64, 16, upscale=4, **kw)
    sd = synth.fill_running_stats(synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0), 0)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net = net.to(dev).eval()
    x = torch.from_numpy(synth.uniform((1, 3, 1080, 1920), 1)).to(dev).half()
    for _ in range(2): y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = net(x)
    e1.record(); torch.cuda.synchronize()
    print(kw, f"{e0.elapsed_time(e1) / 5:.3f} ms", bool(torch.isfinite(y).all()), tuple(y.shape))
