import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth
import innfer_amd.lib as L
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
net = get_network(get_network_G_config("pan", 4))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
for shape in [(1, 3, 540, 960), (16, 3, 200, 200)]:
    x = torch.from_numpy(synth.uniform(shape, 3, 0, 1)).cuda()
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        launches = L.timed_launches(lambda: net(x))
        res.append(sum(m for n, m, f, b in launches if "pan_scpa_split" in n))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): net(x)
    e1.record(); torch.cuda.synchronize()
    print(shape, "scpa_split x16 ms:", [round(r, 4) for r in res], " forward ms:", round(e0.elapsed_time(e1) / 20, 3))
