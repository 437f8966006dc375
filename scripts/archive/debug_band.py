import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from innfer_amd import synth
from innfer_amd.architectures.RRDBNet_arch import RRDBNet
dev = torch.device("cuda:0")
nb, scale, H, W, rows = 1, 1, 150, 70, 16
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=nb, scale=scale), 0).items()}
net = RRDBNet(3, 3, 64, nb, upscale=scale); net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
x = torch.from_numpy(synth.uniform((1, 3, H, W), 22)).to(dev).half()
y = net(x); torch.cuda.synchronize()
A = net._ws.clone().cpu().numpy().view(np.float16)
net.band_rows = rows
z = net(x); torch.cuda.synchronize()
B = net._ws.clone().cpu().numpy().view(np.float16)
px = H * W
def al(v): return (v + 255) & ~255
off = 0
regions = [("fea", 64)] + [(f"slab{i}", 192) for i in range(3)] + [("trunk", 64), ("hr", 64)]
for name, ch in regions:
    n = px * ch
    a = A[off // 2: off // 2 + n].reshape(ch // 32, H, W, 32); b = B[off // 2: off // 2 + n].reshape(ch // 32, H, W, 32)
    for g in range(ch // 32):
        d = np.argwhere(a[g] != b[g])
        if len(d):
            print(name, "group", g, "nbad", len(d), "rows", sorted(set(d[:, 0].tolist()))[:10], "cols", sorted(set(d[:, 1].tolist()))[:10], "ch", sorted(set(d[:, 2].tolist()))[:10],
                  "vals", a[g][tuple(d[0])], b[g][tuple(d[0])])
    off += al(n * 2)
print("out nbad", int((y != z).sum().item()))
