import torch, time
dev = torch.device("cuda:0")
def bw(fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.float16, device=dev).normal_()
    y = torch.empty_like(x)
    r = bw(lambda: x.sum(), n * 2)            # read only
    c = bw(lambda: y.copy_(x), n * 4)         # read + write
    w = bw(lambda: y.fill_(1.0), n * 2)       # write only
    print(f"{mb:5d} MB  read {r:5.2f} TB/s   copy(r+w) {c:5.2f} TB/s   write {w:5.2f} TB/s", flush=True)
