import torch, time, sys
sys.path.insert(0, '.')
from innfer_amd.utils import utils as U
dev = torch.device('cuda')
for (H, W, s) in [(2160, 3840, 4), (1080, 1920, 4), (1081, 1921, 1)]:
    x = torch.rand(1, 3, H, W, device=dev).half()
    t = U.extract_patches_2d(x, (200, 200), [0.5, 0.5], batch_first=True).squeeze(0)
    hr = torch.nn.functional.interpolate(t, scale_factor=float(s), mode='nearest') if s > 1 else t
    for f, name, nbytes in ((lambda: U.extract_patches_2d(x, (200, 200), [0.5, 0.5], batch_first=True), 'extract', 2 * t.numel() * 2),
                            (lambda: U.recompose_tensor(hr, H, W, step=0.5, scale=s), 'recompose', hr.numel() * 2 + 3 * H * W * s * s * 2)):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:10s} {H}x{W} x{s}: {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s algorithmic", flush=True)
# pre / post and colour fix on device buffers (no PCIe): the kernels behind np2tensor / tensor2np / color_fix at 1080p -> 4K
import innfer_amd.lib as L
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H, W, s = 1080, 1920, 4
u8 = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
x16 = torch.empty((1, 3, H, W), dtype=torch.float16, device=dev)
y16 = torch.rand((1, 3, H * s, W * s), device=dev).half()
o8 = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=dev)
ms = timed(lambda: L.check(L.lib.innfer_u8hwc_to_nchw(u8.data_ptr(), H, W, 3, 0, x16.data_ptr(), L.F16, None)))
print(f"np2tensor  {H}x{W}: {ms:8.3f} ms  {(u8.numel() + x16.numel() * 2) / ms / 1e6:8.1f} GB/s algorithmic")
ms = timed(lambda: L.check(L.lib.innfer_nchw_to_u8hwc(y16.data_ptr(), L.F16, H * s, W * s, 3, 0, o8.data_ptr(), None)))
print(f"tensor2np  {H * s}x{W * s}: {ms:8.3f} ms  {(y16.numel() * 2 + o8.numel()) / ms / 1e6:8.1f} GB/s algorithmic")
ws = torch.empty(L.lib.innfer_color_fix_workspace_bytes(H, W, H * s, W * s, 3), dtype=torch.uint8, device=dev)
f8 = torch.empty_like(o8)
ms = timed(lambda: L.check(L.lib.innfer_color_fix(u8.data_ptr(), H, W, o8.data_ptr(), H * s, W * s, 3, f8.data_ptr(), ws.data_ptr(), ws.numel(), None)))
print(f"color_fix  {H}x{W} -> {H * s}x{W * s}: {ms:8.3f} ms  (compulsory: 2 x {o8.numel() / 1e6:.0f} MB uint8 + {u8.numel() / 1e6:.0f} MB: {(2 * o8.numel() + u8.numel()) / ms / 1e6:8.1f} GB/s)")
