import faulthandler, sys, time, os
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
def P(*a): print(f"[{time.time()-T0:7.2f}]", *a, flush=True)
T0 = time.time()
import innfer_amd.lib as L
P("lib loaded")
dev = torch.device("cuda:0")
x = torch.zeros(4, device=dev); torch.cuda.synchronize(); P("torch cuda ok")
h = C.c_void_p()
L.check(L.lib.innfer_rrdbnet_create(C.byref(h), 3, 3, 64, 1, 32, 4, 0)); P("created", L.lib.innfer_net_num_convs(h))
n = L.lib.innfer_net_num_convs(h)
key = C.create_string_buffer(128); K = C.c_int(); Cc = C.c_int()
for i in range(n):
    L.check(L.lib.innfer_net_conv_info(h, i, key, 128, C.byref(K), C.byref(Cc)))
    w = (np.random.rand(K.value, Cc.value, 3, 3).astype(np.float32) - 0.5) / np.sqrt(9 * Cc.value)
    b = np.zeros(K.value, np.float32)
    P("set_conv", i, key.value.decode(), K.value, Cc.value)
    L.check(L.lib.innfer_net_set_conv(h, i, w.ctypes.data, b.ctypes.data))
P("weights set")
N, H, W = 1, 16, 16
xin = torch.rand(N, 3, H, W, device=dev).half()
out = torch.empty(N, 3, 4 * H, 4 * W, device=dev, dtype=torch.float16)
need = L.lib.innfer_net_workspace_bytes(h, N, H, W); P("ws bytes", need)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
cap = 64
ms, fl, kd, nn = (C.c_float * cap)(), (C.c_double * cap)(), (C.c_int * cap)(), C.c_int()
L.check(L.lib.innfer_net_forward(h, xin.data_ptr(), 0, out.data_ptr(), 0, N, H, W, ws.data_ptr(), need, None)); P("forward issued")
torch.cuda.synchronize(); P("forward done", out.float().abs().mean().item())
L.check(L.lib.innfer_net_forward_timed(h, xin.data_ptr(), 0, out.data_ptr(), 0, N, H, W, ws.data_ptr(), need, None, cap, ms, fl, kd, C.byref(nn)))
P("timed", nn.value, [round(ms[i], 4) for i in range(nn.value)])
