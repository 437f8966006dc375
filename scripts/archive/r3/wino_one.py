#!/usr/bin/env python3
"""ONE conv configuration in one mode, a few launches: the program behind `rocprofv3 --pmc ... -- python3 scripts/r3/wino_one.py C K mode`
(mode 0 shipped kernel, 1 Winograd rows on 16 x 32 tiles / 32-channel groups, 2 direct conv on the same tiles)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import innfer_amd.lib as L
Cc, K, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
H, W = 1080, 1920
dev = torch.device("cuda:0")
g = H * W * 32
slab = (torch.rand((Cc // 32) * g, device=dev) - 0.5).half()
out = torch.empty((K // 32) * g, dtype=torch.float16, device=dev)
w = ((np.random.RandomState(1).rand(K, Cc, 3, 3) - 0.5) / np.sqrt(9 * Cc)).astype(np.float32)
if mode == 1:
    packed = np.zeros(L.lib.innfer_conv3x3_wino_packed_bytes(K, Cc), dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3_wino(w.ctypes.data, K, Cc, packed.ctypes.data))
else:
    packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
    L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, K, Cc, packed.ctypes.data))
d_packed, d_bias = torch.from_numpy(packed).to(dev), torch.zeros(64, device=dev)
a = L.ConvArgs()
a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g, Cc
a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g, 0, K
a.N, a.H, a.W, a.act, a.winograd = 1, H, W, 1, mode
for _ in range(6):
    L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
torch.cuda.synchronize()
