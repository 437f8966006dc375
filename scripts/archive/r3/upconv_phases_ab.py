#!/usr/bin/env python3
"""nearest-2x + conv3x3 (block.py:358 upconv_block) two ways on the RRDBNet's up-conv shapes: (a) the shipped form -- the loader reads the input through the
upsampling, nine taps per HR pixel; (b) the same linear map as the four 2x2-tap output phases of ConvTranspose2d(4, 2, 1) (weights summed over the taps that
meet the same LR pixel: rows {3: w0, 1: w1 + w2, 2: w0 + w1, 0: w2}, 2.25 x fewer MACs), innfer_conv_args.transposed2x = 4.  us per launch, max |a - b|."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import innfer_amd.lib as L
dev = torch.device("cuda:0")
R = {3: [0], 1: [1, 2], 2: [0, 1], 0: [2]}


def run(H, W, Cc=64, K=64, reps=20):
    g_in, g_out = H * W * 32, 4 * H * W * 32
    slab = (torch.rand((Cc // 32) * g_in, device=dev) - 0.5).half()
    w = ((np.random.RandomState(1).rand(K, Cc, 3, 3) - 0.5) / np.sqrt(9 * Cc)).astype(np.float32)
    wt = np.zeros((Cc, K, 4, 4), np.float32)
    for ky in range(4):
        for kx in range(4):
            wt[:, :, ky, kx] = sum(w[:, :, i, j] for i in R[ky] for j in R[kx]).T
    d_bias = ((torch.rand(K, device=dev) - 0.5) * 0.1).repeat(4)       # transposed2x: the K biases once per output phase
    outs = {}
    for name in ("upsample2x", "phases"):
        out = torch.zeros((K // 32) * g_out, dtype=torch.float16, device=dev)
        a = L.ConvArgs()
        if name == "phases":
            packed = np.zeros(L.lib.innfer_convt2x_packed_bytes(K, Cc), dtype=np.uint8)
            L.check(L.lib.innfer_pack_convt2x(np.ascontiguousarray(wt).ctypes.data, K, Cc, 4, packed.ctypes.data))
            a.transposed2x = 4
        else:
            packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
            L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, K, Cc, packed.ctypes.data))
            a.upsample2x = 1
        d_packed = torch.from_numpy(packed).to(dev)
        a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g_in, Cc
        a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
        a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g_out, 0, K
        a.N, a.act = 1, 1
        a.H, a.W = (H, W) if name == "phases" else (2 * H, 2 * W)
        fn = lambda: L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        outs[name] = out
        print(f"LR {H}x{W} C={Cc} K={K} {name:11s} {us:8.1f} us", flush=True)
    d = (outs["upsample2x"].float() - outs["phases"].float()).abs()
    print(f"   max |a - b| {d.max().item():.3e}  mean {d.mean().item():.3e}  (values ~ {outs['upsample2x'].float().abs().mean().item():.3f})", flush=True)


if __name__ == "__main__":
    run(1080, 1920)
    run(2160, 3840)
