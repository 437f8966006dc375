#!/usr/bin/env python3
"""A/B of the row-Winograd experiment on trunk-shaped layers at 1080p: the shipped kernel (24-row tiles), the direct conv on 16 x 32 tiles
(innfer_conv_args.winograd = 2) and Winograd F(2,3) along the rows on the same tiles (1).  us per launch from HIP events, package power and sclk
from rocm-smi sampled during a 2 s loop of each variant.  python3 scripts/r3/wino_ab.py [C K]..."""
import ctypes as C, os, re, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import innfer_amd.lib as L
dev = torch.device("cuda:0")


def smi_during(fn, seconds=2.0):
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); pw = re.search(r"Package Power \(W\): ([0-9.]+)", out)
            if clk and pw: samples.append((int(clk.group(1)), float(pw.group(1))))
    th = threading.Thread(target=sampler, daemon=True); th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    stop.set(); th.join(timeout=10)
    s = samples[len(samples) // 3:] or samples
    return (sum(c for c, _ in s) / max(1, len(s)), sum(w for _, w in s) / max(1, len(s)))


def run(Cc, K, H=1080, W=1920, reps=30):
    g = H * W * 32
    slab = (torch.rand((Cc // 32) * g, device=dev) - 0.5).half()
    out = torch.empty((K // 32) * g, dtype=torch.float16, device=dev)
    w = ((np.random.RandomState(1).rand(K, Cc, 3, 3) - 0.5) / np.sqrt(9 * Cc)).astype(np.float32)
    d_bias = torch.zeros(64, device=dev)
    res = {}
    for name, mode in (("shipped", 0), ("direct16", 2), ("wino16", 1)):
        if mode == 2 and K != 32: continue
        if mode == 1:
            packed = np.zeros(L.lib.innfer_conv3x3_wino_packed_bytes(K, Cc), dtype=np.uint8)
            L.check(L.lib.innfer_pack_conv3x3_wino(w.ctypes.data, K, Cc, packed.ctypes.data))
        else:
            packed = np.zeros(L.lib.innfer_conv3x3_packed_bytes(K, Cc), dtype=np.uint8)
            L.check(L.lib.innfer_pack_conv3x3(w.ctypes.data, K, Cc, packed.ctypes.data))
        d_packed = torch.from_numpy(packed).to(dev)
        a = L.ConvArgs()
        a.d_in, a.in_group_stride, a.C = slab.data_ptr(), g, Cc
        a.d_packed, a.d_bias = d_packed.data_ptr(), d_bias.data_ptr()
        a.d_out, a.out_group_stride, a.out_ch_off, a.K = out.data_ptr(), g, 0, K
        a.N, a.H, a.W, a.act, a.winograd = 1, H, W, 1, mode
        fn = lambda: L.check(L.lib.innfer_conv3x3_f16(C.byref(a), None))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        clk, pw = smi_during(fn)
        fl = 2.0 * 9 * Cc * K * H * W
        res[name] = us
        print(f"C={Cc:3d} K={K:2d} {name:9s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s(alg)  sclk {clk:5.0f} MHz  {pw:6.0f} W", flush=True)
    return res


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    shapes = list(zip(args[0::2], args[1::2])) or [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)]
    for Cc, K in shapes:
        run(Cc, K)
