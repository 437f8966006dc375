#!/usr/bin/env python3
"""BASELINE config 5: pix2pix UNet_256, 64 x 3 x 256 x 256 (64 independent batch-1 forwards)."""
import ast, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import synth
from innfer_amd.architectures import get_network
from innfer_amd.utils.defaults import get_network_G_config
dev = torch.device("cuda:0")
net = get_network(get_network_G_config("p2p_256", 1))
sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()}
net.load_state_dict(sd, strict=True)
net = net.to(dev)
for N in ([int(os.environ['UNET_N'])] if os.environ.get('UNET_N') else (1, 8, 64)):
    x = torch.from_numpy(synth.uniform((N, 3, 256, 256), 3, -1, 1)).to(dev).half()
    for _ in range(2): y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("UNET_REPS", "5"))
    e0.record()
    for _ in range(reps): y = net(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = net.flops(N, 256, 256)
    print(f"UNet_256 N={N:2d}: {ms:8.3f} ms  {N / ms * 1e3:8.1f} img/s  {N * 65536 / ms / 1e3:7.2f} MPix/s  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)
    if N == 64:          # BASELINE config 5: one JSON line with a roofline object (algorithmic FLOPs from the engine: 12.1 GFLOP per image)
        import json
        wbytes = sum(p.numel() for p in net.parameters()) * 2.0          # fp16 weight panels, read at least once per launch
        print(json.dumps({"metric": "images/s, pix2pix UNet_256 (BASELINE config 5)", "value": round(N / ms * 1e3, 1), "unit": "img/s", "n_gpus": 1,
                          "ms_per_step": round(ms, 4), "dtype": "f16", "data": "synthetic",
                          "config": {"workload": "UnetGenerator(3,3,8,ngf 64, batch norm on the statistics of each image), 64x3x256x256 -> 64x3x256x256"},
                          "roofline": {"bound": "mfma", "achieved": round(fl / ms / 1e9, 2), "peak": 2516.6, "unit": "TFLOP/s",
                                       "frac": round(fl / ms / 1e9 / 2516.6, 4), "flops_per_forward": fl, "weight_bytes": wbytes,
                                       "note": "whole forward (63 launches), not one kernel: the network is a chain of small GEMM-shaped layers"}}), flush=True)
