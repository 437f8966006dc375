#!/usr/bin/env python3
"""The command line's image loop (innfer_amd/run.py main) on a folder of images: wall time of the pipelined loop against the sum of its stages
(decode, GPU work incl. PCIe, encode + write) timed one by one -- SURVEY 8f n2.  Files go through PIL here (no OpenCV in the image)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from innfer_amd import run as R, synth
from innfer_amd.utils import utils as U

n_img, h, w = int(os.environ.get("N_IMG", 8)), int(os.environ.get("IMG_H", 540)), int(os.environ.get("IMG_W", 960))
with tempfile.TemporaryDirectory() as d:
    os.makedirs(f"{d}/models"); os.makedirs(f"{d}/in")
    torch.save({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4), 0).items()}, f"{d}/models/4x_synth.pth")
    for i in range(n_img):
        U.save_img(synth.image_u8(h, w, 3, 100 + i), f"{d}/in/img{i:02d}.png")
    os.chdir(d)
    R.main(["-m", "4x_synth", "-i", "in", "-o", "warm"])                       # weights, workspace, page cache
    m = R.Model(f"{d}/models/4x_synth.pth", "infer", 4)
    t = [0.0, 0.0, 0.0]
    for i in range(n_img):
        t0 = time.perf_counter(); img = U.read_img(f"{d}/in/img{i:02d}.png")
        t1 = time.perf_counter(); out = m.run_u8(img, normalize=False, fp16=True); torch.cuda.synchronize()
        t2 = time.perf_counter(); U.save_img(out, f"{d}/serial{i}.png")
        t3 = time.perf_counter()
        t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2
    t0 = time.perf_counter()
    R.main(["-m", "4x_synth", "-i", "in", "-o", "out"])
    wall = time.perf_counter() - t0
    print(f"{n_img} images {h}x{w} -> {4 * h}x{4 * w}, RRDBNet-23 4x through chop_forward, PNG via {'OpenCV' if U.cv2_available else 'PIL'}")
    print(f"stages, one by one: decode {t[0] / n_img * 1e3:.1f} ms, GPU (H2D + tiles + D2H) {t[1] / n_img * 1e3:.1f} ms, encode + write {t[2] / n_img * 1e3:.1f} ms per image; sum {sum(t):.2f} s")
    print(f"pipelined loop (1 reader, up to 16 writers): {wall:.2f} s for the folder incl. model load = {wall / n_img * 1e3:.1f} ms per image")
