#!/usr/bin/env python3
"""Sixty random grids at most 16 pixels wide (1..40 rows, batches of 1..5, 32..96 input and 64 / 128 output channels) through both stride-2 forms of the single-conv ABI
(conv3x3_pc's image pairs) against torch in fp32 on the fp16-rounded operands; the bound of test_stride2_conv_and_transposed_conv_vs_torch (4e-3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
import numpy as np, torch, torch.nn.functional as F
import test_gpu_parity as T
dev = torch.device('cuda:0')
rng = np.random.RandomState(77)
bad = 0
for it in range(60):
    N = int(rng.randint(1, 6)); Cc = 32 * int(rng.randint(1, 4)); K = 64 * int(rng.randint(1, 3))
    H = int(rng.randint(1, 41)); W = int(rng.randint(1, 17))
    b = torch.from_numpy(rng.uniform(-0.5, 0.5, K).astype(np.float32))
    x = torch.from_numpy(rng.uniform(-1, 1, (N, Cc, 2 * H, 2 * W)).astype(np.float32)).half()
    w = torch.from_numpy((rng.uniform(-1, 1, (K, Cc, 4, 4)) / np.sqrt(16 * Cc)).astype(np.float32)).half().float()
    ref = F.leaky_relu(F.conv2d(x.float(), w, b, stride=2, padding=1), 0.2)
    got = T._run_stride2(dev, x, w, b, K, "down", act=1)
    e1 = (got - ref).abs().max().item()
    x = torch.from_numpy(rng.uniform(-1, 1, (N, Cc, H, W)).astype(np.float32)).half()
    k = 4 if it % 2 == 0 else 3
    w = torch.from_numpy((rng.uniform(-1, 1, (Cc, K, k, k)) / np.sqrt(k * k * Cc / 4)).astype(np.float32)).half().float()
    ref = F.relu(F.conv_transpose2d(x.float(), w, b, stride=2, padding=1, output_padding=1 if k == 3 else 0))
    got = T._run_stride2(dev, x, w, b, K, "up", k=k, act=2)
    e2 = (got - ref).abs().max().item()
    if e1 > 4e-3 or e2 > 4e-3:
        bad += 1; print("BAD", N, Cc, K, H, W, k, e1, e2)
print("fuzz done, bad =", bad)
