#!/usr/bin/env python3
"""CPU emulation: what would Winograd F(2x2,3x3) with fp16 operands cost the RRDBNet-23 forward in accuracy?

Emulates the HIP engine's numerics layer by layer (fp16 activations between layers, fp16 weights, fp32 accumulation, fused fp32 epilogue, one
rounding to fp16 per stored value) with the trunk convs computed three ways:
  direct   : products of fp16 operands, fp32 accumulation (what conv3x3_pc does today)
  wino16   : V = B^T d B in fp16 arithmetic (two rounding stages), U = fp16(G g G^T), fp32 accumulation over channels, fp32 output transform
  wino32   : V = fp16(B^T d B computed in fp32) (one rounding), otherwise as wino16
and compares with golden G3 (the reference's fp32 forward) and G11 (the reference's own fp16 mode).  Test bounds (tests/test_gpu_parity.py
test_rrdbnet23_x4_golden): vs G3 <= 1e-2, vs G11 <= 4e-3, vs G3 <= 1.5 x |G11 - G3|max + 1e-4, >= 99 % of uint8 codes within +-1.
Run here (no GPU).  Part of profiles/r3/winograd.txt."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from innfer_amd import synth

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def conv_direct(x16, w, b):
    return F.conv2d(x16.float(), w.half().float(), b, padding=1)


def conv_wino(x16, w, b, mode):
    """x16 [N,C,H,W] fp16; F(2x2,3x3) over 2x2 output tiles (H, W padded to even)."""
    N, C, H, W = x16.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x16.float(), (1, 1 + Wp - W, 1, 1 + Hp - H))
    # patches [N,C,th,tw,4,4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    if mode == "wino16":          # fp16 arithmetic, rounding after each 1-D stage
        t = torch.einsum("ij,nchwjk->nchwik", BT, d).half().float()
        V = torch.einsum("nchwik,lk->nchwil", t, BT).half().float()
    else:
        V = torch.einsum("ij,nchwjk,lk->nchwil", BT, d, BT).half().float()
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.float(), G).half().float()          # [K,C,4,4]
    M = torch.einsum("kcim,nchwim->nkhwim", U, V)                                   # fp32 accumulation over c
    Y = torch.einsum("ai,nkhwim,bm->nkhwab", AT, M, AT)                             # [N,K,th,tw,2,2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, Hp, Wp)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1)


def forward(sd, x, conv, nb=23):
    """The engine's schedule: fp16 storage after each fused epilogue."""
    lre = lambda t: F.leaky_relu(t, 0.2)
    st = lambda t: t.half()
    c = lambda key, t: conv(t, sd[key + ".weight"], sd[key + ".bias"])
    fea = st(conv_direct(x.half(), sd["model.0.weight"], sd["model.0.bias"]))
    t = fea
    for bi in range(nb):
        t_in = t
        for r in (1, 2, 3):
            p = f"model.1.sub.{bi}.RDB{r}."
            xs = [t]
            for i in range(1, 5):
                xs.append(st(lre(c(p + f"conv{i}.0", torch.cat(xs, 1)))))
            x5 = c(p + "conv5.0", torch.cat(xs, 1))
            y = x5 * 0.2 + t.float()
            if r == 3:
                y = y * 0.2 + t_in.float()
            t = st(y)
    t = st(c(f"model.1.sub.{nb}", t) + fea.float())
    for k in (3, 6):
        t = st(lre(c(f"model.{k}", F.interpolate(t.float(), scale_factor=2.0, mode="nearest").half())))
    t = st(lre(c("model.8", t)))
    return conv_direct(t, sd["model.10.weight"], sd["model.10.bias"])


def main():
    torch.set_num_threads(8)
    gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
    g3, g11 = np.load(os.path.join(gdir, "g3_rrdbnet23_x4.npz")), np.load(os.path.join(gdir, "g11_rrdbnet23_x4_fp16.npz"))
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(synth.rrdbnet_shapes(nb=23, scale=4), 0).items()}
    for tag, shape, seed in (("out_32", (1, 3, 32, 32), 3), ("out_16", (1, 3, 16, 16), 4)):
        x = torch.from_numpy(synth.uniform(shape, seed))
        ref16 = np.abs(g11[tag] - g3[tag]).max()
        print(f"{tag}: the reference's own fp16 mode vs its fp32: {ref16:.2e}  (bound for the engine: 1.5x + 1e-4 = {1.5 * ref16 + 1e-4:.2e})")
        for name, conv in (("direct", conv_direct), ("wino32", lambda a, w, b: conv_wino(a, w, b, "wino32")), ("wino16", lambda a, w, b: conv_wino(a, w, b, "wino16"))):
            with torch.no_grad():
                y = forward(sd, x, conv).numpy()
            e32, e16 = np.abs(y - g3[tag]), np.abs(y - g11[tag])
            codes = (np.abs(np.clip(y * 255, 0, 255).round() - np.clip(g3[tag] * 255, 0, 255).round()) <= 1).mean()
            print(f"  {name:7s} vs G3 max {e32.max():.2e} mean {e32.mean():.2e} | vs G11 max {e16.max():.2e} | codes within 1: {codes:.4f}", flush=True)


if __name__ == "__main__":
    main()
