"""State-dict key -> shape tables of the reference generators (host logic, no torch).

The nn.Module shells in this package are built FROM these tables, so their
state_dict() keys are the reference's by construction and model-database .pth
files load with strict=True.
"""


def _bn(s, key, c):
    s[key + ".weight"] = (c,); s[key + ".bias"] = (c,)
    s[key + ".running_mean"] = (c,); s[key + ".running_var"] = (c,); s[key + ".num_batches_tracked"] = ()


def rrdbnet_shapes(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, scale=4, plus=False, nr=3, upsample_mode='upconv', norm=False, mode='CNA'):
    """State-dict key -> shape of the reference's old-arch ESRGAN
    (reference RRDBNet_arch.py:16-48; key layout SURVEY.md 3.3).  nr != 3: the dense blocks are `RDBs.<i>` (RRDBNet_arch.py:84-88);
    upsample_mode 'pixelshuffle': the stage's conv (nf -> 4 nf, 9 nf for scale 3) comes first (block.py:333-346)."""
    import math
    s = {"model.0.weight": (nf, in_nc, 3, 3), "model.0.bias": (nf,)}
    for b in range(nb):
        for r in range(1, nr + 1):
            p = f"model.1.sub.{b}.RDB{r}." if nr == 3 else f"model.1.sub.{b}.RDBs.{r - 1}."
            if plus:
                s[p + "conv1x1.weight"] = (gc, nf, 1, 1)
            for i in range(1, 6):
                cin = nf + (i - 1) * gc
                cout = gc if i < 5 else nf
                s[p + f"conv{i}.0.weight"] = (cout, cin, 3, 3)
                s[p + f"conv{i}.0.bias"] = (cout,)
                if norm:                        # conv_block(CNA) = conv, BatchNorm2d[, act] (block.py:244-246)
                    _bn(s, p + f"conv{i}.1", cout)
    lr = nb
    if norm and mode == 'NAC':                  # LR_conv = norm, conv (block.py:246-254): the norm layer takes the conv's slot, the conv the next
        _bn(s, f"model.1.sub.{nb}", nf)
        lr = nb + 1
    s[f"model.1.sub.{lr}.weight"] = (nf, nf, 3, 3)
    s[f"model.1.sub.{lr}.bias"] = (nf,)
    if norm and mode != 'NAC':                  # LR_conv's norm layer is flattened into the trunk's Sequential behind its conv
        _bn(s, f"model.1.sub.{nb + 1}", nf)
    n_up = 1 if scale == 3 else int(math.log(scale, 2))
    idx = 2
    for _ in range(n_up):
        if upsample_mode == 'pixelshuffle':
            f = 9 if scale == 3 else 4
            s[f"model.{idx}.weight"] = (nf * f, nf, 3, 3)
            s[f"model.{idx}.bias"] = (nf * f,)
        else:
            s[f"model.{idx + 1}.weight"] = (nf, nf, 3, 3)
            s[f"model.{idx + 1}.bias"] = (nf,)
        idx += 3
    s[f"model.{idx}.weight"] = (nf, nf, 3, 3)
    s[f"model.{idx}.bias"] = (nf,)
    s[f"model.{idx + 2}.weight"] = (out_nc, nf, 3, 3)
    s[f"model.{idx + 2}.bias"] = (out_nc,)
    return s


def mrrdbnet_shapes(in_nc=3, out_nc=3, nf=64, nb=24, gc=32):
    """State-dict key -> shape of the modified ("new"-arch) ESRGAN, MRRDBNet (reference RRDBNet_arch.py:173-231):
    the graph of the old-arch 4x RRDBNet under different names."""
    s = {"conv_first.weight": (nf, in_nc, 3, 3), "conv_first.bias": (nf,)}
    for b in range(nb):
        for r in (1, 2, 3):
            for i in range(1, 6):
                cin = nf + (i - 1) * gc
                cout = gc if i < 5 else nf
                s[f"RRDB_trunk.{b}.RDB{r}.conv{i}.weight"] = (cout, cin, 3, 3)
                s[f"RRDB_trunk.{b}.RDB{r}.conv{i}.bias"] = (cout,)
    for name in ("trunk_conv", "upconv1", "upconv2", "HRconv"):
        s[name + ".weight"] = (nf, nf, 3, 3)
        s[name + ".bias"] = (nf,)
    s["conv_last.weight"] = (out_nc, nf, 3, 3)
    s["conv_last.bias"] = (out_nc,)
    return s


def mrrdb_key_of(old_key, nb):
    """Old-arch conv key of the engine ('model.1.sub.3.RDB2.conv4.0') -> MRRDBNet's name for the same conv
    (the inverse of mod2normal, utils.py:666-698, for any nb)."""
    fixed = {"model.0": "conv_first", f"model.1.sub.{nb}": "trunk_conv", "model.3": "upconv1",
             "model.6": "upconv2", "model.8": "HRconv", "model.10": "conv_last"}
    if old_key in fixed:
        return fixed[old_key]
    assert old_key.startswith("model.1.sub.") and old_key.endswith(".0"), old_key
    return "RRDB_trunk." + old_key[len("model.1.sub."):-2]


def srresnet_layout(nb, norm=False, mode='CNA'):
    """Positions of a ResNetBlock's / LR_conv's layers inside the flattened Sequentials (SRResNet_arch.py:23-27,68-86; block.py:242-254,
    B.sequential flattens nested Sequentials): {engine key -> (conv key, BatchNorm in front of the conv or None, BatchNorm behind it or None)}.
    Engine keys are the default graph's (norm none, CNA): `model.1.sub.<b>.res.0`, `.res.2`, `model.1.sub.<nb>`."""
    lay = {}
    for b in range(nb):
        p = f"model.1.sub.{b}.res."
        if mode == 'NAC':            # [norm,] act, conv, [norm,] act, conv
            if norm: lay[p + "0"], lay[p + "2"] = (p + "2", p + "0", p + "3"), (p + "5", None, None)     # the 2nd block's norm follows conv0: folded there
            else: lay[p + "0"], lay[p + "2"] = (p + "1", None, None), (p + "3", None, None)
        elif mode == 'CNAC':         # conv, [norm,] act, conv
            if norm: lay[p + "0"], lay[p + "2"] = (p + "0", None, p + "1"), (p + "3", None, None)
            else: lay[p + "0"], lay[p + "2"] = (p + "0", None, None), (p + "2", None, None)
        else:                        # conv, [norm,] act, conv[, norm]
            if norm: lay[p + "0"], lay[p + "2"] = (p + "0", None, p + "1"), (p + "3", None, p + "4")
            else: lay[p + "0"], lay[p + "2"] = (p + "0", None, None), (p + "2", None, None)
    lr = f"model.1.sub.{nb}"
    if not norm: lay[lr] = (lr, None, None)
    elif mode == 'NAC': lay[lr] = (f"model.1.sub.{nb + 1}", lr, None)
    else: lay[lr] = (lr, None, f"model.1.sub.{nb + 1}")
    return lay


def srresnet_shapes(in_nc=3, out_nc=3, nf=64, nb=16, scale=4, upsample_mode='pixelshuffle', norm=False, mode='CNA'):
    """SRGAN/SRResNet keys (reference SRResNet_arch.py:15-46, defaults.py:53-67); upsample_mode 'upconv': Upsample, conv, act per stage;
    norm / mode: BatchNorm2d layers and the layer order of the conv blocks (srresnet_layout)."""
    import math
    s = {"model.0.weight": (nf, in_nc, 3, 3), "model.0.bias": (nf,)}
    lay = srresnet_layout(nb, norm, mode)
    ordered = [f"model.1.sub.{b}.res.{j}" for b in range(nb) for j in (0, 2)] + [f"model.1.sub.{nb}"]
    for ek in ordered:
        conv, pre, post = lay[ek]
        if pre: _bn(s, pre, nf)
        s[conv + ".weight"] = (nf, nf, 3, 3)
        s[conv + ".bias"] = (nf,)
        if post: _bn(s, post, nf)
    n_up = 1 if scale == 3 else int(math.log(scale, 2))
    idx = 2
    for _ in range(n_up):
        if upsample_mode == 'upconv':
            s[f"model.{idx + 1}.weight"] = (nf, nf, 3, 3)
            s[f"model.{idx + 1}.bias"] = (nf,)
        else:
            s[f"model.{idx}.weight"] = (nf * (9 if scale == 3 else 4), nf, 3, 3)
            s[f"model.{idx}.bias"] = (nf * (9 if scale == 3 else 4),)
        idx += 3
    s[f"model.{idx}.weight"] = (nf, nf, 3, 3)
    s[f"model.{idx}.bias"] = (nf,)
    s[f"model.{idx + 2}.weight"] = (out_nc, nf, 3, 3)
    s[f"model.{idx + 2}.bias"] = (out_nc,)
    return s
