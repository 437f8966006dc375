"""Oracle network forwards (test infrastructure; see oracle/__init__.py).

Functional restatements over a plain {key: tensor} state dict with the
reference's key names.  All NCHW, computed in the dtype of the inputs
(fp32 for parity; the reference's `-cpu` mode is always fp32, run.py:345).
"""
import math

import torch
import torch.nn.functional as F


def _conv3(sd, key, x):
    # conv_block -> nn.Conv2d(k=3, s=1, zero padding 1, bias)  (block.py:213-236,163-166)
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=1, padding=1)


def _conv3_bn(sd, key, x):
    """conv_block(mode='CNA', norm_type='batch') of a dense block: conv `key`.0 then the eval-mode BatchNorm2d `key`.1 when the state dict has one
    (block.py:244-246; RRDBNet(norm_type='batch'), RRDBNet_arch.py:27-29)."""
    y = _conv3(sd, key + ".0", x)
    bk = key + ".1"
    if bk + ".running_mean" in sd:
        y = F.batch_norm(y, sd[bk + ".running_mean"], sd[bk + ".running_var"], sd[bk + ".weight"], sd[bk + ".bias"], training=False, eps=1e-5)
    return y


def _lrelu(x):
    # act('leakyrelu') -> nn.LeakyReLU(0.2)  (block.py:81-90)
    return F.leaky_relu(x, 0.2)


def rdb_forward(sd, prefix, x, plus=False, a=_lrelu):
    """ResidualDenseBlock_5C.forward (RRDBNet_arch.py:152-165); a: the act_type of its conv blocks."""
    x1 = a(_conv3_bn(sd, prefix + "conv1", x))
    x2 = a(_conv3_bn(sd, prefix + "conv2", torch.cat((x, x1), 1)))
    if plus:
        x2 = x2 + F.conv2d(x, sd[prefix + "conv1x1.weight"])          # :155-156
    x3 = a(_conv3_bn(sd, prefix + "conv3", torch.cat((x, x1, x2), 1)))
    x4 = a(_conv3_bn(sd, prefix + "conv4", torch.cat((x, x1, x2, x3), 1)))
    if plus:
        x4 = x4 + x2                                                   # :159-160
    x5 = _conv3_bn(sd, prefix + "conv5", torch.cat((x, x1, x2, x3, x4), 1))
    return x5 * 0.2 + x                                                # :165


def rrdb_forward(sd, prefix, x, plus=False, nr=3, a=_lrelu):
    """RRDB.forward (RRDBNet_arch.py:91-98): RDB1..RDB3, or the Sequential `RDBs` when nr != 3 (:84-88)."""
    out = x
    for r in range(1, nr + 1):
        out = rdb_forward(sd, f"{prefix}RDB{r}." if nr == 3 else f"{prefix}RDBs.{r - 1}.", out, plus, a)
    return out * 0.2 + x


def _n_upscale(scale):
    # RRDBNet_arch.py:21-23 / SRResNet_arch.py:20-22
    return 1 if scale == 3 else int(math.log(scale, 2))


def rrdbnet_forward(sd, x, nb=23, scale=4, plus=False, taps=None, finalact=None, nr=3, act_type="leakyrelu", upsample_mode="upconv", outm=None):
    """RRDBNet.forward with the flat Sequential of RRDBNet_arch.py:25-48.

    taps: optional dict filled with named intermediates (golden G3 stages).
    nr / act_type / upsample_mode: the constructor arguments of the same names (:16-18, :30-38).
    """
    _act = {"leakyrelu": _lrelu, "relu": F.relu}[act_type]
    fea = _conv3(sd, "model.0", x)
    if taps is not None:
        taps["conv_first"] = fea
    t = fea
    for b in range(nb):
        t = rrdb_forward(sd, f"model.1.sub.{b}.", t, plus, nr, _act)
        if taps is not None and b == 0:
            taps["rrdb0"] = t
    if f"model.1.sub.{nb}.running_mean" in sd:               # mode='NAC' with norm_type='batch': LR_conv = BatchNorm2d, conv (block.py:246-254)
        bk = f"model.1.sub.{nb}"
        t = F.batch_norm(t, sd[bk + ".running_mean"], sd[bk + ".running_var"], sd[bk + ".weight"], sd[bk + ".bias"], training=False, eps=1e-5)
        t = _conv3(sd, f"model.1.sub.{nb + 1}", t)
    else:
        t = _conv3(sd, f"model.1.sub.{nb}", t)
    if f"model.1.sub.{nb + 1}.running_mean" in sd:           # LR_conv's BatchNorm2d (norm_type='batch'), flattened behind its conv
        bk = f"model.1.sub.{nb + 1}"
        t = F.batch_norm(t, sd[bk + ".running_mean"], sd[bk + ".running_var"], sd[bk + ".weight"], sd[bk + ".bias"], training=False, eps=1e-5)
    t = fea + t                                   # ShortcutBlock (block.py:189-191)
    if taps is not None:
        taps["trunk"] = t
    idx = 2
    for u in range(_n_upscale(scale)):
        f = 3 if scale == 3 else 2
        if upsample_mode == "pixelshuffle":                           # block.py:333-346: conv, PixelShuffle, act
            t = _act(F.pixel_shuffle(_conv3(sd, f"model.{idx}", t), f))
        else:
            t = F.interpolate(t, scale_factor=float(f), mode="nearest")   # block.py:321-322,358
            t = _act(_conv3(sd, f"model.{idx + 1}", t))
        if taps is not None:
            taps[f"up{u}"] = t
        idx += 3
    t = _act(_conv3(sd, f"model.{idx}", t))       # HR_conv0
    y = _conv3(sd, f"model.{idx + 2}", t)         # HR_conv1
    if finalact:                                  # outact (RRDBNet_arch.py:45-48; block.py:81-101)
        y = {"relu": F.relu, "leakyrelu": _lrelu, "lrelu": _lrelu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[finalact.lower()](y)
    return _outm(y, outm)


def mrrdbnet_forward(sd, x, nb=24):
    """MRRDBNet.forward (RRDBNet_arch.py:173-199) with ResidualDenseBlock_5CM / RRDBM (:201-231): restated on its own
    parameter names, not through the old-arch function, so that the key mapping of the product is checked too."""
    def conv(key, t):
        return F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], padding=1)
    fea = conv("conv_first", x)
    t = fea
    for b in range(nb):
        r_in = t
        for r in (1, 2, 3):
            p = f"RRDB_trunk.{b}.RDB{r}."
            x0 = t
            x1 = _lrelu(conv(p + "conv1", x0))
            x2 = _lrelu(conv(p + "conv2", torch.cat((x0, x1), 1)))
            x3 = _lrelu(conv(p + "conv3", torch.cat((x0, x1, x2), 1)))
            x4 = _lrelu(conv(p + "conv4", torch.cat((x0, x1, x2, x3), 1)))
            x5 = conv(p + "conv5", torch.cat((x0, x1, x2, x3, x4), 1))
            t = x5 * 0.2 + x0
        t = t * 0.2 + r_in
    fea = fea + conv("trunk_conv", t)
    fea = _lrelu(conv("upconv1", F.interpolate(fea, scale_factor=2, mode="nearest")))
    fea = _lrelu(conv("upconv2", F.interpolate(fea, scale_factor=2, mode="nearest")))
    return conv("conv_last", _lrelu(conv("HRconv", fea)))


def _outm(y, outm):
    """The range limiters of RRDBNet.forward / SRResNet.forward (RRDBNet_arch.py:50-62, SRResNet_arch.py:47-59)."""
    if outm == "scaltanh":
        return (torch.tanh(y) + 1.0) / 2.0
    if outm == "tanh":
        return torch.tanh(y)
    if outm == "sigmoid":
        return torch.sigmoid(y)
    if outm == "clamp":
        return torch.clamp(y, min=0.0, max=1.0)
    return y


def srresnet_forward(sd, x, nb=16, scale=4, act_type="relu", res_scale=1, upsample_mode="pixelshuffle", outm=None, norm_type=None, mode="CNA"):
    """SRResNet.forward in eval mode (SRResNet_arch.py:15-91; defaults.py:53-67 builds it with norm none, CNA, relu, pixelshuffle, res_scale 1;
    the class defaults are norm_type='batch', mode='NAC').  The conv blocks follow block.py:213-254: 'CNA' / 'CNAC' = conv[, norm][, act],
    'NAC' = [norm,][act,] conv; a ResNetBlock (SRResNet_arch.py:68-91) drops the second block's act under 'CNA' and its norm + act under 'CNAC';
    LR_conv has no act.  B.sequential flattens the blocks, so the layer indices count norm and act layers."""
    a = {"relu": F.relu, "leakyrelu": _lrelu}[act_type]

    def bn(t, key):
        return F.batch_norm(t, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], training=False, eps=1e-5)

    def block(t, p, layers):          # layers: sequence of 'n' / 'a' / 'c' in module order; indices are positions in the flattened Sequential
        for i, kind in enumerate(layers):
            t = bn(t, f"{p}{i}") if kind == "n" else a(t) if kind == "a" else _conv3(sd, f"{p}{i}", t)
        return t

    n = "n" if norm_type else ""
    if mode == "NAC":
        res_layers, lr_layers = n + "a" + "c" + n + "a" + "c", n + "c"
    elif mode == "CNAC":
        res_layers, lr_layers = "c" + n + "a" + "c", "c" + n
    else:
        res_layers, lr_layers = "c" + n + "a" + "c" + n, "c" + n
    fea = _conv3(sd, "model.0", x)
    t = fea
    for b in range(nb):
        t = t + block(t, f"model.1.sub.{b}.res.", res_layers) * res_scale          # :88-91
    for i, kind in enumerate(lr_layers):                                            # LR_conv's layers continue the trunk Sequential's numbering
        t = bn(t, f"model.1.sub.{nb + i}") if kind == "n" else _conv3(sd, f"model.1.sub.{nb + i}", t)
    t = fea + t
    idx = 2
    for _ in range(_n_upscale(scale)):
        f = 3 if scale == 3 else 2
        if upsample_mode == "upconv":              # block.py:348-361
            t = a(_conv3(sd, f"model.{idx + 1}", F.interpolate(t, scale_factor=float(f), mode="nearest")))
        else:
            t = a(F.pixel_shuffle(_conv3(sd, f"model.{idx}", t), f))  # block.py:333-346
        idx += 3
    t = a(_conv3(sd, f"model.{idx}", t))
    return _outm(_conv3(sd, f"model.{idx + 2}", t), outm)


def unet_forward(sd, x, num_downs=8, eps=1e-5, training=True, norm_type="batch", upsample_mode="deconv"):
    """UnetGenerator(norm=batch, deconv).forward.  training=True: BatchNorm in TRAINING mode
    (batch statistics), as run.py runs pix2pix (meval=False, run.py:299-303); training=False:
    eval mode on the checkpoint's running statistics (Model's default meval=True, run.py:96-97).
    UNet_arch.py:107-161.  x: [N,3,256,256]; training-mode statistics are over the batch
    given (callers loop batch-1 for the per-image semantics of SURVEY D6)."""

    # norm_type 'instance' (UNet_arch.py:38-41): nn.InstanceNorm2d -- no parameters, always the statistics of the image; the convs then have
    # biases (use_bias, :101-104), read with sd.get below
    def bn(t, key):
        if norm_type in ("IN", "instance"):
            return F.instance_norm(t, eps=eps)
        if not training:
            return F.batch_norm(t, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                                training=False, momentum=0.0, eps=eps)
        return F.batch_norm(t, None, None, sd[key + ".weight"], sd[key + ".bias"],
                            training=True, momentum=0.0, eps=eps)

    # upsample_mode 'upconv' (UNet_arch.py:119-122,131-134,143-146): upconv_block (block.py:348-361) = Sequential(Upsample(nearest 2x),
    # Conv2d(3x3, zero padding)), so the conv's keys are `<i>.1.*`
    def up(t, key):
        if upsample_mode == "upconv":
            t = F.interpolate(t, scale_factor=2, mode="nearest")
            return F.conv2d(t, sd[key + ".1.weight"], sd.get(key + ".1.bias"), padding=1)
        return F.conv_transpose2d(t, sd[key + ".weight"], sd.get(key + ".bias"), stride=2, padding=1)

    def block(t, prefix, depth):
        outermost = depth == 0
        innermost = depth == num_downs - 1
        if outermost:
            d = F.conv2d(t, sd[prefix + "model.0.weight"], sd.get(prefix + "model.0.bias"), stride=2, padding=1)
            m = block(d, prefix + "model.1.", depth + 1)
            return torch.tanh(up(F.relu(m), prefix + "model.3"))
        # nn.LeakyReLU(0.2, inplace=True) is the block's first layer (UNet_arch.py:109,
        # 135,148): it rewrites the block input in place, so the skip branch of
        # torch.cat([x, model(x)]) (:160-161) carries lrelu(x), not x.
        t = F.leaky_relu(t, 0.2)
        if innermost:
            d = F.conv2d(t, sd[prefix + "model.1.weight"], sd.get(prefix + "model.1.bias"), stride=2, padding=1)
            u = bn(up(F.relu(d), prefix + "model.3"), prefix + "model.4")
            return torch.cat([t, u], 1)
        d = F.conv2d(t, sd[prefix + "model.1.weight"], sd.get(prefix + "model.1.bias"), stride=2, padding=1)
        d = bn(d, prefix + "model.2")
        m = block(d, prefix + "model.3.", depth + 1)
        u = bn(up(F.relu(m), prefix + "model.5"), prefix + "model.6")
        return torch.cat([t, u], 1)

    return block(x, "model.", 0)


def pan_forward(sd, x, nb=16, scale=4, self_attention=True, double_scpa=False, ups_inter_mode="nearest"):
    """PAN.forward (PAN_arch.py:178-222) with the defaults of defaults.py:78-89 (nf 40, unf 24,
    self_attention, nearest up-blocks): SCPA blocks (PAN_arch.py:56-99), PA / PACnv pixel attention
    (:21-55), max-pooled SAGAN self attention (block.py:398-473), bilinear(align_corners) global skip.
    self_attention / double_scpa: the constructor arguments of the same names (:115-141, :193-203)."""
    def conv(t, key, pad=0):
        return F.conv2d(t, sd[key + ".weight"], sd.get(key + ".bias"), padding=pad)

    def scpa_trunk(t, name):
        for b in range(nb):
            p = f"{name}.{b}."
            a = F.leaky_relu(conv(t, p + "conv1_a"), 0.2)
            bb = F.leaky_relu(conv(t, p + "conv1_b"), 0.2)
            a = F.leaky_relu(conv(a, p + "k1.0", 1), 0.2)
            y = torch.sigmoid(conv(bb, p + "PACnv.k2"))
            bb = conv(conv(bb, p + "PACnv.k3", 1) * y, p + "PACnv.k4", 1)
            bb = F.leaky_relu(bb, 0.2)
            t = conv(torch.cat([a, bb], 1), p + "conv3") + t
        return t

    fea = conv(x, "conv_first", 1)
    trunk = conv(scpa_trunk(fea, "SCPA_trunk"), "trunk_conv", 1)
    if double_scpa:                                                   # :195-196
        trunk = conv(scpa_trunk(trunk, "SCPA_trunk2"), "trunk_conv2", 1)
    inp = fea + trunk
    if self_attention:                                                # FSA(fea + trunk), :200-201
        xp = F.max_pool2d(inp, 4, 4)
        B, C, h, w = xp.shape
        xv = xp.reshape(B, C, h * w)
        f = F.conv1d(xv, sd["FSA.conv_f.weight"], sd["FSA.conv_f.bias"])
        g = F.conv1d(xv, sd["FSA.conv_g.weight"], sd["FSA.conv_g.bias"])
        hh = F.conv1d(xv, sd["FSA.conv_h.weight"], sd["FSA.conv_h.bias"])
        att = torch.softmax(torch.bmm(f.permute(0, 2, 1), g), dim=-1)
        out = torch.bmm(hh, att.permute(0, 2, 1)).reshape(B, C, h, w)
        out = F.interpolate(out, size=(inp.shape[2], inp.shape[3]), mode="bicubic", align_corners=False)
        t = sd["FSA.gamma"] * out + inp
    else:
        t = inp
    n_up = 1 if scale == 3 else int(math.log(scale, 2))
    for u in range(n_up):
        # pa_upconv_block builds sequential(upsample, upconv, att, a, HRconv, a) with ONE LeakyReLU
        # instance `a` (PAN_arch.py:11-19).  With two stages (4x) PAN's outer B.sequential flattens them with children(), which
        # yields each module once, so a block is 5 modules (indices 5u..5u+4) and nothing follows HRconv.  With ONE stage
        # (2x, 3x) B.sequential returns the stage's own nn.Sequential untouched (block.py:197-210), whose six slots hold `a`
        # twice: there HRconv IS followed by the LeakyReLU (pinned by golden G18 `double_noattn_x2`).
        i = 5 * u
        # B.Upsample(scale_factor, mode) (block.py:286-323): align_corners stays None, i.e. False for 'bilinear'
        t = F.interpolate(t, scale_factor=2.0 if scale != 3 else 3.0, mode=ups_inter_mode)
        t = conv(t, f"upsample.{i + 1}", 1)
        t = F.leaky_relu(t * torch.sigmoid(conv(t, f"upsample.{i + 2}.conv")), 0.2)
        t = conv(t, f"upsample.{i + 4}", 1)
        if n_up == 1:
            t = F.leaky_relu(t, 0.2)
    out = conv(t, "conv_last", 1)
    if scale > 1:
        out = out + F.interpolate(x, scale_factor=float(scale), mode="bilinear", align_corners=True)
    else:
        out = out + x
    return out


def ppon_forward(sd, x, nb=24, scale=4, alpha=1.0):
    """PPON.forward (PPON_arch.py:65-76) with RRBlock_32 / _ResBlock_32 (:79-129): every residual block is
    c1 3x3 -> LeakyReLU(0.2) -> eight dilated 3x3 convs 64->32 (rates 1..8, :83-91) -> running sums
    (:104-110) -> cat -> LeakyReLU -> 1x1 256->64 -> *0.2 + input.  Returns (out_c, out_s, out_p);
    run.py keeps element [2] (run.py:191-192,220-221)."""
    def conv(t, key, pad=1, dil=1):
        return F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], padding=pad, dilation=dil)

    def resblock(t, p):
        o1 = F.leaky_relu(conv(t, p + "c1"), 0.2)
        d = [conv(o1, p + f"d{r}", pad=r, dil=r) for r in range(1, 9)]
        parts, run = [d[0]], d[0]
        for r in range(1, 8):
            run = run + d[r]
            parts.append(run)
        o2 = conv(F.leaky_relu(torch.cat(parts, 1), 0.2), p + "c2", pad=0)
        return t + o2 * 0.2

    def rrblock(t, p):
        o = t
        for k in (1, 2, 3):
            o = resblock(o, p + f"RB{k}.")
        return o * 0.2 + t

    def recon(t, p):
        n_up = 1 if scale == 3 else int(math.log(scale, 2))
        i = 0
        for _ in range(n_up):
            t = F.interpolate(t, scale_factor=3.0 if scale == 3 else 2.0, mode="nearest")
            t = F.leaky_relu(conv(t, p + f"{i + 1}"), 0.2)
            i += 3
        t = F.leaky_relu(conv(t, p + f"{i}"), 0.2)
        return conv(t, p + f"{i + 2}")

    fea = conv(x, "CFEM.0")
    t = fea
    for b in range(nb):
        t = rrblock(t, f"CFEM.1.sub.{b}.")
    cfem = fea + conv(t, f"CFEM.1.sub.{nb}")
    out_c = recon(cfem, "CRM.")
    sfem = cfem
    for b in range(2):
        sfem = rrblock(sfem, f"SFEM.{b}.")
    out_s = recon(sfem, "SRM.") + out_c
    pfem = sfem
    for b in range(2):
        pfem = rrblock(pfem, f"PFEM.{b}.")
    out_p = alpha * recon(pfem, "PRM.") + out_s
    return out_c, out_s, out_p


def resnet_forward(sd, x, n_blocks=9, eps=1e-5, padding_type="reflect", use_dropout=False, norm_type="instance", training=False):
    """ResnetGenerator(norm=instance, padding=reflect, upsample=deconv).forward (ResNet_arch.py:19-86) with
    ResnetBlock (:89-151): c7s1-64, two stride-2 convs, n_blocks reflect-padded residual blocks, two
    ConvTranspose2d(3, s2, p1, op1), c7s1-out, tanh.  InstanceNorm2d has no affine parameters and always
    uses the statistics of the instance (track_running_stats=False), also under eval()."""
    # norm_type 'batch' (the constructor's default, :39-49): nn.BatchNorm2d behind every conv but the last -- its parameters sit at the next index of
    # the Sequential -- and no bias on those convs (use_bias is True for InstanceNorm2d only); training=True: statistics of the batch given
    def conv(t, key, **kw):
        return F.conv2d(t, sd[key + ".weight"], sd.get(key + ".bias"), **kw)

    def inorm(t, key=None):
        if norm_type in ("IN", "instance"):
            return F.instance_norm(t, eps=eps)
        if training:
            return F.batch_norm(t, None, None, sd[key + ".weight"], sd[key + ".bias"], training=True, momentum=0.0, eps=eps)
        return F.batch_norm(t, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], training=False, momentum=0.0, eps=eps)

    t = F.relu(inorm(conv(F.pad(x, (3, 3, 3, 3), mode="reflect"), "model.1"), "model.2"))
    t = F.relu(inorm(conv(t, "model.4", stride=2, padding=1), "model.5"))
    t = F.relu(inorm(conv(t, "model.7", stride=2, padding=1), "model.8"))
    # ResnetBlock.conv_block (:118-146): [pad] conv norm relu [dropout: identity in eval mode] [pad] conv norm; 'zero' pads inside the conv
    pad_layer = padding_type != "zero"
    c1 = 1 if pad_layer else 0
    c2 = c1 + 3 + (1 if use_dropout else 0) + (1 if pad_layer else 0)
    mode = {"reflect": "reflect", "replicate": "replicate", "zero": "constant"}[padding_type]
    for i in range(10, 10 + n_blocks):
        r = F.relu(inorm(conv(F.pad(t, (1, 1, 1, 1), mode=mode), f"model.{i}.conv_block.{c1}"), f"model.{i}.conv_block.{c1 + 1}"))
        r = inorm(conv(F.pad(r, (1, 1, 1, 1), mode=mode), f"model.{i}.conv_block.{c2}"), f"model.{i}.conv_block.{c2 + 1}")
        t = t + r
    i = 10 + n_blocks
    for k in (i, i + 3):
        if f"model.{k}.1.weight" in sd:             # upsample_mode 'upconv' (:70-73): upconv_block = Upsample(nearest 2x), Conv2d(3x3) (block.py:348-361)
            t = conv(F.interpolate(t, scale_factor=2.0, mode="nearest"), f"model.{k}.1", padding=1)
        else:
            t = F.conv_transpose2d(t, sd[f"model.{k}.weight"], sd.get(f"model.{k}.bias"), stride=2, padding=1, output_padding=1)
        t = F.relu(inorm(t, f"model.{k + 1}"))
    return torch.tanh(conv(F.pad(t, (3, 3, 3, 3), mode="reflect"), f"model.{i + 7}"))


def wbcunet_forward(sd, x, mode="pt"):
    """UnetGeneratorWBC(mode).forward (WBCNet_arch.py:22-99) with ResBlock (:8-20): no norm layers,
    LeakyReLU(0.2), 2x upsampling + skip additions.  mode 'pt': zero-padded stride-2 convs, F.interpolate
    bilinear (align_corners=False); mode 'tf': tf_same_padding (:140-142) and tf_2xupsample_bilinear (:126-137)."""
    def conv(t, key, stride=1, pad=1):
        return F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=pad)
    lr = lambda t: F.leaky_relu(t, 0.2)

    def down(t, key):
        if mode == "tf":
            return conv(F.pad(t, (0, 1, 0, 1)), key, stride=2, pad=0)
        return conv(t, key, stride=2)

    def up(t):
        if mode != "tf":
            return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
        b, c, h, w = t.shape
        out = torch.zeros(b, c, h * 2, w * 2, dtype=t.dtype)
        out[:, :, ::2, ::2] = t
        p = F.pad(t, (0, 1, 0, 1), mode="replicate")
        out[:, :, 1::2, ::2] = (p[:, :, :-1, :-1] + p[:, :, 1:, :-1]) / 2
        out[:, :, ::2, 1::2] = (p[:, :, :-1, :-1] + p[:, :, :-1, 1:]) / 2
        out[:, :, 1::2, 1::2] = (p[:, :, :-1, :-1] + p[:, :, 1:, 1:]) / 2
        return out

    x0 = lr(conv(x, "conv", pad=3))
    x1 = lr(conv(lr(down(x0, "conv_1")), "conv_2"))
    x2 = lr(conv(lr(down(x1, "conv_3")), "conv_4"))
    for b in range(4):
        x2 = conv(lr(conv(x2, f"block_{b}.conv1")), f"block_{b}.conv2") + x2
    x2 = lr(conv(x2, "conv_5"))
    x3 = lr(conv(lr(conv(up(x2) + x1, "conv_6")), "conv_7"))
    x4 = lr(conv(up(x3) + x0, "conv_8"))
    return conv(x4, "conv_9", pad=3)
