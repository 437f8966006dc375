"""RRDBNet (old-arch ESRGAN) shell.  Mirrors the constructor surface of the
reference's RRDBNet (RRDBNet_arch.py:16-48); the graph itself is built inside
libinnfer_amd.so (csrc/net.hip)."""
import ctypes as C

from .. import lib as L
from .engine_module import EngineModule
from .keys import mrrdb_key_of, mrrdbnet_shapes, rrdbnet_shapes


# `finalact` (block.py:81-101 act()) -> activation code of the last conv's epilogue
_FINAL_ACT = {'relu': 2, 'leakyrelu': 1, 'lrelu': 1, 'tanh': 3, 'sigmoid': 6}
# `act_type` of every conv block (block.py:81-90: LeakyReLU slope 0.2) -> activation code of the conv epilogues
_TRUNK_ACT = {'leakyrelu': 1, 'lrelu': 1, 'relu': 2}          # act() lower-cases its argument and takes both spellings (block.py:86-90)


class RRDBNet(EngineModule):
    def __init__(self, in_nc, out_nc, nf, nb, nr=3, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', upsample_mode='upconv', convtype='Conv2D',
                 finalact=None, gaussian_noise=False, plus=False):
        if upsample_mode not in ('upconv', 'pixelshuffle'):      # the reference's own error (RRDBNet_arch.py:32-33)
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        unsupported = []
        if not isinstance(nr, int) or nr < 1: unsupported.append(f'nr={nr}')
        if norm_type and norm_type.lower() != 'batch':
            unsupported.append(f'norm_type={norm_type}')          # built: BatchNorm2d behind the convs (folded at load); in front of LR_conv under 'NAC' (input map)
        if str(act_type).lower() not in _TRUNK_ACT: unsupported.append(f'act_type={act_type}')
        # mode: the dense blocks are built with mode='CNA' whatever is passed (RRDBNet_arch.py:27-29); only LR_conv takes it, and without
        # a norm layer or an activation a 'NAC' / 'CNAC' conv_block is the bare conv (block.py:237-254)
        if mode not in ('CNA', 'NAC', 'CNAC'): unsupported.append(f'mode={mode}')
        if upsample_mode == 'pixelshuffle' and upscale == 3 and nf != 64: unsupported.append(f'upsample_mode=pixelshuffle with upscale={upscale}, nf={nf}')
        if convtype != 'Conv2D': unsupported.append(f'convtype={convtype}')
        if finalact and finalact.lower() not in _FINAL_ACT: unsupported.append(f'finalact={finalact}')
        if unsupported:
            raise NotImplementedError('RRDBNet option(s) not built on the HIP path yet: ' + ', '.join(unsupported))
        # gaussian_noise: GaussianNoise is the identity in eval mode (block.py:382-388)
        super().__init__(rrdbnet_shapes(in_nc, out_nc, nf, nb, 32, upscale, plus, nr, upsample_mode, bool(norm_type), mode))
        self.norm = bool(norm_type)
        self.lr_norm_first = bool(norm_type) and mode == 'NAC'          # LR_conv = BatchNorm2d, conv: the norm cannot be folded (the zero padding is not mapped)
        self.in_nc, self.out_nc, self.nf, self.nb, self.gc, self.upscale = in_nc, out_nc, nf, nb, 32, upscale
        self.plus, self.nr = bool(plus), nr
        self.trunk_act = _TRUNK_ACT[str(act_type).lower()]
        self.pixelshuffle_up = upsample_mode == 'pixelshuffle'
        self.final_act = _FINAL_ACT[finalact.lower()] if finalact else 0

    # norm_type='batch' (RRDBNet_arch.py:27-29, block.py:244-246): every conv of the dense blocks and LR_conv is followed by a BatchNorm2d.  Under
    # eval() -- Model's default (run.py:96-97) -- that is a per-channel affine map of the conv's output, so it is folded into the conv's weights and
    # bias when they are uploaded (float64 arithmetic): y = (conv(x) + b - mean) * g / sqrt(var + eps) + beta.
    def _bn_key(self, k):
        if not self.norm:
            return None
        if k.endswith('.0') and '.conv' in k:
            return k[:-2] + '.1'
        return f'model.1.sub.{self.nb + 1}' if k == f'model.1.sub.{self.nb}' and not self.lr_norm_first else None

    def _param_key(self, engine_key):
        if self.lr_norm_first and engine_key == f'model.1.sub.{self.nb}':
            return f'model.1.sub.{self.nb + 1}'
        return engine_key

    def _ensure_engine(self):
        ver = self._weights_version()
        if self._handle is not None and ver == self._uploaded_version:
            return
        super()._ensure_engine()
        if not self.lr_norm_first:
            return
        import numpy as np
        sd, bk = self.state_dict(), f'model.1.sub.{self.nb}'
        a = sd[bk + '.weight'].double().cpu().numpy() / np.sqrt(sd[bk + '.running_var'].double().cpu().numpy() + 1e-5)
        sh = sd[bk + '.bias'].double().cpu().numpy() - sd[bk + '.running_mean'].double().cpu().numpy() * a
        a, sh = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(sh, np.float32)
        key = C.create_string_buffer(128)
        K, Cc = C.c_int(), C.c_int()
        for i in range(L.lib.innfer_net_num_convs(self._handle)):
            L.check(L.lib.innfer_net_conv_info(self._handle, i, key, 128, C.byref(K), C.byref(Cc)))
            if key.value.decode() == bk:
                L.check(L.lib.innfer_net_set_conv_input_map(self._handle, i, a.ctypes.data, sh.ctypes.data, 0))

    def _conv_tensors(self, k, sd):
        w, b = super()._conv_tensors(k, sd)
        bk = self._bn_key(k)
        if bk is None:
            return w, b
        import numpy as np
        g, beta = sd[bk + '.weight'].double().cpu().numpy(), sd[bk + '.bias'].double().cpu().numpy()
        mean, var = sd[bk + '.running_mean'].double().cpu().numpy(), sd[bk + '.running_var'].double().cpu().numpy()
        a = g / np.sqrt(var + 1e-5)
        w = (w.astype(np.float64) * a[:, None, None, None]).astype(np.float32)
        b = ((0.0 if b is None else b.astype(np.float64)) - mean) * a + beta
        return w, b.astype(np.float32)

    def forward(self, x, outm=None, out=None):
        if self.norm and self.training:
            raise NotImplementedError("RRDBNet(norm_type='batch') in train mode normalises with batch statistics; the engine folds the eval-mode BatchNorm (net.eval())")
        return super().forward(x, outm, out)

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_rrdbnet_create_ex(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb, self.gc, self.upscale,
                                               int(self.plus), self.nr, self.trunk_act, int(self.pixelshuffle_up)))
        L.check(L.lib.innfer_net_set_final_act(h, self.final_act))
        return h



class MRRDBNet(EngineModule):
    """Modified ("new"-arch) ESRGAN (reference RRDBNet_arch.py:173-231): conv_first, RRDB_trunk.<b>.RDB<r>.conv<i>,
    trunk_conv, upconv1/2 behind nearest 2x, HRconv, conv_last -- the 4x RRDBNet graph under other parameter names,
    so it runs on the same engine (csrc/net.hip); only the state-dict keys differ."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32):
        super().__init__(mrrdbnet_shapes(in_nc, out_nc, nf, nb, 32))
        if gc != 32:
            raise NotImplementedError(f'MRRDBNet: gc={gc} is not built on the HIP path (gc=32 only)')
        self.in_nc, self.out_nc, self.nf, self.nb, self.gc, self.upscale = in_nc, out_nc, nf, nb, 32, 4

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_rrdbnet_create(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb, self.gc, 4, 0))
        return h

    def _param_key(self, engine_key):
        return mrrdb_key_of(engine_key, self.nb)
