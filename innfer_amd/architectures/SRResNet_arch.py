"""SRResNet / SRGAN shell (reference SRResNet_arch.py:15-46; utils/defaults.py:53-67 builds it with no norm, ReLU, CNA, pixelshuffle,
res_scale 1; the class's own defaults are norm_type='batch', mode='NAC', upconv)."""
import ctypes as C

import numpy as np

from .. import lib as L
from .engine_module import EngineModule
from .keys import srresnet_layout, srresnet_shapes
from .RRDBNet_arch import _FINAL_ACT, _TRUNK_ACT


class SRResNet(EngineModule):
    def __init__(self, in_nc, out_nc, nf, nb, upscale=4, norm_type='batch', act_type='relu',
                 mode='NAC', res_scale=1, upsample_mode='upconv', convtype='Conv2D', finalact=None):
        if upsample_mode not in ('upconv', 'pixelshuffle'):      # the reference's own error (SRResNet_arch.py:33-34)
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        unsupported = []
        if norm_type and norm_type.lower() != 'batch': unsupported.append(f'norm_type={norm_type}')
        if str(act_type).lower() not in _TRUNK_ACT: unsupported.append(f'act_type={act_type}')
        if mode not in ('CNA', 'NAC', 'CNAC'): unsupported.append(f'mode={mode}')
        if convtype != 'Conv2D': unsupported.append(f'convtype={convtype}')
        if finalact and finalact.lower() not in _FINAL_ACT: unsupported.append(f'finalact={finalact}')
        if upscale == 3 and upsample_mode != 'upconv' and nf != 64: unsupported.append(f'upscale=3 with upsample_mode=pixelshuffle, nf={nf}')
        if unsupported:
            raise NotImplementedError('SRResNet option(s) not built on the HIP path yet: ' + ', '.join(unsupported))
        super().__init__(srresnet_shapes(in_nc, out_nc, nf, nb, upscale, upsample_mode, bool(norm_type), mode))
        self.norm, self.mode = bool(norm_type), mode
        self._layout = srresnet_layout(nb, self.norm, mode)          # engine key -> (conv key, BatchNorm in front, BatchNorm behind)
        self.in_nc, self.out_nc, self.nf, self.nb, self.upscale = in_nc, out_nc, nf, nb, upscale
        self.trunk_act, self.res_scale, self.upconv_up = _TRUNK_ACT[str(act_type).lower()], float(res_scale), upsample_mode == 'upconv'
        self.final_act = _FINAL_ACT[finalact.lower()] if finalact else 0

    # Eval-mode BatchNorm2d is a per-channel affine map.  BEHIND a conv (mode 'CNA' / 'CNAC'; and the norm in front of a NAC block's second conv, which
    # follows the first conv directly) it is folded into the conv's weights and bias at upload, float64 arithmetic.  IN FRONT of a conv whose input
    # is the residual stream (mode 'NAC': a block's first conv, LR_conv) it cannot be folded -- the zero padding is not mapped -- so the engine applies
    # act(alpha * x + shift) in an elementwise pass (innfer_net_set_conv_input_map).
    @staticmethod
    def _affine(sd, bk):
        g, beta = sd[bk + '.weight'].double().cpu().numpy(), sd[bk + '.bias'].double().cpu().numpy()
        mean, var = sd[bk + '.running_mean'].double().cpu().numpy(), sd[bk + '.running_var'].double().cpu().numpy()
        a = g / np.sqrt(var + 1e-5)
        return a, beta - mean * a

    def _param_key(self, engine_key):
        return self._layout[engine_key][0] if engine_key in self._layout else engine_key

    def _conv_tensors(self, k, sd):
        w, b = super()._conv_tensors(k, sd)
        post = next((v[2] for v in self._layout.values() if v[0] == k), None)
        if post is None:
            return w, b
        a, sh = self._affine(sd, post)
        w = (w.astype(np.float64) * a[:, None, None, None]).astype(np.float32)
        b = (0.0 if b is None else b.astype(np.float64)) * a + sh
        return w, b.astype(np.float32)

    def _ensure_engine(self):
        ver = self._weights_version()
        if self._handle is not None and ver == self._uploaded_version:
            return
        super()._ensure_engine()
        if self.mode != 'NAC':
            return
        key = C.create_string_buffer(128)
        K, Cc = C.c_int(), C.c_int()
        sd = self.state_dict()
        for i in range(L.lib.innfer_net_num_convs(self._handle)):
            L.check(L.lib.innfer_net_conv_info(self._handle, i, key, 128, C.byref(K), C.byref(Cc)))
            ek = key.value.decode()
            if ek not in self._layout:
                continue
            pre, lr = self._layout[ek][1], ek == f'model.1.sub.{self.nb}'
            act = 0 if lr else self.trunk_act          # LR_conv is built with act_type=None (SRResNet_arch.py:26)
            if ek.endswith('.res.2') or (pre is None and act == 0):
                continue                               # second conv of a block: its norm -> act is the first conv's epilogue
            if pre is None:
                L.check(L.lib.innfer_net_set_conv_input_map(self._handle, i, None, None, act))
            else:
                a, sh = self._affine(sd, pre)
                a, sh = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(sh, np.float32)
                L.check(L.lib.innfer_net_set_conv_input_map(self._handle, i, a.ctypes.data, sh.ctypes.data, act))

    def forward(self, x, outm=None, out=None):
        if self.norm and self.training:
            raise NotImplementedError("SRResNet(norm_type='batch') in train mode normalises with batch statistics; the engine runs the eval-mode BatchNorm (net.eval())")
        return super().forward(x, outm, out)

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_srresnet_create_ex(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb, self.upscale,
                                                self.trunk_act, self.res_scale, int(self.upconv_up)))
        L.check(L.lib.innfer_net_set_final_act(h, self.final_act))
        return h
