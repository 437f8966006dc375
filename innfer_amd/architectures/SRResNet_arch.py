"""SRResNet / SRGAN shell (reference SRResNet_arch.py:15-46 with the defaults of
utils/defaults.py:53-67: no norm, ReLU, CNA, pixelshuffle, res_scale 1)."""
import ctypes as C

from .. import lib as L
from .engine_module import EngineModule
from .keys import srresnet_shapes
from .RRDBNet_arch import _FINAL_ACT, _TRUNK_ACT


class SRResNet(EngineModule):
    def __init__(self, in_nc, out_nc, nf, nb, upscale=4, norm_type='batch', act_type='relu',
                 mode='NAC', res_scale=1, upsample_mode='upconv', convtype='Conv2D', finalact=None):
        if upsample_mode not in ('upconv', 'pixelshuffle'):      # the reference's own error (SRResNet_arch.py:33-34)
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        unsupported = []
        if norm_type: unsupported.append(f'norm_type={norm_type}')          # NAC blocks put BatchNorm + ReLU IN FRONT of the conv: nothing to fold
        if act_type not in _TRUNK_ACT: unsupported.append(f'act_type={act_type}')
        if mode != 'CNA': unsupported.append(f'mode={mode}')
        if convtype != 'Conv2D': unsupported.append(f'convtype={convtype}')
        if finalact and finalact.lower() not in _FINAL_ACT: unsupported.append(f'finalact={finalact}')
        if upscale == 3: unsupported.append('upscale=3')
        if unsupported:
            raise NotImplementedError('SRResNet option(s) not built on the HIP path yet: ' + ', '.join(unsupported))
        super().__init__(srresnet_shapes(in_nc, out_nc, nf, nb, upscale, upsample_mode))
        self.in_nc, self.out_nc, self.nf, self.nb, self.upscale = in_nc, out_nc, nf, nb, upscale
        self.trunk_act, self.res_scale, self.upconv_up = _TRUNK_ACT[act_type], float(res_scale), upsample_mode == 'upconv'
        self.final_act = _FINAL_ACT[finalact.lower()] if finalact else 0

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_srresnet_create_ex(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb, self.upscale,
                                                self.trunk_act, self.res_scale, int(self.upconv_up)))
        L.check(L.lib.innfer_net_set_final_act(h, self.final_act))
        return h
