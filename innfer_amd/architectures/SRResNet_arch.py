"""SRResNet / SRGAN shell (reference SRResNet_arch.py:15-46 with the defaults of
utils/defaults.py:53-67: no norm, ReLU, CNA, pixelshuffle, res_scale 1)."""
import ctypes as C

from .. import lib as L
from .engine_module import EngineModule
from .keys import srresnet_shapes
from .RRDBNet_arch import _FINAL_ACT


class SRResNet(EngineModule):
    def __init__(self, in_nc, out_nc, nf, nb, upscale=4, norm_type='batch', act_type='relu',
                 mode='NAC', res_scale=1, upsample_mode='upconv', convtype='Conv2D', finalact=None):
        unsupported = []
        if norm_type: unsupported.append(f'norm_type={norm_type}')
        if act_type != 'relu': unsupported.append(f'act_type={act_type}')
        if mode != 'CNA': unsupported.append(f'mode={mode}')
        if res_scale != 1: unsupported.append(f'res_scale={res_scale}')
        if upsample_mode != 'pixelshuffle': unsupported.append(f'upsample_mode={upsample_mode}')
        if convtype != 'Conv2D': unsupported.append(f'convtype={convtype}')
        if finalact and finalact.lower() not in _FINAL_ACT: unsupported.append(f'finalact={finalact}')
        if upscale == 3: unsupported.append('upscale=3')
        if unsupported:
            raise NotImplementedError('SRResNet option(s) not built on the HIP path yet: ' + ', '.join(unsupported))
        super().__init__(srresnet_shapes(in_nc, out_nc, nf, nb, upscale))
        self.in_nc, self.out_nc, self.nf, self.nb, self.upscale = in_nc, out_nc, nf, nb, upscale
        self.final_act = _FINAL_ACT[finalact.lower()] if finalact else 0

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_srresnet_create(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb, self.upscale))
        L.check(L.lib.innfer_net_set_final_act(h, self.final_act))
        return h
