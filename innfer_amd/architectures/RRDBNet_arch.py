"""RRDBNet (old-arch ESRGAN) shell.  Mirrors the constructor surface of the
reference's RRDBNet (RRDBNet_arch.py:16-48); the graph itself is built inside
libinnfer_amd.so (csrc/net.hip)."""
import ctypes as C

from .. import lib as L
from .engine_module import EngineModule
from .keys import rrdbnet_shapes


class RRDBNet(EngineModule):
    def __init__(self, in_nc, out_nc, nf, nb, nr=3, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', upsample_mode='upconv', convtype='Conv2D',
                 finalact=None, gaussian_noise=False, plus=False):
        unsupported = []
        if nr != 3: unsupported.append(f'nr={nr}')
        if norm_type: unsupported.append(f'norm_type={norm_type}')
        if act_type not in ('leakyrelu', 'lrelu'): unsupported.append(f'act_type={act_type}')
        if mode != 'CNA': unsupported.append(f'mode={mode}')
        if upsample_mode != 'upconv': unsupported.append(f'upsample_mode={upsample_mode}')
        if convtype != 'Conv2D': unsupported.append(f'convtype={convtype}')
        if finalact: unsupported.append(f'finalact={finalact}')
        if upscale == 3: unsupported.append('upscale=3')
        if unsupported:
            raise NotImplementedError('RRDBNet option(s) not built on the HIP path yet: ' + ', '.join(unsupported))
        # gaussian_noise: GaussianNoise is the identity in eval mode (block.py:382-388)
        super().__init__(rrdbnet_shapes(in_nc, out_nc, nf, nb, 32, upscale, plus))
        self.in_nc, self.out_nc, self.nf, self.nb, self.gc, self.upscale = in_nc, out_nc, nf, nb, 32, upscale
        self.plus = bool(plus)

    def _create_handle(self):
        h = C.c_void_p()
        L.check(L.lib.innfer_rrdbnet_create(C.byref(h), self.in_nc, self.out_nc, self.nf, self.nb,
                                            self.gc, self.upscale, int(self.plus)))
        return h
