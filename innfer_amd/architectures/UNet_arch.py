"""pix2pix UnetGenerator shell (reference UNet_arch.py:11-161): parameters and buffers carry the
reference's state-dict keys (queried from the engine), forward runs in libinnfer_amd.so (csrc/unet.hip).
BatchNorm follows the module's mode like nn.BatchNorm2d does: train() (how run.py runs pix2pix: meval=False, run.py:299-303)
normalises with the statistics of the current image, eval() (Model's default meval=True, run.py:96-97) with the running
statistics of the checkpoint."""
from .. import lib as L
from .param_module import ParamEngineModule


class UnetGenerator(ParamEngineModule):
    _api = 'unet'

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_type="batch", use_dropout=False,
                 upsample_mode="deconv"):
        super().__init__()
        if norm_type not in ('BN', 'batch') or use_dropout or upsample_mode != 'deconv':
            raise NotImplementedError('UnetGenerator: only norm=batch, no dropout, deconv is built on the HIP path')
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self._init_engine(input_nc, output_nc, num_downs, ngf)

    def _out_shape(self, N, H, W):
        return (N, self.output_nc, H, W)

    def forward(self, x):
        L.check(L.lib.innfer_unet_set_eval(self._handle, int(not self.training)))
        return super().forward(x)

    def flops(self, N, H, W):
        return L.lib.innfer_unet_flops(self._handle, N, H, W)
