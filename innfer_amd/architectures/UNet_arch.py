"""pix2pix UnetGenerator shell (reference UNet_arch.py:11-161): parameters and buffers carry the
reference's state-dict keys (queried from the engine), forward runs in libinnfer_amd.so (csrc/unet.hip).
BatchNorm follows the module's mode like nn.BatchNorm2d does: train() (how run.py runs pix2pix: meval=False, run.py:299-303)
normalises with the statistics of the current image, eval() (Model's default meval=True, run.py:96-97) with the running
statistics of the checkpoint."""
from .. import lib as L
from .param_module import ParamEngineModule


class UnetGenerator(ParamEngineModule):
    _api = 'unet'
    _has_fp32 = True         # float32 tensors: innfer_unet_set_precision(1), the fp32 forward (csrc/f32ops.hip)

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_type="batch", use_dropout=False,
                 upsample_mode="deconv"):
        super().__init__()
        if norm_type not in ('BN', 'batch', 'IN', 'instance'):           # the reference's own error (UNet_arch.py:42-43)
            raise NameError("Unknown norm layer")
        if upsample_mode not in ('deconv', 'upconv'):
            # the reference documents 'pixelshuffle' (UNet_arch.py:94) but builds no layer for it: its constructor dies on an unbound `upconv`
            raise NotImplementedError("UnetGenerator: upsample_mode is 'deconv' or 'upconv' (the reference builds no other)")
        self.upsample_mode = upsample_mode
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self.instance_norm = norm_type in ('IN', 'instance')
        # use_dropout: nn.Dropout(0.5) at the end of the ngf*8 blocks (UNet_arch.py:153-154) -- no parameters; the identity under eval()
        self.use_dropout = bool(use_dropout)
        self._init_engine(input_nc, output_nc, num_downs, ngf, int(self.instance_norm), int(upsample_mode == 'upconv'))

    def _fn(self, name):
        return super()._fn('create_ex' if name == 'create' else name)

    def _out_shape(self, N, H, W):
        return (N, self.output_nc, H, W)

    def forward(self, x, out=None):
        if self.use_dropout and self.training:
            raise NotImplementedError('UnetGenerator(use_dropout=True) in train mode draws random masks; the engine runs the eval-mode graph (net.eval())')
        L.check(L.lib.innfer_unet_set_eval(self._handle, int(not self.training)))
        return super().forward(x, out)

    def flops(self, N, H, W):
        return L.lib.innfer_unet_flops(self._handle, N, H, W)
