"""CycleGAN ResnetGenerator shell (reference ResNet_arch.py:9-86): parameters carry the reference's
state-dict keys (model.1, model.4, model.7, model.<10+i>.conv_block.{1,5}, model.<10+n>, model.<13+n>,
model.<17+n>), forward runs in libinnfer_amd.so (csrc/resnet.hip).  Built: the configuration
utils/defaults.py:124-140 produces (instance norm, reflect padding, deconv upsampling, no dropout) and the constructor's other values
(norm_type 'batch' -- its default --, the paddings, dropout, upconv)."""
from .param_module import ParamEngineModule

_PADDING = {'reflect': 0, 'replicate': 1, 'zero': 2}          # innfer_resnet_create_ex codes


class ResnetGenerator(ParamEngineModule):
    _api = 'resnet'
    _has_fp32 = True         # float32 tensors: innfer_resnet_set_precision(1), the fp32 forward (csrc/f32ops.hip)

    def __init__(self, input_nc, output_nc, ngf=64, norm_type="batch", use_dropout=False, n_blocks=6,
                 padding_type='reflect', upsample_mode="deconv"):
        super().__init__()
        if padding_type not in _PADDING:                       # the reference's own error (ResNet_arch.py:128)
            raise NotImplementedError('padding {} is not implemented'.format(padding_type))
        if norm_type not in ('BN', 'batch', 'IN', 'instance'):  # the reference's own error (ResNet_arch.py:44)
            raise NameError("Unknown norm layer")
        if upsample_mode not in ('deconv', 'upconv'):
            raise NotImplementedError("ResnetGenerator: upsample_mode 'deconv' and 'upconv' are built (the reference builds no other)")
        self.batch_norm = norm_type in ('BN', 'batch')
        self.input_nc, self.output_nc, self.ngf, self.n_blocks = input_nc, output_nc, ngf, n_blocks
        self.padding_type, self.use_dropout = padding_type, bool(use_dropout)
        self.upsample_mode = upsample_mode
        self._init_engine(input_nc, output_nc, ngf, n_blocks, _PADDING[padding_type], int(self.use_dropout), int(upsample_mode == 'upconv'), int(self.batch_norm))

    def _fn(self, name):
        return super()._fn('create_ex' if name == 'create' else name)

    def forward(self, x, out=None):
        if self.use_dropout and self.training:
            raise NotImplementedError('ResnetGenerator(use_dropout=True) in train mode draws random masks; the engine runs the eval-mode graph (net.eval())')
        from .. import lib as L
        L.check(L.lib.innfer_resnet_set_eval(self._handle, int(not self.training)))       # BatchNorm follows the module's mode like nn.BatchNorm2d
        return super().forward(x, out)

    def _out_shape(self, N, H, W):
        return (N, self.output_nc, H, W)
