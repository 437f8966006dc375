"""CycleGAN ResnetGenerator shell (reference ResNet_arch.py:9-86): parameters carry the reference's
state-dict keys (model.1, model.4, model.7, model.<10+i>.conv_block.{1,5}, model.<10+n>, model.<13+n>,
model.<17+n>), forward runs in libinnfer_amd.so (csrc/resnet.hip).  Built: the configuration
utils/defaults.py:124-140 produces (instance norm, reflect padding, deconv upsampling, no dropout)."""
from .param_module import ParamEngineModule


class ResnetGenerator(ParamEngineModule):
    _api = 'resnet'

    def __init__(self, input_nc, output_nc, ngf=64, norm_type="instance", use_dropout=False, n_blocks=6,
                 padding_type='reflect', upsample_mode="deconv"):
        super().__init__()
        if norm_type not in ('IN', 'instance') or use_dropout or padding_type != 'reflect' or upsample_mode != 'deconv':
            raise NotImplementedError('ResnetGenerator: only norm=instance, padding=reflect, deconv, no dropout is built')
        self.input_nc, self.output_nc, self.ngf, self.n_blocks = input_nc, output_nc, ngf, n_blocks
        self._init_engine(input_nc, output_nc, ngf, n_blocks)

    def _out_shape(self, N, H, W):
        return (N, self.output_nc, H, W)
