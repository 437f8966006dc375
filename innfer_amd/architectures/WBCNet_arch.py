"""White-box-Cartoonization UNet shell (reference WBCNet_arch.py:22-99): parameters carry the reference's
state-dict keys (conv, conv_1..conv_9, block_<b>.conv{1,2}), forward runs in libinnfer_amd.so
(csrc/wbcunet.hip).  Built: mode 'pt' (PyTorch padding / bilinear upsampling) and 'tf' (tf_same_padding,
tf_2xupsample_bilinear), nf 32, slope 0.2."""
from .param_module import ParamEngineModule


class UnetGeneratorWBC(ParamEngineModule):
    _api = 'wbc'
    _has_fp32 = True         # float32 tensors: innfer_wbc_set_precision(1), the fp32 forward (csrc/f32ops.hip)

    def __init__(self, nf=32, mode='pt', slope=0.2):
        super().__init__()
        if mode not in ('pt', 'tf') or abs(slope - 0.2) > 1e-12:
            raise NotImplementedError("UnetGeneratorWBC: modes 'pt' / 'tf' with slope 0.2 are built")
        self.nf, self.mode = nf, mode
        self._init_engine(nf, 1 if mode == 'tf' else 0)

    def _out_shape(self, N, H, W):
        return (N, 3, H, W)
