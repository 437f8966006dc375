"""PAN shell (reference PAN_arch.py:109-222): parameters carry the reference's state-dict keys
(conv_first, SCPA_trunk.<b>.{conv1_a,conv1_b,k1.0,PACnv.k2,PACnv.k3,PACnv.k4,conv3}, trunk_conv,
FSA.{gamma,conv_f,conv_g,conv_h}, upsample.<i>, conv_last), forward runs in libinnfer_amd.so
(csrc/pan.hip).  Built: the configuration utils/defaults.py:78-89 produces (self-attention on,
single SCPA, nearest up-blocks) and the constructor's other values (self_attention, double_scpa, ups_inter_mode 'bilinear')."""
from .param_module import ParamEngineModule


class PAN(ParamEngineModule):
    _api = 'pan'
    _has_fp32 = True         # float32 tensors: innfer_pan_set_precision(1), the fp32 forward (csrc/f32ops.hip)

    def __init__(self, in_nc=3, out_nc=3, nf=40, unf=24, nb=16, scale=4, self_attention=True,
                 double_scpa=False, ups_inter_mode='nearest'):
        super().__init__()
        if ups_inter_mode not in ('nearest', 'bilinear'):
            raise NotImplementedError("PAN: ups_inter_mode 'nearest' and 'bilinear' are built")
        self.ups_inter_mode = ups_inter_mode
        self.in_nc, self.out_nc, self.nf, self.unf, self.nb, self.scale = in_nc, out_nc, nf, unf, nb, scale
        self.self_attention, self.double_scpa = bool(self_attention), bool(double_scpa)
        self._init_engine(in_nc, out_nc, nf, unf, nb, scale, int(self.self_attention), int(self.double_scpa), int(ups_inter_mode == 'bilinear'))

    def _fn(self, name):
        return super()._fn('create_ex' if name == 'create' else name)

    fused_scpa = True        # innfer_pan_set_fused_scpa: 1 / True = an SCPA block as one launch (csrc/pan_scpa.hip) + the FSA attention on the matrix cores; 0 / False: the
                             # five-launch blocks and the VALU attention of rounds 1-3; 2: the fused blocks with the VALU attention (A/B, parity tests); 3: no compact
                             # channel plane; 4: HRconv and conv_last as two launches (A/B of the fused tail)

    def _forward_on_device(self, x, out=None):
        from .. import lib as L
        L.check(L.lib.innfer_pan_set_fused_scpa(self._handle, int(self.fused_scpa)))
        return super()._forward_on_device(x, out)

    def _out_shape(self, N, H, W):
        return (N, self.out_nc, H * self.scale, W * self.scale)
