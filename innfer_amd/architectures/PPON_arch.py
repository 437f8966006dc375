"""PPON shell (reference PPON_arch.py:12-76): parameters carry the reference's state-dict keys
(CFEM.0, CFEM.1.sub.<b>.RB<k>.{c1,d1..d8,c2}, CFEM.1.sub.<nb>, SFEM.<b>..., PFEM.<b>..., CRM/SRM/PRM.<i>),
forward runs in libinnfer_amd.so (csrc/ppon.hip) and returns the reference's 3-tuple
(out_c, out_s, out_p); run.py keeps the last element."""
import torch

from .. import lib as L
from .param_module import ParamEngineModule


class PPON(ParamEngineModule):
    _api = 'ppon'
    _has_fp32 = True         # float32 tensors: innfer_ppon_set_precision(1), the fp32 forward (csrc/f32ops.hip)
    _n_outputs = 3
    _accepts_out = False     # three results per forward: chop batches are copied into the tile buffer (run.py keeps the last)

    def _out_shape(self, N, H, W):
        return (N, self.out_nc, H * self.scale, W * self.scale)

    def __init__(self, in_nc=3, nf=64, nb=24, out_nc=3, upscale=4, act_type='lrelu', alpha=1.0):
        super().__init__()
        if str(act_type).lower() not in ('lrelu', 'leakyrelu'):
            raise NotImplementedError('PPON: only the LeakyReLU(0.2) activation is built')
        self.in_nc, self.out_nc, self.nf, self.nb, self.scale, self.alpha = in_nc, out_nc, nf, nb, upscale, float(alpha)
        self._init_engine(in_nc, out_nc, nf, nb, upscale, float(alpha))

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError('expected a 4D [N,C,H,W] tensor')
        if not x.is_cuda:
            raise RuntimeError('innfer_amd runs its forward on an MI355X only: there is no CPU path')
        self._check_dtype(x)
        with torch.cuda.device(x.device):        # the library allocates and launches on the process's current HIP device
            return self._forward_on_device(x)

    def _forward_on_device(self, x):
        self._claim_device(x.device)
        self._upload()
        L.check(L.lib.innfer_ppon_set_precision(self._handle, int(x.dtype == torch.float32)))      # the dtype IS the arithmetic (run.py:345,421-422)
        x = x.contiguous()
        N, _, H, W = x.shape
        s = self.scale
        outs = [torch.empty((N, self.out_nc, H * s, W * s), dtype=x.dtype, device=x.device) for _ in range(3)]
        need = L.lib.innfer_ppon_workspace_bytes(self._handle, N, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        dt = L.F16 if x.dtype == torch.float16 else L.F32
        L.check(L.lib.innfer_ppon_forward(self._handle, x.data_ptr(), dt, outs[0].data_ptr(), outs[1].data_ptr(),
                                          outs[2].data_ptr(), dt, N, H, W, self._ws.data_ptr(), self._ws.numel(),
                                          torch.cuda.current_stream(x.device).cuda_stream))
        return tuple(outs)
