"""pix2pix UnetGenerator shell (reference UNet_arch.py:11-161): parameters and buffers carry the
reference's state-dict keys (queried from the engine), forward runs in libinnfer_amd.so (csrc/unet.hip).
BatchNorm always uses the statistics of the current image -- that is how run.py runs pix2pix
(meval=False, run.py:299-303); an eval()-mode forward with running statistics is not built."""
import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import lib as L
from .engine_module import _Node


class UnetGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_type="batch", use_dropout=False,
                 upsample_mode="deconv"):
        super().__init__()
        if norm_type not in ('BN', 'batch') or use_dropout or upsample_mode != 'deconv':
            raise NotImplementedError('UnetGenerator: only norm=batch, no dropout, deconv is built on the HIP path')
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        h = C.c_void_p()
        L.check(L.lib.innfer_unet_create(C.byref(h), input_nc, output_nc, num_downs, ngf))
        self._handle = h
        self._keys = []
        key, nd, shp = C.create_string_buffer(256), C.c_int(), (C.c_int * 4)()
        for i in range(L.lib.innfer_unet_num_params(h)):
            L.check(L.lib.innfer_unet_param_info(h, i, key, 256, C.byref(nd), shp))
            k = key.value.decode()
            shape = tuple(shp[j] for j in range(nd.value))
            *path, leaf = k.split('.')
            node = self
            for name in path:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            if leaf == 'num_batches_tracked':
                node.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
            elif leaf.startswith('running_'):
                node.register_buffer(leaf, torch.ones(shape) if leaf == 'running_var' else torch.zeros(shape))
            else:
                node.register_parameter(leaf, nn.Parameter(torch.zeros(*shape), requires_grad=False))
            self._keys.append(k)
        self._version = None
        self._ws = None

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h is not None:
            try:
                L.lib.innfer_unet_destroy(h)
            except Exception:
                pass

    def _upload(self):
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if ver == self._version:
            return
        sd = self.state_dict()
        for i, k in enumerate(self._keys):
            if 'running_' in k or 'num_batches' in k:
                continue
            a = np.ascontiguousarray(sd[k].detach().float().cpu().numpy())
            L.check(L.lib.innfer_unet_set_param(self._handle, i, a.ctypes.data))
        self._version = ver

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError('expected a 4D [N,C,H,W] tensor')
        if not x.is_cuda:
            raise RuntimeError('innfer_amd runs its forward on an MI355X only: there is no CPU path')
        if x.dtype not in (torch.float16, torch.float32):
            raise TypeError(f'unsupported dtype {x.dtype}')
        self._upload()
        x = x.contiguous()
        N, _, H, W = x.shape
        out = torch.empty((N, self.output_nc, H, W), dtype=x.dtype, device=x.device)
        need = L.lib.innfer_unet_workspace_bytes(self._handle, N, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        dt = L.F16 if x.dtype == torch.float16 else L.F32
        L.check(L.lib.innfer_unet_forward(self._handle, x.data_ptr(), dt, out.data_ptr(), dt, N, H, W,
                                          self._ws.data_ptr(), self._ws.numel(),
                                          torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def flops(self, N, H, W):
        return L.lib.innfer_unet_flops(self._handle, N, H, W)
