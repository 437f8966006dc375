"""Shell for the engines whose C API addresses parameters by state-dict key
(innfer_unet_*, innfer_pan_*): the module tree is built from the keys the engine
reports, so load_state_dict(strict=True) sees exactly the reference's names, and
forward hands device pointers to libinnfer_amd.so.  torch is plumbing only."""
import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import lib as L
from .engine_module import _Node, _result_tensor


class ParamEngineModule(nn.Module):
    _api = None                      # 'unet' | 'pan' -> innfer_<api>_* entry points

    def _fn(self, name):
        return getattr(L.lib, f'innfer_{self._api}_{name}')

    def _init_engine(self, *create_args):
        h = C.c_void_p()
        L.check(self._fn('create')(C.byref(h), *create_args))
        self._handle = h
        self._keys = []
        key, nd, shp = C.create_string_buffer(256), C.c_int(), (C.c_int * 4)()
        for i in range(self._fn('num_params')(h)):
            L.check(self._fn('param_info')(h, i, key, 256, C.byref(nd), shp))
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
        self._weights_device = None
        self._ws = None

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h is not None:
            try:
                self._fn('destroy')(h)
            except Exception:
                pass

    def invalidate_weights(self):
        self.__dict__.pop('_version_tensors', None)

    def _apply(self, fn, *args, **kwargs):          # .to / .cuda / .half rebind buffers: drop the cached walk of the module tree
        r = super()._apply(fn, *args, **kwargs)
        self.invalidate_weights()
        return r

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.invalidate_weights()
        return r

    def _upload(self):
        # parameters AND buffers: the running statistics of a BatchNorm are part of an eval()-mode forward
        ts = self.__dict__.get('_version_tensors')      # walked once; _apply / load_state_dict drop the cache
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__['_version_tensors'] = ts
        ver = tuple([(t.data_ptr(), t._version) for t in ts])
        if ver == self._version:
            return
        sd = self.state_dict()
        for i, k in enumerate(self._keys):
            if 'num_batches' in k:
                continue
            a = np.ascontiguousarray(sd[k].detach().float().cpu().numpy())
            L.check(self._fn('set_param')(self._handle, i, a.ctypes.data))
        self._version = ver

    def _out_shape(self, N, H, W):
        raise NotImplementedError

    def release_workspace(self):
        self._ws = None

    def _home_device(self, device=None):
        """Device for the calls that carry no tensor: the one asked for, else where the weights already are, else where the parameters are,
        else the current device (EngineModule._home_device's rule)."""
        if device is not None:
            device = torch.device(device)
            return device if device.index is not None else torch.device('cuda', torch.cuda.current_device())
        if self._weights_device is not None:
            return self._weights_device
        for p in self.parameters():
            if p.is_cuda:
                return p.device
            break
        return torch.device('cuda', torch.cuda.current_device())

    def _claim_device(self, device):
        """The engine's device allocations (packed weights, fp32 panels) are made on the process's current HIP device by the first call that needs
        them and stay there: record where, refuse another GPU (a size query used to upload on whatever device was current)."""
        if self._weights_device is not None and self._weights_device != device:
            raise NotImplementedError('this engine was first used on %s; build a second module for %s' % (self._weights_device, device))

    def _upload_on(self, device, fp32=None):
        """Upload (and, with fp32 given, select the precision) on `device`; the module is pinned to that GPU only once both succeeded (a failed
        upload -- out of memory inside a size query -- used to leave the module bound to a device that holds nothing)."""
        self._claim_device(device)
        self._upload()
        if fp32 is not None:
            L.check(self._fn('set_precision')(self._handle, int(fp32)))
        self._weights_device = device

    def tile_batch_bytes(self, b, ps, dtype=torch.float16, device=None):
        """Device bytes a forward of b tiles of ps x ps takes beyond the weights (workspace + input tiles + the output twice: the batch's result
        and its copy in the tile buffer the blend reads); parallel.engine_tile_cap sizes the chop batches with it and names the GPU the tiles are
        on.  Side effect: the engine is put into the precision `dtype` selects (innfer_<api>_set_precision -- its workspace differs per mode), which
        in the fp32 mode uploads the weights and builds their fp32 panels; that happens under torch.cuda.device(device), like the forward that follows."""
        import math
        out = self._out_shape(1, ps, ps)
        n_out = getattr(self, '_n_outputs', 1)
        in_nc = getattr(self, 'in_nc', None) or getattr(self, 'input_nc', 3)
        elt = 4 if dtype == torch.float32 else 2
        device = self._home_device(device)
        with torch.cuda.device(device):
            if self._has_fp32:
                self._upload_on(device, dtype == torch.float32)
            return self._fn('workspace_bytes')(self._handle, b, ps, ps) + b * (in_nc * ps * ps + (n_out + 1) * math.prod(out)) * elt

    _has_fp32 = False                # engines with an fp32 mode (innfer_<api>_set_precision): every shipped one (UNet, PAN, PPON, CycleGAN ResNet, WBC UNet)

    def _check_dtype(self, x):
        """The input's dtype is the arithmetic the caller asks for (the reference: model.half() / t_img.half(), run.py:345,383,421-422): float16
        tensors run the fp16 engine, float32 tensors the engine's fp32 mode (csrc/f32ops.hip).  Every shipped generator has both; the refusal below
        is for a future engine that is built fp16-only -- it must not serve fp16 accuracy under -no_fp16."""
        if x.dtype == torch.float32 and self._has_fp32:
            return                       # float32 tensors run the engine's fp32 mode (the reference's -no_fp16: run.py:345,421-422)
        if x.dtype == torch.float32:
            raise NotImplementedError(f"{type(self).__name__}: this engine has no fp32 mode; pass x.half() (the reference's default mode, run.py:345,421-422)")
        if x.dtype != torch.float16:
            raise TypeError(f'unsupported dtype {x.dtype}')

    _accepts_out = True              # forward(x, out=...): see EngineModule

    def forward(self, x, out=None):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError('expected a 4D [N,C,H,W] tensor')
        if not x.is_cuda:
            raise RuntimeError('innfer_amd runs its forward on an MI355X only: there is no CPU path')
        self._check_dtype(x)
        with torch.cuda.device(x.device):        # the library allocates and launches on the process's current HIP device
            return self._forward_on_device(x, out)

    def _forward_on_device(self, x, out=None):
        # (the input's dtype IS the arithmetic, as model.half() / t_img.half() are in the reference)
        self._upload_on(x.device, (x.dtype == torch.float32) if self._has_fp32 else None)
        x = x.contiguous()
        N, _, H, W = x.shape
        out = _result_tensor(out, self._out_shape(N, H, W), x)
        need = self._fn('workspace_bytes')(self._handle, N, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        dt = L.F16 if x.dtype == torch.float16 else L.F32
        L.check(self._fn('forward')(self._handle, x.data_ptr(), dt, out.data_ptr(), dt, N, H, W,
                                    self._ws.data_ptr(), self._ws.numel(),
                                    torch.cuda.current_stream(x.device).cuda_stream))
        return out
