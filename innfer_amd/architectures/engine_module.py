"""nn.Module shell whose parameters carry the reference's state-dict keys and whose
forward runs entirely in libinnfer_amd.so (HIP, gfx950).

PyTorch here is plumbing: parameter storage, load_state_dict(strict=True), device
memory for input / output / workspace and the current HIP stream.  No arithmetic
of the forward pass is executed by torch.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import lib as L


class _Node(nn.Module):
    """Anonymous container used to spell dotted key paths ("model.1.sub.0...")."""


def _result_tensor(out, shape, x):
    """The tensor a forward writes: a fresh one, or the caller's `out` when it is exactly what a fresh one would be."""
    if out is None:
        return torch.empty(shape, dtype=x.dtype, device=x.device)
    if tuple(out.shape) != tuple(shape) or out.dtype != x.dtype or out.device != x.device or not out.is_contiguous():
        raise ValueError(f'out= must be a contiguous {tuple(shape)} {x.dtype} tensor on {x.device}; got {tuple(out.shape)} {out.dtype} on {out.device}'
                         f'{"" if out.is_contiguous() else ", not contiguous"}')
    return out


class EngineModule(nn.Module):
    def __init__(self, shapes):
        super().__init__()
        for key, shape in shapes.items():
            *path, leaf = key.split('.')
            node = self
            for name in path:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            if leaf == 'num_batches_tracked':               # BatchNorm2d state (RRDBNet(norm_type='batch')): buffers, as in nn.BatchNorm2d
                node.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
            elif leaf.startswith('running_'):
                node.register_buffer(leaf, torch.ones(shape) if leaf == 'running_var' else torch.zeros(shape))
            else:
                node.register_parameter(leaf, nn.Parameter(torch.zeros(*shape), requires_grad=False))
        self._handle = None
        self._uploaded_version = None
        self._weights_device = None          # GPU the engine's packed weights live on
        self._ws = None
        self.band_rows = 0
        self.upconv_phases = True    # innfer_net_set_upconv_phases: upconv_block convs as four 2x2-tap phases on the LR grid (summed weights, one rounding); True / 1: all four
                                     # phases in one visit of a tile, 2: one phase per visit (rounds 3's form; same bits), False / 0: nine taps on the HR grid
        self.hr_chain = True         # innfer_net_set_hr_chain: the last upconv_block -> HR_conv0 -> conv_last as ONE kernel chained through LDS (bit-identical to the launches it replaces)
        self.fused_tail = True       # innfer_net_set_fused_tail: HR_conv0 -> conv_last as one kernel where the shapes allow it (results agree to the last fp16 rounding with the two-launch form)
        self.residual_lds = 1        # innfer_net_set_residual_lds: the dense block's `x5 * 0.2 + x` takes x from the conv's own staged LDS tiles -- 1 the RRDB-end blocks (measured gain), 2 every block, 0 never (all agree to the last fp16 rounding)

    # ---- subclasses provide the C handle ------------------------------------
    def _create_handle(self):
        raise NotImplementedError

    def _param_key(self, engine_key):
        """State-dict prefix of the conv the engine calls `engine_key` (old-arch names); identity by default."""
        return engine_key

    # ---- weight upload (load time, not forward time) -------------------------
    def _weights_version(self):
        # (data pointer, in-place version) of every tensor the engine was loaded from.  The walk over the module tree -- ~700 parameters for
        # RRDBNet-23 -- is cached; everything that REBINDS tensor objects drops the cache: nn.Module._apply (.to / .cuda / .half replace buffers,
        # and parameters under torch.__future__.set_overwrite_module_params_on_conversion) and load_state_dict (assign=True).
        ts = self.__dict__.get('_version_tensors')
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__['_version_tensors'] = ts
        return tuple([(t.data_ptr(), t._version) for t in ts])

    def invalidate_weights(self):
        """Forget which tensors the engine was loaded from: the next forward walks the module tree again and re-uploads if anything differs.
        Call it after replacing a parameter / buffer OBJECT of a sub-module by hand (setattr, register_buffer)."""
        self.__dict__.pop('_version_tensors', None)

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self.invalidate_weights()
        return r

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.invalidate_weights()
        return r

    def _ensure_engine(self):
        ver = self._weights_version()
        if self._handle is not None and ver == self._uploaded_version:
            return
        if self._handle is None:
            self._handle = self._create_handle()
        sd = self.state_dict()
        n = L.lib.innfer_net_num_convs(self._handle)
        key = C.create_string_buffer(128)
        K, Cc = C.c_int(), C.c_int()
        for i in range(n):
            L.check(L.lib.innfer_net_conv_info(self._handle, i, key, 128, C.byref(K), C.byref(Cc)))
            k = self._param_key(key.value.decode())
            w, b = self._conv_tensors(k, sd)
            w = np.ascontiguousarray(w)
            if w.shape[:2] != (K.value, Cc.value):
                raise RuntimeError(f'size mismatch for {k}.weight: {tuple(w.shape)} vs engine ({K.value},{Cc.value},3,3)')
            bp = None
            if b is not None:
                b = np.ascontiguousarray(b)
                bp = b.ctypes.data
            L.check(L.lib.innfer_net_set_conv(self._handle, i, w.ctypes.data, bp))
        self._uploaded_version = ver

    def _conv_tensors(self, k, sd):
        """fp32 (weight, bias or None) the engine's conv `k` is loaded with; subclasses fold what follows the conv into them."""
        w = sd[k + '.weight'].detach().float().cpu().numpy()
        b = sd.get(k + '.bias')
        return w, (None if b is None else b.detach().float().cpu().numpy())

    def _destroy_handle(self):
        h = getattr(self, '_handle', None)
        if h is not None:
            try:
                L.lib.innfer_net_destroy(h)
            except Exception:
                pass
        self._handle, self._uploaded_version, self._ws = None, None, None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:          # interpreter shutdown: torch's module machinery may already be torn down
            pass

    # ---- forward ---------------------------------------------------------------
    _OUTM = {None: 0, 'scaltanh': 1, 'tanh': 2, 'sigmoid': 3, 'clamp': 4}

    _accepts_out = True              # forward(x, out=...) writes its result into a caller's tensor (parallel.run_tile_batches lands chop batches in the tile buffer)

    def forward(self, x, outm=None, out=None):
        """outm: the range limiter of RRDBNet.forward / SRResNet.forward (RRDBNet_arch.py:50-62); any other value means none, as there.
        out: optional contiguous [N, out_nc, s*H, s*W] tensor of x's dtype on x's device that receives the result (and is returned) instead of a fresh one.
        The input's dtype selects the arithmetic, as `model.half()` / `t_img.half()` do in the reference (run.py:345,383,421-422): a float16 tensor
        runs the fp16 engine, a float32 tensor the fp32-accurate one (innfer_net_set_precision: <= 1e-4 against the fp32 reference, 3x the MFMA work)."""
        self._outm = self._OUTM.get(outm, 0)
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError('expected a 4D [N,C,H,W] tensor')
        if not x.is_cuda:
            raise RuntimeError(
                'innfer_amd runs its forward on an MI355X only: there is no CPU path '
                '(the CPU restatement lives in oracle/ and is test infrastructure).')
        if x.dtype not in (torch.float16, torch.float32):
            raise TypeError(f'unsupported dtype {x.dtype}')
        # the library allocates and launches on the process's CURRENT HIP device: make that the input's for the whole call
        # (weights packed for another GPU are re-uploaded)
        with torch.cuda.device(x.device):
            return self._forward_on_device(x, out)

    def _engine_on(self, device):
        """The engine with its packed weights on `device`.  Call with `device` as the process's current HIP device (torch.cuda.device): the
        library allocates on the current device, so an engine whose weights live on another GPU is destroyed and rebuilt here.  EVERY path that
        may create the handle goes through this (forward, forward_u8, tile_batch_bytes, flops) -- a handle created outside it would put the
        weights on whatever device happened to be current and the first forward elsewhere would launch with foreign pointers."""
        device = torch.device(device)
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._weights_device is not None and self._weights_device != device:
            self._destroy_handle()
        self._ensure_engine()
        self._weights_device = device

    def _home_device(self, device=None):
        """Device for the calls that carry no tensor: the one asked for, else where the weights already are, else where the parameters are,
        else the current device."""
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

    def _forward_on_device(self, x, out=None):
        self._engine_on(x.device)
        L.check(L.lib.innfer_net_set_band_rows(self._handle, int(self.band_rows)))
        L.check(L.lib.innfer_net_set_fused_tail(self._handle, int(bool(self.fused_tail))))
        L.check(L.lib.innfer_net_set_hr_chain(self._handle, int(bool(self.hr_chain))))
        L.check(L.lib.innfer_net_set_residual_lds(self._handle, int(self.residual_lds)))
        L.check(L.lib.innfer_net_set_upconv_phases(self._handle, int(self.upconv_phases)))
        L.check(L.lib.innfer_net_set_outm(self._handle, int(getattr(self, '_outm', 0))))
        L.check(L.lib.innfer_net_set_precision(self._handle, int(x.dtype == torch.float32)))
        x = x.contiguous()
        N, _, H, W = x.shape
        s = L.lib.innfer_net_scale(self._handle)
        out = _result_tensor(out, (N, self.out_nc, H * s, W * s), x)
        need = L.lib.innfer_net_workspace_bytes(self._handle, N, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        dt = L.F16 if x.dtype == torch.float16 else L.F32
        stream = torch.cuda.current_stream(x.device).cuda_stream
        L.check(L.lib.innfer_net_forward(self._handle, x.data_ptr(), dt, out.data_ptr(), dt, N, H, W,
                                         self._ws.data_ptr(), self._ws.numel(), stream))
        return out

    def forward_u8(self, img, normalize=False, fp16=True, out=None):
        """uint8 HWC BGR(A) image(s) on the GPU -> uint8 HWC BGR(A) image(s): np2tensor is the first conv's prologue and tensor2np the last conv's
        epilogue (innfer_net_forward with INNFER_U8 at both ends).  Equals tensor2np(self(np2tensor(img)[.half()]), denormalize=normalize) bit for
        bit.  img: [H,W,C] or [N,H,W,C] uint8 cuda tensor; returns the same rank."""
        if not isinstance(img, torch.Tensor) or img.dtype != torch.uint8 or img.dim() not in (3, 4):
            raise ValueError('forward_u8: expected a uint8 [H,W,C] or [N,H,W,C] tensor')
        if not img.is_cuda:
            raise RuntimeError('innfer_amd runs its forward on an MI355X only: there is no CPU path')
        with torch.cuda.device(img.device):
            self._engine_on(img.device)
            L.check(L.lib.innfer_net_set_band_rows(self._handle, int(self.band_rows)))
            L.check(L.lib.innfer_net_set_fused_tail(self._handle, int(bool(self.fused_tail))))
            L.check(L.lib.innfer_net_set_hr_chain(self._handle, int(bool(self.hr_chain))))
            L.check(L.lib.innfer_net_set_residual_lds(self._handle, int(self.residual_lds)))
            L.check(L.lib.innfer_net_set_upconv_phases(self._handle, int(self.upconv_phases)))
            L.check(L.lib.innfer_net_set_u8_io(self._handle, int(bool(normalize)), int(bool(fp16))))
            L.check(L.lib.innfer_net_set_precision(self._handle, int(not fp16)))
            L.check(L.lib.innfer_net_set_outm(self._handle, 0))
            x = img.contiguous()
            batched = x.dim() == 4
            N, (H, W, Cc) = (x.shape[0] if batched else 1), x.shape[-3:]
            if Cc != self.in_nc:
                raise ValueError(f'forward_u8: the image has {Cc} channels, the network takes {self.in_nc}')
            s = L.lib.innfer_net_scale(self._handle)
            shape = (N, H * s, W * s, self.out_nc) if batched else (H * s, W * s, self.out_nc)
            if out is None:
                out = torch.empty(shape, dtype=torch.uint8, device=x.device)
            elif tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_contiguous():
                raise ValueError(f'forward_u8: out must be a contiguous uint8 tensor of shape {shape}')
            need = L.lib.innfer_net_workspace_bytes(self._handle, N, H, W)
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            L.check(L.lib.innfer_net_forward(self._handle, x.data_ptr(), L.U8, out.data_ptr(), L.U8, N, H, W,
                                             self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def release_workspace(self):
        self._ws = None

    def tile_batch_bytes(self, b, ps, dtype=torch.float16, device=None):
        """Device bytes a forward of b tiles of ps x ps takes beyond the weights: the engine workspace (innfer_net_workspace_bytes in the precision
        `dtype` selects -- twice as much in the fp32-accurate mode) + the input tiles + the HR tiles twice (the batch's result and its copy in the
        tile buffer the blend reads).  parallel.engine_tile_cap sizes the chop batches with it and names the GPU the tiles are on: the engine
        (packed weights included) is built THERE, as the forward that follows would."""
        device = self._home_device(device)
        with torch.cuda.device(device):
            self._engine_on(device)
            L.check(L.lib.innfer_net_set_precision(self._handle, int(dtype == torch.float32)))
            elt = 4 if dtype == torch.float32 else 2
            s = L.lib.innfer_net_scale(self._handle)
            return L.lib.innfer_net_workspace_bytes(self._handle, b, ps, ps) + b * (self.in_nc * ps * ps + 2 * self.out_nc * (ps * s) ** 2) * elt

    def _out_shape(self, N, H, W, device=None):
        """Shape of forward's result for an [N, in_nc, H, W] input (the scale is the engine's)."""
        device = self._home_device(device)
        with torch.cuda.device(device):
            self._engine_on(device)
            s = L.lib.innfer_net_scale(self._handle)
        return (N, self.out_nc, H * s, W * s)

    def flops(self, N, H, W, device=None):
        device = self._home_device(device)
        with torch.cuda.device(device):
            self._engine_on(device)
            return L.lib.innfer_net_flops(self._handle, N, H, W)
