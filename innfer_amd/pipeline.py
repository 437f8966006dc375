"""Image loop with the PCIe copies hidden behind the forward.

The reference's loop (run.py:404-442) is strictly serial per image: read -> np2tensor -> .half() ->
model -> tensor2np (.cpu() of an fp32 tensor) -> write.  Here the uint8 frame crosses PCIe in both
directions (4x / 2x fewer bytes than the reference's fp32 / fp16 tensors), through pinned staging
buffers, on two side streams: while frame i runs on the compute stream, frame i+1 is uploaded and
frame i-1 is downloaded.  Results are identical to `tensor2np(model(np2tensor(img).half()))` per frame.
"""
import numpy as np
import torch

from . import lib as L
from .utils.utils import _dt


class FramePipeline:
    """`for out in FramePipeline(model, scale)(frames)`: frames are uint8 HWC BGR numpy arrays of ONE
    shape; yields uint8 HWC BGR numpy arrays.  A yielded array is a view of a pinned buffer that stays untouched until the
    NEXT frame has been yielded (the pipeline owns depth + 1 output buffers: `depth` in flight and the one lent out) -- copy it if
    it must live longer than that."""

    def __init__(self, model, scale, device='cuda', half=True, normalize=False, depth=3, color_fix=False):
        self.model, self.scale, self.dev = model, scale, torch.device(device)
        self.half, self.normalize, self.depth, self.cf = half, normalize, max(2, depth), color_fix
        self.s_in, self.s_out = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self._slots = None
        self._free_out, self._lent = [], None

    def _alloc(self, shape):
        H, W, C = shape
        s = self.scale
        self._slots = [dict(h_in=torch.empty((H, W, C), dtype=torch.uint8).pin_memory(),
                            d_in=torch.empty((H, W, C), dtype=torch.uint8, device=self.dev),
                            d_out=torch.empty((H * s, W * s, C), dtype=torch.uint8, device=self.dev), h_out=None,
                            up=torch.cuda.Event(), done=torch.cuda.Event(), down=torch.cuda.Event(), busy=False)
                       for _ in range(self.depth)]
        self._free_out = [torch.empty((H * s, W * s, C), dtype=torch.uint8).pin_memory() for _ in range(self.depth + 1)]
        self._lent = None
        self._ws_cf = None

    def _compute(self, sl, shape):
        H, W, C = shape
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(sl['up'])
        s = self.scale
        if hasattr(self.model, 'run_u8'):            # run.Model: np2tensor / tensor2np fused into the tile gather / blend or the first / last conv
            self.model.run_u8(sl['d_in'], normalize=self.normalize, fp16=self.half, out=sl['d_out'])
        elif hasattr(self.model, 'forward_u8'):      # an RRDBNet / SRResNet module
            self.model.forward_u8(sl['d_in'], normalize=self.normalize, fp16=self.half, out=sl['d_out'])
        else:
            x = torch.empty((1, C, H, W), dtype=torch.float16 if self.half else torch.float32, device=self.dev)
            L.check(L.lib.innfer_u8hwc_to_nchw(sl['d_in'].data_ptr(), H, W, C, int(self.normalize), x.data_ptr(), _dt(x), cur.cuda_stream))
            y = self.model(x)
            L.check(L.lib.innfer_nchw_to_u8hwc(y.data_ptr(), _dt(y), H * s, W * s, C, int(self.normalize), sl['d_out'].data_ptr(), cur.cuda_stream))
        if self.cf:
            need = L.lib.innfer_color_fix_workspace_bytes(H, W, H * s, W * s, C)
            if self._ws_cf is None or self._ws_cf.numel() < need:
                self._ws_cf = torch.empty(need, dtype=torch.uint8, device=self.dev)
            fixed = torch.empty_like(sl['d_out'])
            L.check(L.lib.innfer_color_fix(sl['d_in'].data_ptr(), H, W, sl['d_out'].data_ptr(), H * s, W * s, C, fixed.data_ptr(),
                                           self._ws_cf.data_ptr(), self._ws_cf.numel(), cur.cuda_stream))
            sl['d_out'].copy_(fixed)
        sl['done'].record(cur)

    def __call__(self, frames):
        pending = []                                   # slots in flight, oldest first
        k = 0
        try:
            for img in frames:
                if not isinstance(img, np.ndarray) or img.dtype != np.uint8 or img.ndim != 3:
                    raise TypeError('FramePipeline: uint8 HWC numpy frames expected')
                if self._slots is None:
                    self._alloc(img.shape)
                sl = self._slots[k % self.depth]
                if sl['busy']:                             # the ring is full: hand out its oldest frame first
                    yield self._finish(pending.pop(0))
                if tuple(sl['h_in'].shape) != img.shape:
                    raise ValueError('FramePipeline: all frames must have the shape of the first one')
                sl['h_in'].numpy()[...] = img              # pageable -> pinned (host memcpy)
                with torch.cuda.stream(self.s_in):
                    sl['d_in'].copy_(sl['h_in'], non_blocking=True)
                    sl['up'].record(self.s_in)
                self._compute(sl, img.shape)
                sl['h_out'] = self._free_out.pop()         # never the buffer lent to the consumer: that one returns in _finish
                sl['busy'] = True
                pending.append(sl)                         # from here on the `finally` below owns the buffer: a raise in the copy cannot leak it
                k += 1
                with torch.cuda.stream(self.s_out):
                    self.s_out.wait_event(sl['done'])
                    sl['h_out'].copy_(sl['d_out'], non_blocking=True)
                    sl['down'].record(self.s_out)
            while pending:
                yield self._finish(pending.pop(0))
        finally:
            # the consumer may abandon the generator mid-stream (break, exception): wait for what is in flight and return every buffer to the
            # pool, so that the next call starts from a clean ring instead of failing on a drained one
            for sl in pending:
                sl['down'].synchronize()
                sl['busy'] = False
                if sl['h_out'] is not None:
                    self._free_out.append(sl['h_out'])
                    sl['h_out'] = None
            if self._lent is not None:
                self._free_out.append(self._lent)
                self._lent = None

    def _finish(self, sl):
        sl['down'].synchronize()
        sl['busy'] = False
        if self._lent is not None:                     # the consumer asked for another frame: the previous one's buffer is free again
            self._free_out.append(self._lent)
        self._lent, sl['h_out'] = sl['h_out'], None
        return self._lent.numpy()
