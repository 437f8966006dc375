"""Mirror of the reference's `Model` wrapper (run.py:23-225): .pth loading with
architecture / scale inference, and `Model.__call__` -> chop_forward.

Differences that are the point of this build:
  * the network forward, tile extraction and blend run in HIP (libinnfer_amd.so);
  * chop tiles are pushed through the network in BATCHES (the reference loops
    batch-1 and calls empty_cache() per tile, run.py:186-197); per-tile results
    are identical because tiles are independent.  tile_batch=None sizes the
    batches from the engine's workspace and the free memory (parallel.engine_tile_cap:
    <= 272 tiles per launch; an allocator OOM halves the batch and goes on);
  * with torch.distributed initialised, tiles can be sharded over ranks and
    gathered on rank 0 (parallel.py).
There is no CPU execution path: device must be a GPU.
"""
import torch

from .architectures import get_network
from .utils.defaults import get_network_G_config
from .utils.utils import extract_patches_2d, mod2normal, recompose_tensor, swa2normal

# families the HIP engine implements; the other branches of the reference's key
# sniffing are recognised and refused explicitly
_SNIFF = (
    ('SCPA_trunk.0.conv1_a.weight', 'pan'),
    ('model.1.sub.0.res.0.weight', 'srgan'),
    ('conv_first.weight', 'mesrgan'),
    ('model.0.weight', 'esrgan'),
    ('CFEM.0.weight', 'ppon'),
    ('conv_9.weight', 'wbcunet'),
)


def infer_from_state_dict(state_dict, scale=None, in_nc=3, out_nc=3):
    """Architecture, scale and hyper-parameters from a checkpoint's key names and
    shapes -- host logic of Model.load_model / infer_params (run.py:44-72,103-165).
    Returns dict(arch, scale, in_nc, out_nc, nf, nb, plus, net_params, state_dict)
    where state_dict has been SWA-unwrapped / converted to old-arch keys."""
    if 'n_averaged' in state_dict:
        state_dict = swa2normal(state_dict)
    for probe, arch in _SNIFF:
        if probe in state_dict:
            break
    else:
        raise Exception("Could not infer model parameters.")
    if arch == 'mesrgan':                    # new-arch checkpoints run as old-arch (run.py:57-61)
        state_dict = mod2normal(state_dict)
        arch = 'esrgan'
    if arch == 'pan':
        return _infer_pan(state_dict, scale, in_nc, out_nc)
    if arch == 'ppon':
        return _infer_ppon(state_dict, scale, in_nc, out_nc)
    if arch == 'wbcunet':                    # run.py:150-156: scale 1, mode 'pt', nf from the first conv
        cfg = {'type': 'wbcunet', 'mode': 'pt', 'nf': int(state_dict['conv.weight'].shape[0])}
        return dict(arch='wbcunet', scale=1, in_nc=3, out_nc=3, nf=cfg['nf'], nb=4, plus=False, state_dict=state_dict,
                    net_params=get_network_G_config(cfg, 1))
    if arch not in ('esrgan', 'srgan'):
        raise NotImplementedError(f"'{arch}' checkpoints are recognised but not on the HIP path yet")
    top = {}                                 # N -> out channels of 'model.N.weight|bias'
    nb = None
    n_2x = 0
    for key, val in state_dict.items():
        parts = key.split('.')
        if len(parts) == 5 and parts[2] == 'sub':
            nb = int(parts[3])               # the trunk conv sits right after the last block
        elif len(parts) == 3:
            n = int(parts[1])
            top.setdefault(n, val.shape[0])
            if n > 6 and parts[0] == 'model' and parts[2] == 'weight':
                n_2x += 1                    # every top-level conv past index 6 = one 2x stage
    w0 = state_dict['model.0.weight']
    info = dict(arch=arch, scale=2 ** n_2x, in_nc=int(w0.shape[1]), out_nc=int(top[max(top)]),
                nf=int(w0.shape[0]), nb=nb, plus=False, state_dict=state_dict)
    cfg = {'type': arch, 'in_nc': info['in_nc'], 'out_nc': info['out_nc'], 'nf': info['nf'], 'nb': nb}
    if arch == 'esrgan':
        info['plus'] = any('conv1x1' in k for k in state_dict)
        cfg['plus'] = info['plus']
    info['net_params'] = get_network_G_config(cfg, info['scale'])
    return info


def _infer_pan(state_dict, scale, in_nc, out_nc):
    """PAN checkpoints: the reference leaves "custom params inference TBD" and builds the defaults
    with the caller's scale / in_nc / out_nc (run.py:157-163).  Same here when a scale is given;
    without one (the reference would fail in PAN.__init__) it is read off the up-block keys."""
    if not scale:
        ups = {int(k.split('.')[1]) for k in state_dict if k.startswith('upsample.')}
        scale = 2 ** sum(1 for i in ups if i % 5 == 1)
    cfg = {'type': 'pan', 'in_nc': in_nc, 'out_nc': out_nc}
    return dict(arch='pan', scale=int(scale), in_nc=in_nc, out_nc=out_nc, nf=40, nb=16, plus=False,
                state_dict=state_dict, net_params=get_network_G_config(cfg, int(scale)))


def _infer_ppon(state_dict, scale, in_nc, out_nc):
    """PPON checkpoints: like PAN the reference builds the defaults with the caller's scale / in_nc / out_nc
    (run.py:157-163); without a scale it is read off the reconstruction head (one up-conv per 2x)."""
    if not scale:
        idx = sorted(int(k.split('.')[1]) for k in state_dict if k.startswith('CRM.') and k.endswith('.weight'))
        scale = 2 ** (len(idx) - 2)                      # the last two convs are HR_conv0 / HR_conv1
    cfg = {'type': 'ppon', 'in_nc': in_nc, 'out_nc': out_nc}
    return dict(arch='ppon', scale=int(scale), in_nc=in_nc, out_nc=out_nc, nf=64, nb=24, plus=False,
                state_dict=state_dict, net_params=get_network_G_config(cfg, int(scale)))


class Model:
    def __init__(self, model_path, arch=None, scale=None, in_nc=3, out_nc=3, device='cuda',
                 meval=True, strict=True, chop=True, tile_batch=None, state_dict=None):
        self.model_path = model_path
        self.arch = arch
        self.scale = scale
        self.in_nc = in_nc
        self.out_nc = out_nc
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("innfer_amd.Model needs device='cuda' (MI355X); the reference's -cpu mode "
                               "is not accelerated here and is not silently emulated")
        self.model = None
        self.eval = meval
        self.strict = strict
        self.chop = chop
        self.tile_batch = tile_batch
        self.load_model(state_dict)

    # ------------------------------------------------------------- loading
    def load_model(self, state_dict=None):
        if self.arch == 'ts':
            raise NotImplementedError('TorchScript models are opaque graphs and cannot run on the HIP engine')
        if state_dict is None:
            state_dict = torch.load(self.model_path, map_location='cpu')
        if self.arch == 'infer':
            info = infer_from_state_dict(state_dict, self.scale, self.in_nc, self.out_nc)
            state_dict = info['state_dict']
            self.arch, self.scale = info['arch'], info['scale']
            self.in_nc, self.out_nc = info['in_nc'], info['out_nc']
            net_params = info['net_params']
        else:
            if 'n_averaged' in state_dict:
                state_dict = swa2normal(state_dict)
            if not self.scale:
                self.scale = 1
            net_params = get_network_G_config({'type': self.arch}, self.scale)
        net = get_network(net_params)
        net.load_state_dict(state_dict, strict=self.strict)
        for p in net.parameters():
            p.requires_grad = False
        if self.eval:
            net.eval()
        self.model = net.to(self.device)

    def infer_params(self, state_dict):
        return infer_from_state_dict(state_dict)['net_params']

    # ------------------------------------------------------------- forward
    def get_torch_ctx(self):
        """The context a forward runs under (run.py:204-209): torch.no_grad() -- the TorchScript special case of the reference does not arise (no
        'ts' architecture here)."""
        return torch.no_grad()

    def chop_forward(self, data, patch_size=200, step=1.0, tile_range=None):
        """Tile, run, blend (run.py:167-202).  tile_range=(begin,count) runs a
        sub-range of tiles and returns the raw HR tiles instead of the blend."""
        _, _, H, W = data.shape
        patch_size = min(H, W, patch_size)
        tiles = extract_patches_2d(data, (patch_size, patch_size), [step, step], batch_first=True,
                                   tile_range=tile_range).squeeze(0)
        from .parallel import run_tile_batches
        with torch.no_grad():
            hr = run_tile_batches(self.model, tiles, self.tile_batch, pick=self._pick if self.arch == 'ppon' else None, out=self._tile_buffer(tiles))
        if tile_range is not None:
            return hr
        return recompose_tensor(hr, H, W, step=step, scale=self.scale)

    def _tile_buffer(self, tiles):
        """The [n, C', P, P] buffer the blend reads, allocated once so that every batch's result is written in place (no per-batch tensor + torch.cat):
        known for the engines that state their output shape; None (concatenate) otherwise."""
        m = self.model
        if self.arch == 'ppon' or not getattr(m, '_accepts_out', False):
            return None
        n, _, ps, _ = tiles.shape
        from .architectures.engine_module import EngineModule
        shape = m._out_shape(n, ps, ps, device=tiles.device) if isinstance(m, EngineModule) else m._out_shape(n, ps, ps)
        return torch.empty(tuple(shape), dtype=tiles.dtype, device=tiles.device)

    def _pick(self, y):
        """PPON returns (content, structure, perceptual) and run.py keeps the last (run.py:191-192,220-221)."""
        return y[2] if self.arch == 'ppon' else y

    def _predict(self, x):
        return self._pick(self.model(x))

    def __call__(self, data):
        if self.chop:
            return self.chop_forward(data, patch_size=200, step=0.5)
        with torch.no_grad():
            return self._predict(data)

    def run_u8(self, img, normalize=False, fp16=True, out=None):
        """Image in, image out: tensor2np(self(np2tensor(img, normalize)[.half()]), denormalize=normalize) (run.py:421-431) with the two
        conversions fused into the neighbouring kernels -- the tile gather / the blend on the chop path (innfer_extract_tiles_u8,
        innfer_recompose_u8), the first / last conv otherwise (EngineModule.forward_u8).  Bit-identical to the separate passes.
        img: uint8 HWC BGR(A), a numpy array (uploaded / downloaded as uint8) or a cuda tensor (stays on the GPU)."""
        import numpy as np
        from . import lib as L
        from .architectures.engine_module import EngineModule
        from .parallel import run_tile_batches
        from .utils import utils as U
        host = isinstance(img, np.ndarray)
        d = torch.from_numpy(np.ascontiguousarray(img)).to(self.device) if host else img.contiguous()
        if d.dtype != torch.uint8 or d.dim() != 3:
            raise TypeError('run_u8: expected a uint8 HWC image')
        H, W, Cc = d.shape
        s = int(self.scale or 1)
        dt = torch.float16 if fp16 else torch.float32
        code = L.F16 if fp16 else L.F32
        stream = torch.cuda.current_stream(d.device).cuda_stream
        fused_net = isinstance(self.model, EngineModule) and self.arch != 'ppon'
        with torch.no_grad(), torch.cuda.device(d.device):
            if self.chop:
                ps = min(H, W, 200)
                _, ys, xs = L.chop_plan(H, W, ps, 0.5)
                n = len(ys) * len(xs)
                tiles = torch.empty((n, Cc, ps, ps), dtype=dt, device=d.device)
                L.check(L.lib.innfer_extract_tiles_u8(d.data_ptr(), Cc, H, W, int(bool(normalize)), ps, 0.5, 0, n, tiles.data_ptr(), code, stream))
                hr = run_tile_batches(self.model, tiles, self.tile_batch, pick=self._pick if self.arch == 'ppon' else None, out=self._tile_buffer(tiles))
                Co, P = hr.shape[1], hr.shape[2]
                if out is None:
                    out = torch.empty((H * s, W * s, Co), dtype=torch.uint8, device=d.device)
                L.check(L.lib.innfer_recompose_u8(hr.data_ptr(), U._dt(hr), n, Co, P, H, W, 0.5, s, U._dt(hr), int(bool(normalize)),
                                                  out.data_ptr(), stream))
            elif fused_net:
                out = self.model.forward_u8(d, normalize=normalize, fp16=fp16, out=out)
            else:
                x = torch.empty((1, Cc, H, W), dtype=dt, device=d.device)
                L.check(L.lib.innfer_u8hwc_to_nchw(d.data_ptr(), H, W, Cc, int(bool(normalize)), x.data_ptr(), code, stream))
                y = self._predict(x).contiguous()
                if out is None:
                    out = torch.empty((y.shape[2], y.shape[3], y.shape[1]), dtype=torch.uint8, device=d.device)
                L.check(L.lib.innfer_nchw_to_u8hwc(y.data_ptr(), U._dt(y), y.shape[2], y.shape[3], y.shape[1], int(bool(normalize)), out.data_ptr(), stream))
        return out.cpu().numpy() if host else out


# ------------------------------------------------------------------- command line (run.py:225-445)
def parse_models(models_paths, scales_list=None):
    """`a+b` / `a>b` model chains and the per-model scale guessed from the file name (run.py:227-250)."""
    from .utils.utils import get_models_paths
    model_chain = models_paths.split("+") if "+" in models_paths else models_paths.split(">")
    try:
        all_models = get_models_paths("./models")
    except AssertionError:          # the reference insists on a ./models folder even for absolute paths; only the partial-name search needs it
        all_models = []
    full_chain = [check_model_path(m, all_models) for m in model_chain]
    if not scales_list:
        scales_list = [get_scale_name(m, None) for m in full_chain]
    elif len(scales_list) != len(model_chain):
        raise ValueError(f"The num. of scales {len(scales_list)} is != from number of models {len(model_chain)}")
    return full_chain, scales_list


def check_model_path(model_path, all_models=None):
    """Absolute path, ./models/<name>, or a unique partial-name match in ./models (run.py:253-274)."""
    import os.path as osp
    if osp.isfile(model_path):
        return model_path
    model_path_a = osp.join("models", model_path)
    if osp.isfile(model_path_a):
        return model_path_a
    if not all_models:
        raise ValueError(f"Model {model_path} not found.")
    m_list = [m for m in all_models if str(model_path.lower()) in str(m).lower()]
    if len(m_list) > 1:
        raise ValueError(f"Filter {model_path} returned multiple models: {m_list}.")
    if not m_list:
        raise ValueError(f"Model {model_path} not found.")
    return m_list[0]


def get_scale_name(model_path, scale=None):
    """The scale a model file announces in the first two characters of its name -- `4x_name.pth` -> 4, `x2net.pth` -> 2 -- or None; an explicit
    `scale` wins, with the reference's warning when the name disagrees (run.py:277-293)."""
    import os.path as osp
    import re
    head = osp.basename(model_path)[:2].lower()
    digits = head.replace('x', '')
    from_name = int(digits) if 'x' in head and re.fullmatch(r'[+-]?\d+', digits.strip()) else None
    if not scale:
        return from_name
    if from_name and from_name != scale:
        print(f"Warning: possible model scale mismatch on {model_path}")
    return scale


pix2pix_extras = {'meval': False, 'strict': True, 'normalize': True}       # run.py:299-303
cyglegan_extras = {'meval': True, 'strict': False, 'normalize': True}      # run.py:305-309
default_extras = {'meval': True, 'strict': True, 'normalize': False}       # run.py:311-315


def build_parser():
    """The reference's flags, names and destinations (run.py:320-331)."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('-models', '-m', type=str, required=True, help='Path to models.')
    parser.add_argument('-arch', '-a', type=str, required=False, default='infer', help='Model architecture.')
    parser.add_argument('-input', '-i', type=str, required=False, default='./input', help='Path to read input images.')
    parser.add_argument('-output', '-o', type=str, required=False, default='./output', help='Path to save output images.')
    parser.add_argument('-scale', '-s', type=str, required=False, default='-1', help='Model scaling factor.')
    parser.add_argument('-cf', required=False, action='store_true', help='Use color correction if enabled.')
    parser.add_argument('-comp', required=False, action='store_true', help='Save as comparison images if enabled.')
    parser.add_argument('-no_gpu', '-cpu', required=False, action='store_false', help='Run in CPU if enabled.')
    parser.add_argument('-no_fp16', required=False, action='store_false', help='Disable fp16 mode if needed.')
    parser.add_argument('-norm', required=False, action='store_true', help='Normalizes images in range [-1,1] if set, else [0,1].')
    return parser


def main(argv=None):
    """The image loop of the reference's command line (run.py:318-445) on the HIP engine: same flags, same per-architecture presets, same
    sequence read -> [linear_resize | modcrop] -> np2tensor -> model chain [-> guided filter] -> tensor2np [-> color_fix] -> save.  The
    uint8 image is what crosses PCIe in both directions; files go through OpenCV when it is installed and through PIL otherwise."""
    import os
    import os.path as osp
    import numpy as np
    from .utils import utils as U
    args = build_parser().parse_args(argv)
    if not args.no_gpu:
        raise RuntimeError("-cpu / -no_gpu: innfer_amd runs on an MI355X only; use the reference for a CPU run")
    if args.arch == 'ts':
        raise NotImplementedError('TorchScript models are opaque graphs and cannot run on the HIP engine')
    fp16 = args.no_fp16
    use_guided_filter = use_modcrop = False
    if 'unet_' in args.arch or 'p2p_' in args.arch:
        defaults, chop = pix2pix_extras, False
        resize = 512 if '512' in args.arch else 256 if '256' in args.arch else 128 if '128' in args.arch else False
    elif 'resnet_' in args.arch or 'cg_' in args.arch:
        defaults, chop, resize = cyglegan_extras, True, False
    elif 'wbc' in args.arch or 'wbc' in args.models:
        args.arch = "wbcunet_tf" if ('tf' in args.arch or 'tf' in args.models) else "wbcunet"
        defaults, chop, resize = pix2pix_extras, False, False
        use_guided_filter = use_modcrop = True
    else:
        defaults, resize, chop = default_extras, False, True
    meval, strict = defaults['meval'], defaults['strict']
    normalize = defaults['normalize'] or args.norm
    device = torch.device('cuda')
    scale = args.scale if args.scale != -1 else None        # (sic) the string '-1' never equals -1: the flag's value is what parse_models ignores
    del scale
    model_chain, scale_chain = parse_models(args.models)
    models = [Model(mc, args.arch, sc, device=device, meval=meval, strict=strict, chop=chop) for mc, sc in zip(model_chain, scale_chain)]
    if not fp16:
        # -no_fp16 = fp32 arithmetic on the GPU (run.py:345,421-422).  RRDBNet / SRResNet have an fp32-accurate engine, every other shipped generator an fp32
        # mode (float32 tensors select them).  The check stays for an engine built fp16-only: it must refuse, not hand out fp16 accuracy under the flag.
        from .architectures.engine_module import EngineModule
        for m in models:
            if not isinstance(m.model, EngineModule) and not getattr(m.model, '_has_fp32', False):
                raise NotImplementedError(f"-no_fp16: no fp32-accurate engine is built for '{m.arch}' ({type(m.model).__name__}); drop the flag to run its fp16 engine")
    images = U.get_images_paths(args.input)
    os.makedirs(args.output, exist_ok=True)
    # The loop is pipelined over the images: one thread decodes the next image file while the GPU works on this one, up to sixteen threads
    # encode and write finished images (PNG coding releases the GIL and is the slowest stage by far: profiles/r2/cli_pipeline.txt).  Everything that touches the GPU stays on this thread; outputs and messages are those
    # of the serial loop, in its order.
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    def save(img, img_out, path):
        if args.comp:
            U.save_img_comp([img, img_out], path)
        else:
            U.save_img(img_out, path)

    n_writers = max(2, min(16, os.cpu_count() or 2))
    reader, writer, pending = ThreadPoolExecutor(1), ThreadPoolExecutor(n_writers), deque()
    last_write = {}                 # output path -> future of the last write submitted for it

    def save_after(prev, img, img_out, path):
        # a/x.png and b/x.png (or x.png and x.jpg) share output/x.png: the reference's serial loop keeps the LAST one, so writes to one path are chained
        if prev is not None:
            prev.result()
        save(img, img_out, path)
    nxt = reader.submit(U.read_img, images[0]) if images else None
    try:
        for idx, image_path in enumerate(images):
            img_name = osp.splitext(osp.basename(image_path))[0]
            img = nxt.result()
            nxt = reader.submit(U.read_img, images[idx + 1]) if idx + 1 < len(images) else None
            if img is None:
                print(f'Error reading image {image_path}, skipping.')
                continue
            if resize:
                img = U.linear_resize(img, resize)
            if use_modcrop:
                img = U.modcrop(img, 4)
            if len(models) == 1 and not use_guided_filter and img.dtype == np.uint8 and img.ndim == 3:
                img_out = models[0].run_u8(img, normalize=normalize, fp16=fp16)          # conversions fused into the tile gather / blend / first and last conv
            else:
                t_img = U.np2tensor(img, normalize=normalize, device=device, dtype=torch.float16 if fp16 else torch.float32)
                t_out = t_img
                for mod in models:
                    t_out = mod(t_out)
                    if use_guided_filter:
                        t_out = U.guided_filter(t_img, t_out, r=1, eps=5e-3)
                img_out = U.tensor2np(t_out.detach(), denormalize=normalize)
            if args.cf:
                img_out = U.color_fix(img, img_out)
            out_path = osp.join(args.output, f'{img_name:s}.png')
            fut = writer.submit(save_after, last_write.get(out_path), img, img_out, out_path)
            last_write[out_path] = fut
            pending.append(fut)
            while len(pending) > n_writers:           # a bounded number of finished images wait for their files (an 8K RGB output is 100 MB)
                pending.popleft().result()
        while pending:
            pending.popleft().result()
    finally:
        reader.shutdown(wait=True)
        writer.shutdown(wait=True)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
