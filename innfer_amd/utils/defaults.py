"""Per-architecture default hyper-parameters: mirror of the reference's
utils/defaults.py:3-147 for the families on the HIP hot path.  Same input
contract (a kind string or a dict with 'type' / 'which_model_G'), same output keys,
so configs written for the reference resolve identically."""

_RRDB = ('rrdb_net', 'esrgan', 'evsrgan', 'esrgan-lite')
_SRRES = ('sr_resnet', 'srresnet', 'srgan')


def _pick(src, name, default):
    return src.pop(name, default)


def get_network_G_config(network_G, scale):
    scale = int(scale)
    if isinstance(network_G, str):
        kind, opts = network_G.lower(), {}
    elif isinstance(network_G, dict):
        opts = network_G
        name_key = 'which_model_G' if 'which_model_G' in opts else 'type'
        kind = opts.pop(name_key).lower()
    else:
        raise TypeError('network_G must be a str or a dict')

    cfg = {}
    if kind in _RRDB:
        lite = kind == 'esrgan-lite'
        cfg['type'] = 'rrdb_net'
        cfg['norm_type'] = _pick(opts, 'norm_type', None)
        cfg['mode'] = _pick(opts, 'mode', 'CNA')
        cfg['nf'] = _pick(opts, 'nf', 32 if lite else 64)
        cfg['nb'] = _pick(opts, 'nb', 12 if lite else 23)
        cfg['nr'] = _pick(opts, 'nr', 3)
        cfg['in_nc'] = _pick(opts, 'in_nc', 3)
        cfg['out_nc'] = _pick(opts, 'out_nc', 3)
        cfg['gc'] = _pick(opts, 'gc', 32)
        cfg['convtype'] = _pick(opts, 'convtype', 'Conv3D' if kind == 'evsrgan' else 'Conv2D')
        cfg['act_type'] = _pick(opts, 'net_act', None) or _pick(opts, 'act_type', 'leakyrelu')
        cfg['gaussian_noise'] = _pick(opts, 'gaussian', True)
        cfg['plus'] = _pick(opts, 'plus', False)
        cfg['finalact'] = _pick(opts, 'finalact', None)
        cfg['upscale'] = _pick(opts, 'scale', scale)
        cfg['upsample_mode'] = _pick(opts, 'upsample_mode', 'upconv')
    elif kind in _SRRES:
        cfg['type'] = 'sr_resnet'
        cfg['in_nc'] = _pick(opts, 'in_nc', 3)
        cfg['out_nc'] = _pick(opts, 'out_nc', 3)
        cfg['nf'] = _pick(opts, 'nf', 64)
        cfg['nb'] = _pick(opts, 'nb', 16)
        cfg['upscale'] = _pick(opts, 'scale', scale)
        cfg['norm_type'] = _pick(opts, 'norm_type', None)
        cfg['act_type'] = _pick(opts, 'net_act', None) or _pick(opts, 'act_type', 'relu')
        cfg['mode'] = _pick(opts, 'mode', 'CNA')
        cfg['upsample_mode'] = _pick(opts, 'upsample_mode', 'pixelshuffle')
        cfg['convtype'] = _pick(opts, 'convtype', 'Conv2D')
        cfg['finalact'] = _pick(opts, 'finalact', None)
        cfg['res_scale'] = _pick(opts, 'res_scale', 1)
    elif 'wbcunet' in kind:
        cfg['type'] = 'wbcunet_net'
        cfg['nf'] = _pick(opts, 'nf', 32)
        cfg['mode'] = 'tf' if 'tf' in kind else _pick(opts, 'mode', 'pt')
    elif 'unet' in kind or 'p2p' in kind:
        cfg['type'] = 'unet_net'
        cfg['input_nc'] = _pick(opts, 'in_nc', 3)
        cfg['output_nc'] = _pick(opts, 'out_nc', 3)
        cfg['num_downs'] = _pick(opts, 'num_downs', 7 if kind in ('unet_128', 'p2p_128') else 8)
        cfg['ngf'] = _pick(opts, 'ngf', 64)
        cfg['norm_type'] = _pick(opts, 'norm_type', 'batch')
        cfg['use_dropout'] = _pick(opts, 'use_dropout', False)
        cfg['upsample_mode'] = _pick(opts, 'upsample_mode', 'deconv')
    elif kind in ('pan_net', 'pan'):
        cfg['type'] = 'pan_net'
        cfg['in_nc'] = _pick(opts, 'in_nc', 3)
        cfg['out_nc'] = _pick(opts, 'out_nc', 3)
        cfg['nf'] = _pick(opts, 'nf', 40)
        cfg['unf'] = _pick(opts, 'unf', 24)
        cfg['nb'] = _pick(opts, 'nb', 16)
        cfg['scale'] = _pick(opts, 'scale', scale)
        cfg['self_attention'] = _pick(opts, 'self_attention', True)
        cfg['double_scpa'] = _pick(opts, 'double_scpa', False)
        cfg['ups_inter_mode'] = _pick(opts, 'ups_inter_mode', 'nearest')
    elif 'ppon' in kind:
        cfg['type'] = 'ppon'
        cfg['in_nc'] = _pick(opts, 'in_nc', 3)
        cfg['out_nc'] = _pick(opts, 'out_nc', 3)
        cfg['nf'] = _pick(opts, 'nf', 64)
        cfg['nb'] = _pick(opts, 'nb', 24)
        cfg['upscale'] = _pick(opts, 'scale', scale)
        cfg['act_type'] = _pick(opts, 'net_act', None) or _pick(opts, 'act_type', 'leakyrelu')
        cfg['alpha'] = _pick(opts, 'alpha', 1)
    elif ('resnet' in kind and kind != 'sr_resnet') or 'cg' in kind:
        cfg['type'] = 'resnet_net'
        cfg['input_nc'] = _pick(opts, 'in_nc', 3)
        cfg['output_nc'] = _pick(opts, 'out_nc', 3)
        cfg['n_blocks'] = _pick(opts, 'n_blocks', 6 if kind in ('resnet_6blocks', 'resnet_6', 'cg_6') else 9)
        cfg['ngf'] = _pick(opts, 'ngf', 64)
        cfg['norm_type'] = _pick(opts, 'norm_type', 'instance')
        cfg['use_dropout'] = _pick(opts, 'use_dropout', False)
        cfg['upsample_mode'] = _pick(opts, 'upsample_mode', 'deconv')
        cfg['padding_type'] = _pick(opts, 'padding_type', 'reflect')
    elif kind in ('mrrdb_net', 'mesrgan'):                 # modified ("new"-arch) ESRGAN (defaults.py:45-52)
        cfg['type'] = 'mrrdb_net'
        cfg['in_nc'] = _pick(opts, 'in_nc', 3)
        cfg['out_nc'] = _pick(opts, 'out_nc', 3)
        cfg['nf'] = _pick(opts, 'nf', 64)
        cfg['nb'] = _pick(opts, 'nb', 24)
        cfg['gc'] = _pick(opts, 'gc', 32)
    else:
        raise NotImplementedError(f'Generator model [{kind:s}] not recognized')

    if opts:            # the reference prints unprocessed options (defaults.py:145-146)
        print(opts)
    return cfg
