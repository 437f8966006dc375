"""Generator factory: string -> nn.Module, mirror of the reference's
architectures.get_network (architectures/__init__.py:5-40) for the hot-path
families.  Modules are imported lazily so that `architectures.keys` stays
importable without torch or the HIP library."""


def get_network(opt_net):
    """Instantiate a generator from a config dict made by
    utils.defaults.get_network_G_config (same contract as the reference:
    'type' is popped, the rest are constructor keyword arguments)."""
    kind = opt_net.pop('type').lower()
    if kind == 'rrdb_net':
        from .RRDBNet_arch import RRDBNet as net
    elif kind == 'sr_resnet':
        from .SRResNet_arch import SRResNet as net
    elif kind == 'unet_net':
        from .UNet_arch import UnetGenerator as net
    elif kind == 'pan_net':
        from .PAN_arch import PAN as net
    elif kind == 'ppon':
        from .PPON_arch import PPON as net
    elif kind == 'resnet_net':
        from .ResNet_arch import ResnetGenerator as net
    elif kind == 'wbcunet_net':
        from .WBCNet_arch import UnetGeneratorWBC as net
    elif kind == 'mrrdb_net':
        from .RRDBNet_arch import MRRDBNet as net
    else:
        raise NotImplementedError('Model [{:s}] not recognized'.format(kind))
    return net(**opt_net)
