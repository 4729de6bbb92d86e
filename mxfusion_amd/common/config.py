"""Global defaults, mirroring mxfusion/common/config.py:18-52 (users mutate DEFAULT_DTYPE, e.g.
examples/notebooks/gp_regression.ipynb:115).  The device is always an MI355X; there is no CPU context."""
import torch

DEFAULT_DTYPE = 'float32'
MXNET_DEFAULT_MODE = None      # kept for source compatibility; unused (imperative only)
_DEFAULT_DEVICE = None


def torch_dtype(dtype=None):
    dtype = DEFAULT_DTYPE if dtype is None else dtype
    if isinstance(dtype, torch.dtype):
        return dtype
    name = getattr(dtype, '__name__', None) or str(dtype)
    name = name.replace('numpy.', '').replace("<class '", '').replace("'>", '')
    if name in ('float32', 'torch.float32', 'float'):
        return torch.float32
    if name in ('float64', 'torch.float64', 'double'):
        return torch.float64
    raise TypeError('unsupported dtype %r' % (dtype,))


def get_default_device():
    """The HIP device all arrays live on (reference: get_default_device -> mx.cpu(), config.py:43-52)."""
    if _DEFAULT_DEVICE is not None:
        return _DEFAULT_DEVICE
    if not torch.cuda.is_available():
        from .._lib import MXFError
        raise MXFError('mxfusion_amd needs an MI355X (no HIP device visible). There is no CPU fallback.')
    return torch.device('cuda', torch.cuda.current_device())


def set_default_device(dev):
    global _DEFAULT_DEVICE
    _DEFAULT_DEVICE = dev
