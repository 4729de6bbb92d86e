"""The reference's two custom operators (mxfusion/util/customop.py:22-148).

`make_diagonal` runs as HIP kernels behind the C ABI (mxf_make_diagonal / mxf_diag_of) with the reverse mode of customop.py:49-61;
`broadcast_to_w_samples` is a reshape + stride-0 expand (no copy, no kernel: the sample axis is never materialised here), whose reverse
mode -- the sum over the broadcast axes (customop.py:102-110) -- is autograd's."""
import torch

from .. import ops


class _MakeDiagonalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return ops.make_diagonal(a)

    @staticmethod
    def backward(ctx, g):
        return ops.diag_of(g.contiguous())


def make_diagonal(F, x, name="make_diagonal"):
    """(..., M) -> (..., M, M) with x on the diagonal (customop.py:22-81); `F` and `name` are kept for signature compatibility."""
    return _MakeDiagonalFn.apply(x)


def broadcast_to_w_samples(F, data, shape, isSamples=True):
    """customop.py:130-148: broadcast `data` to `shape`; with isSamples the leading axis of `data` is the sample axis and stays in front."""
    shape = tuple(shape)
    n_dim = len(shape)
    if isSamples:
        num_samples = max(data.shape[0], shape[0])
        t_shape = (data.shape[0],) + (1,) * (n_dim - data.dim()) + tuple(data.shape[1:])
        shape = (num_samples,) + shape[1:]
    else:
        t_shape = (1,) * (n_dim - data.dim()) + tuple(data.shape)
    return data.reshape(t_shape).expand(shape)
