"""Variable transformations (mxfusion/components/variables/var_trans.py:53-102).  Softplus runs as a HIP
kernel (mxf_softplus_fwd / _bwd) wrapped in an autograd Function."""
import torch

from ... import ops


class _SoftplusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.softplus(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.softplus_bwd_(x, dy, torch.zeros_like(x))


class VariableTransformation(object):
    def transform(self, var, F=None, dtype=None):
        raise NotImplementedError

    def inverseTransform(self, out_var, F=None, dtype=None):
        raise NotImplementedError


class Softplus(VariableTransformation):
    """f = log(1+exp(x)) + c ; f^-1 = log(exp(x-c)-1)   (var_trans.py:53-91)."""

    def __init__(self, offset):
        self._offset = offset

    def transform(self, var, F=None, dtype=None):
        y = _SoftplusFn.apply(var)
        return y + self._offset if self._offset else y

    def inverseTransform(self, out_var, F=None, dtype=None):
        # host-side, once per parameter assignment (var_trans.py:91: log(expm1(y)))
        return torch.log(torch.expm1(out_var - self._offset))


class PositiveTransformation(Softplus):
    def __init__(self):
        super(PositiveTransformation, self).__init__(offset=0.)
