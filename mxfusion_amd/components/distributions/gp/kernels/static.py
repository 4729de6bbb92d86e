"""Linear / Bias / White kernels (kernels/linear.py:24-103, static.py:24-164).  Forward is the HIP Gram
kernel; these are 'next-row' kernels: their reverse mode is not implemented yet, so their
parameters are treated as constants by autograd."""
import torch

from mxfusion_amd import ops
from mxfusion_amd.components.variables.variable import Variable
from mxfusion_amd.components.variables.var_trans import PositiveTransformation
from .kernel import NativeKernel


class Linear(NativeKernel):
    def __init__(self, input_dim, ARD=False, variances=1., name='linear', active_dims=None, dtype=None, ctx=None):
        super(Linear, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        self.ARD = ARD
        if not isinstance(variances, Variable):
            variances = Variable(shape=(input_dim if ARD else 1,), transformation=PositiveTransformation(), initial_value=variances)
        self.variances = variances

    def _compute_K(self, F, X, variances, X2=None):
        return ops.gram('linear', X.detach(), None if X2 is None else X2.detach(), variances.detach(), None, self.ARD)

    def _compute_Kdiag(self, F, X, variances):
        return ((X ** 2) * variances.unsqueeze(-2)).sum(-1)


class Bias(NativeKernel):
    def __init__(self, input_dim, variance=1., name='bias', active_dims=None, dtype=None, ctx=None):
        super(Bias, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        if not isinstance(variance, Variable):
            variance = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=variance)
        self.variance = variance

    def _compute_K(self, F, X, variance, X2=None):
        return ops.gram('bias', X.detach(), None if X2 is None else X2.detach(), None, variance.detach(), False)

    def _compute_Kdiag(self, F, X, variance):
        return torch.zeros(X.shape[:-1], dtype=X.dtype, device=X.device) + variance


class White(NativeKernel):
    def __init__(self, input_dim, variance=1., name='white', active_dims=None, dtype=None, ctx=None):
        super(White, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        if not isinstance(variance, Variable):
            variance = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=variance)
        self.variance = variance

    def _compute_K(self, F, X, variance, X2=None):
        return ops.gram('white', X.detach(), None if X2 is None else X2.detach(), None, variance.detach(), False)

    def _compute_Kdiag(self, F, X, variance):
        return torch.zeros(X.shape[:-1], dtype=X.dtype, device=X.device) + variance
