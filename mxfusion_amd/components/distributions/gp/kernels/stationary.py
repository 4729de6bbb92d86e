"""Stationary kernels (kernels/stationary.py:45-124, rbf.py:24-72, matern.py:24-151)."""
import torch

from mxfusion_amd.components.variables.variable import Variable
from mxfusion_amd.components.variables.var_trans import PositiveTransformation
from .kernel import NativeKernel, _GramFn


class StationaryKernel(NativeKernel):
    _kind = None

    def __init__(self, input_dim, ARD=False, variance=1., lengthscale=1., name='stationary', active_dims=None, dtype=None, ctx=None):
        super(StationaryKernel, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        self.ARD = ARD
        if not isinstance(variance, Variable):
            variance = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=variance)
        if not isinstance(lengthscale, Variable):
            lengthscale = Variable(shape=(input_dim if ARD else 1,), transformation=PositiveTransformation(), initial_value=lengthscale)
        self.variance = variance
        self.lengthscale = lengthscale

    def _compute_K(self, F, X, lengthscale, variance, X2=None):
        return _GramFn.apply(self._kind, bool(self.ARD), X, X2, lengthscale, variance)

    def _compute_Kdiag(self, F, X, lengthscale, variance):
        """stationary.py:109-124: zeros(X.shape[:-1]) + variance."""
        return torch.zeros(X.shape[:-1], dtype=X.dtype, device=X.device) + variance

    def fused_spec(self):
        return (self._kind, bool(self.ARD)) if self.active_dims is None else None


class RBF(StationaryKernel):
    _kind = 'rbf'

    def __init__(self, input_dim, ARD=False, variance=1., lengthscale=1., name='rbf', active_dims=None, dtype=None, ctx=None):
        super(RBF, self).__init__(input_dim, ARD, variance, lengthscale, name, active_dims, dtype, ctx)


class Matern52(StationaryKernel):
    _kind = 'matern52'

    def __init__(self, input_dim, ARD=False, variance=1., lengthscale=1., name='matern52', active_dims=None, dtype=None, ctx=None):
        super(Matern52, self).__init__(input_dim, ARD, variance, lengthscale, name, active_dims, dtype, ctx)


class Matern32(StationaryKernel):
    _kind = 'matern32'

    def __init__(self, input_dim, ARD=False, variance=1., lengthscale=1., name='matern32', active_dims=None, dtype=None, ctx=None):
        super(Matern32, self).__init__(input_dim, ARD, variance, lengthscale, name, active_dims, dtype, ctx)


class Matern12(StationaryKernel):
    _kind = 'matern12'

    def __init__(self, input_dim, ARD=False, variance=1., lengthscale=1., name='matern12', active_dims=None, dtype=None, ctx=None):
        super(Matern12, self).__init__(input_dim, ARD, variance, lengthscale, name, active_dims, dtype, ctx)
