"""Linear / Bias / White kernels (kernels/linear.py:24-103, static.py:24-164).  Forward is the HIP Gram kernel (mxf_gram);
the reverse modes are closed forms over mxf_gemm:
  Linear  K = X diag(v) X2^T : dX = dK (X2 . v), dX2 = dK^T (X . v), dv_q = sum_i X_iq (dK X2)_iq
  Bias    K = v 1 1^T        : dv = 1^T dK 1
  White   K = v I (X2 None)  : dv = tr(dK)"""
import torch

from mxfusion_amd import ops
from mxfusion_amd.components.variables.variable import Variable
from mxfusion_amd.components.variables.var_trans import PositiveTransformation
from .kernel import NativeKernel


def _sum_to(g, like):
    """gradient of a broadcast operand: sum over the sample axis when the primal had a unit sample axis."""
    if like.shape[0] == 1 and g.shape[0] > 1:
        g = g.sum(0, keepdim=True)
    return g.reshape(like.shape)


class _LinearGramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ard, X, X2, variances):
        ctx.save_for_backward(X, X2, variances)
        return ops.gram('linear', X, X2, variances, None, ard)

    @staticmethod
    def backward(ctx, dK):
        X, X2, v = ctx.saved_tensors
        dK = dK.contiguous()
        vb = v.unsqueeze(-2)                                                       # (S|1, 1, Q|1)
        gX = gX2 = gv = None
        if X2 is None:
            T = ops.gemm(dK, X) + ops.gemm(dK, X, transA=True)                     # (dK + dK^T) X
            if ctx.needs_input_grad[1]:
                gX = _sum_to(T * vb, X)
            if ctx.needs_input_grad[3]:
                gv = 0.5 * (X * T).sum(-2)
        else:
            T = ops.gemm(dK, X2)                                                   # dK X2   (S, N, Q)
            if ctx.needs_input_grad[1]:
                gX = _sum_to(T * vb, X)
            if ctx.needs_input_grad[2]:
                gX2 = _sum_to(ops.gemm(dK, X, transA=True) * vb, X2)
            if ctx.needs_input_grad[3]:
                gv = (X * T).sum(-2)
        if gv is not None:
            if v.shape[-1] == 1:
                gv = gv.sum(-1, keepdim=True)
            gv = _sum_to(gv, v)
        return None, gX, gX2, gv


class _ConstGramFn(torch.autograd.Function):
    """Bias / White: K does not depend on the inputs' values."""

    @staticmethod
    def forward(ctx, kind, X, X2, variance):
        ctx.kind, ctx.square = kind, X2 is None
        ctx.vshape = variance.shape
        return ops.gram(kind, X, X2, None, variance, False)

    @staticmethod
    def backward(ctx, dK):
        if not ctx.needs_input_grad[3]:
            return None, None, None, None
        dK = dK.contiguous()
        S, N, N2 = dK.shape
        if ctx.kind == 'white':
            g = torch.diagonal(dK, dim1=-2, dim2=-1).sum(-1, keepdim=True) if ctx.square else torch.zeros(S, 1, dtype=dK.dtype, device=dK.device)
        else:
            ones_r = torch.ones(1, 1, N, dtype=dK.dtype, device=dK.device)
            ones_c = torch.ones(1, N2, 1, dtype=dK.dtype, device=dK.device)
            g = ops.gemm(ops.gemm(ones_r, dK), ones_c).reshape(S, 1)               # 1^T dK 1 through mxf_gemm
        if ctx.vshape[0] == 1 and S > 1:
            g = g.sum(0, keepdim=True)
        return None, None, None, g.reshape(ctx.vshape)


class Linear(NativeKernel):
    def __init__(self, input_dim, ARD=False, variances=1., name='linear', active_dims=None, dtype=None, ctx=None):
        super(Linear, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        self.ARD = ARD
        if not isinstance(variances, Variable):
            variances = Variable(shape=(input_dim if ARD else 1,), transformation=PositiveTransformation(), initial_value=variances)
        self.variances = variances

    def _compute_K(self, F, X, variances, X2=None):
        return _LinearGramFn.apply(self.ARD, X, X2, variances)

    def _compute_Kdiag(self, F, X, variances):
        return ((X ** 2) * variances.unsqueeze(-2)).sum(-1)


class Bias(NativeKernel):
    def __init__(self, input_dim, variance=1., name='bias', active_dims=None, dtype=None, ctx=None):
        super(Bias, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        if not isinstance(variance, Variable):
            variance = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=variance)
        self.variance = variance

    def _compute_K(self, F, X, variance, X2=None):
        return _ConstGramFn.apply('bias', X, X2, variance)

    def _compute_Kdiag(self, F, X, variance):
        return torch.zeros(X.shape[:-1], dtype=X.dtype, device=X.device) + variance


class White(NativeKernel):
    def __init__(self, input_dim, variance=1., name='white', active_dims=None, dtype=None, ctx=None):
        super(White, self).__init__(input_dim=input_dim, name=name, active_dims=active_dims, dtype=dtype, ctx=ctx)
        if not isinstance(variance, Variable):
            variance = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=variance)
        self.variance = variance

    def _compute_K(self, F, X, variance, X2=None):
        return _ConstGramFn.apply('white', X, X2, variance)

    def _compute_Kdiag(self, F, X, variance):
        return torch.zeros(X.shape[:-1], dtype=X.dtype, device=X.device) + variance
