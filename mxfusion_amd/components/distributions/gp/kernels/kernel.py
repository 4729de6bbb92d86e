"""Kernel base classes (mxfusion/components/distributions/gp/kernels/kernel.py:96-373, add_kernel.py,
multiply_kernel.py).  K / Kdiag keep the reference signature K(F, X, X2=None, **kernel_params) with the
`<name>_` prefix convention; the arithmetic is the HIP Gram kernel (mxf_gram / mxf_gram_bwd)."""
import torch

from mxfusion_amd import ops
from mxfusion_amd.components.variables.variable import Variable


class _GramFn(torch.autograd.Function):
    """One fused pass for K, one fused pass for its reverse mode."""

    @staticmethod
    def forward(ctx, kind, ard, X, X2, ls, var):
        ctx.kind, ctx.ard = kind, ard
        ctx.save_for_backward(X, X2, ls, var)
        return ops.gram(kind, X, X2, ls, var, ard)

    @staticmethod
    def backward(ctx, dK):
        X, X2, ls, var = ctx.saved_tensors
        need = []
        if ctx.needs_input_grad[2]:
            need.append('X')
        if X2 is not None and ctx.needs_input_grad[3]:
            need.append('X2')
        if ctx.needs_input_grad[4]:
            need.append('ls')
        if ctx.needs_input_grad[5]:
            need.append('var')
        dX, dX2, dls, dvar = ops.gram_bwd(ctx.kind, X, X2, ls, var, ctx.ard, dK.contiguous(), need=need)
        return None, None, dX, dX2, dls, dvar


class Kernel(object):
    def __init__(self, input_dim, name, active_dims=None, dtype=None, ctx=None):
        self.input_dim = input_dim
        self.name = name
        self.active_dims = active_dims
        self.dtype = dtype
        self.ctx = ctx
        self._parameter_names = []

    def __setattr__(self, name, value):
        if isinstance(value, Variable) and not name.startswith('_'):
            if name not in self.__dict__.get('_parameter_names', []):
                self._parameter_names.append(name)
        object.__setattr__(self, name, value)

    @property
    def local_parameters(self):
        return {n: getattr(self, n) for n in self._parameter_names}

    def _strip(self, kernel_params):
        off = len(self.name) + 1
        return {k[off:]: v for k, v in kernel_params.items() if k.startswith(self.name + '_')}

    def _slice(self, X):
        if self.active_dims is not None and X is not None:
            return X[..., list(self.active_dims)].contiguous()
        return X

    def K(self, F, X, X2=None, **kernel_params):
        """kernel.py:96-123."""
        return self._compute_K(F=F, X=self._slice(X), X2=self._slice(X2), **self._strip(kernel_params))

    def Kdiag(self, F, X, **kernel_params):
        """kernel.py:125-147."""
        return self._compute_Kdiag(F=F, X=self._slice(X), **self._strip(kernel_params))

    def fetch_parameters(self, variables):
        """kernel.py:232-245: {prefixed name: runtime array}."""
        return {n: variables[v.uuid] for n, v in self.parameters.items()}

    def add(self, other, name='add'):
        if not isinstance(other, Kernel):
            raise TypeError('only kernels can be added to kernels')
        return AddKernel([self, other], name=name, dtype=self.dtype, ctx=self.ctx)

    def __add__(self, other):
        return self.add(other)

    def multiply(self, other, name='mul'):
        if not isinstance(other, Kernel):
            raise TypeError('only kernels can be multiplied with kernels')
        return MultiplyKernel([self, other], name=name, dtype=self.dtype, ctx=self.ctx)

    def __mul__(self, other):
        return self.multiply(other)

    # description used by the fused module algorithms: (kind, ard) for a single stationary kernel, else None
    def fused_spec(self):
        return None


class NativeKernel(Kernel):
    @property
    def parameters(self):
        return {self.name + '_' + n: getattr(self, n) for n in self._parameter_names}

    @property
    def parameter_names(self):
        return [self.name + '_' + n for n in self._parameter_names]


class CombinationKernel(Kernel):
    """kernel.py:317-373: sub-kernel parameters are exposed as `<comb>_<sub>_<param>`."""

    def __init__(self, sub_kernels, name, dtype=None, ctx=None):
        super(CombinationKernel, self).__init__(input_dim=sub_kernels[0].input_dim, name=name, dtype=dtype, ctx=ctx)
        names = [k.name for k in sub_kernels]
        if len(set(names)) != len(names):   # reference renames duplicates <name>0, <name>1 ...
            seen = {}
            for k in sub_kernels:
                c = seen.get(k.name, 0)
                seen[k.name] = c + 1
                if names.count(k.name) > 1:
                    k.name = k.name + str(c)
        self.sub_kernels = sub_kernels

    @property
    def parameters(self):
        return {self.name + '_' + n: v for k in self.sub_kernels for n, v in k.parameters.items()}

    @property
    def parameter_names(self):
        return list(self.parameters)


class AddKernel(CombinationKernel):
    def __init__(self, sub_kernels, name='add', dtype=None, ctx=None):
        super(AddKernel, self).__init__(sub_kernels, name, dtype, ctx)

    def _compute_K(self, F, X, X2=None, **params):
        """add_kernel.py:44-68."""
        K = self.sub_kernels[0].K(F, X, X2, **params)
        for k in self.sub_kernels[1:]:
            K = K + k.K(F, X, X2, **params)
        return K

    def _compute_Kdiag(self, F, X, **params):
        K = self.sub_kernels[0].Kdiag(F, X, **params)
        for k in self.sub_kernels[1:]:
            K = K + k.Kdiag(F, X, **params)
        return K


class MultiplyKernel(CombinationKernel):
    def __init__(self, sub_kernels, name='mul', dtype=None, ctx=None):
        super(MultiplyKernel, self).__init__(sub_kernels, name, dtype, ctx)

    def _compute_K(self, F, X, X2=None, **params):
        """multiply_kernel.py:44-67."""
        K = self.sub_kernels[0].K(F, X, X2, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.K(F, X, X2, **params)
        return K

    def _compute_Kdiag(self, F, X, **params):
        K = self.sub_kernels[0].Kdiag(F, X, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.Kdiag(F, X, **params)
        return K
