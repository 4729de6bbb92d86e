"""Kernel base classes (mxfusion/components/distributions/gp/kernels/kernel.py:96-373, add_kernel.py,
multiply_kernel.py).  K / Kdiag keep the reference signature K(F, X, X2=None, **kernel_params) with the
`<name>_` prefix convention; the arithmetic is the HIP Gram kernel (mxf_gram / mxf_gram_bwd)."""
import torch

from mxfusion_amd import ops
from mxfusion_amd.components.variables.variable import Variable
from mxfusion_amd.components.factor import Factor
from mxfusion_amd.components.functions.function_evaluation import FunctionEvaluation


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


class _Gram2Fn(torch.autograd.Function):
    """k1 + k2 or k1 * k2 of two stationary kernels in ONE forward pass (mxf_gram2); the reverse mode is mxf_gram_bwd per sub-kernel --
    with dK for a sum, with dK times the other kernel's Gram (recomputed: one mxf_gram each) for a product."""

    @staticmethod
    def forward(ctx, spec, X, X2, ls1, var1, ls2, var2):
        ctx.spec = spec                               # (kind1, ard1, kind2, ard2, op)
        ctx.save_for_backward(X, X2, ls1, var1, ls2, var2)
        k1, a1, k2, a2, op = spec
        return ops.gram2(k1, k2, op, X, X2, ls1, var1, a1, ls2, var2, a2)

    @staticmethod
    def backward(ctx, dK):
        X, X2, ls1, var1, ls2, var2 = ctx.saved_tensors
        k1, a1, k2, a2, op = ctx.spec
        dK = dK.contiguous()
        need = []
        if ctx.needs_input_grad[1]:
            need.append('X')
        if X2 is not None and ctx.needs_input_grad[2]:
            need.append('X2')
        out = [None, None]          # dX, dX2
        grads = []
        for (kind, ard, ls, var, other, gi) in ((k1, a1, ls1, var1, (k2, a2, ls2, var2), 3), (k2, a2, ls2, var2, (k1, a1, ls1, var1), 5)):
            nd = list(need) + (['ls'] if ctx.needs_input_grad[gi] else []) + (['var'] if ctx.needs_input_grad[gi + 1] else [])
            if not nd:                    # nothing of this sub-kernel is differentiated: no recomputed Gram, no reverse pass
                grads += [None, None]
                continue
            dKi = dK if op == ops.ACC_ADD else dK * ops.gram(other[0], X, X2, other[2], other[3], other[1])
            dX, dX2, dls, dvar = ops.gram_bwd(kind, X, X2, ls, var, ard, dKi, need=nd)
            out[0] = dX if out[0] is None else (out[0] if dX is None else out[0] + dX)
            out[1] = dX2 if out[1] is None else (out[1] if dX2 is None else out[1] + dX2)
            grads += [dls, dvar]
        return (None, out[0], out[1]) + tuple(grads)


def rename_duplicate_names(names):
    """util/util.py:65-100: [(index, new name)] for every repeated name -- the repeat gets the first free `<prefix><count>` (a trailing
    integer of the name counts as its counter): ['a', 'b', 'a', 'a'] -> [(2, 'a0'), (3, 'a1')]."""
    import re
    all_names = set(names)
    if len(all_names) == len(names):
        return []
    cur, prog, renames = set(), re.compile(r'^(.*)(\d+)$'), []
    for i, n in enumerate(names):
        if n in cur:
            res = prog.match(n)
            prefix, count = (n, 0) if res is None else (res.groups()[0], int(res.groups()[1]) + 1)
            while prefix + str(count) in all_names:
                count += 1
            renames.append((i, prefix + str(count)))
            all_names.add(prefix + str(count))
        else:
            cur.add(n)
    return renames


class KernelFunctionEvaluation(FunctionEvaluation):
    """A kernel used as a function inside a model (kernel.py:29 `Kernel(MXFusionFunction)`, mxfusion_function.py:55-79
    FunctionEvaluationWithParameters): inputs X (, X2) and the kernel's parameters under their prefixed names, one output `covariance`;
    the inputs are attributes of the factor (`fe.X`, `fe.rbf_lengthscale`) as in the reference's factor."""

    def __init__(self, kernel, inputs):
        out = Variable(shape=None)
        Factor.__init__(self, inputs, [('covariance', out)], [n for n, _ in inputs], ['covariance'])       # (a FunctionEvaluation to the factor graph's walk)
        self._func = self._kernel = kernel
        for n, v in inputs:
            object.__setattr__(self, n, v)

    def eval(self, F, variables, always_return_tuple=False):
        kw = {n: variables[v.uuid] for n, v in self.inputs}
        X, X2 = kw.pop('X'), kw.pop('X2', None)
        K = self._kernel.K(F, X, X2, **kw)
        return (K,) if always_return_tuple else K


class Kernel(object):
    def __call__(self, X, X2=None, **kernel_params):
        """rbf(X_var, X2_var, rbf_lengthscale=l_var, ...) -> the covariance Variable (its .factor evaluates K); parameters that are not
        given are the kernel's own Variables (mxfusion_function.py:55-79, _parse_arguments)."""
        own = self.parameters
        unknown = [k for k in kernel_params if k not in own]
        if unknown:
            raise TypeError('%s(): unknown kernel parameter(s) %s (the kernel has %s)' % (self.name, unknown, sorted(own)))
        inputs = [('X', X)] + ([('X2', X2)] if X2 is not None else [])
        inputs += [(n, kernel_params.get(n, own[n])) for n in self.parameter_names]
        fe = KernelFunctionEvaluation(self, inputs)
        return fe.outputs[0][1]

    def eval(self, F, X, X2=None, **kernel_params):
        """kernel.py:247-259."""
        return self.K(F, X, X2, **kernel_params)

    def replicate_self(self, attribute_map=None):
        """kernel.py:261-273 / :365-373: a copy of the kernel whose parameter Variables are replicas (same UUIDs) -- what a module's internal
        graphs hold of the kernel the user passed in."""
        import copy
        rep = copy.copy(self)
        object.__setattr__(rep, '_parameter_names', [])
        rep.active_dims = copy.copy(self.active_dims)
        for n in self._parameter_names:
            setattr(rep, n, getattr(self, n).replicate_self())
        if hasattr(self, 'sub_kernels'):
            rep.sub_kernels = [k.replicate_self(attribute_map) for k in self.sub_kernels]
            for k in rep.sub_kernels:
                object.__setattr__(rep, k.name, k)
        return rep

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
        sub_kernels = list(sub_kernels)
        for i, n in rename_duplicate_names([k.name for k in sub_kernels]):       # kernel.py:333-335: rbf, rbf -> rbf, rbf0
            sub_kernels[i].name = n
        super(CombinationKernel, self).__init__(input_dim=max(k.input_dim for k in sub_kernels), name=name, dtype=dtype, ctx=ctx)
        self.sub_kernels = sub_kernels
        for k in sub_kernels:
            object.__setattr__(self, k.name, k)                                  # kernel.py:339-340: kern.rbf, kern.linear

    @property
    def parameters(self):
        return {self.name + '_' + n: v for k in self.sub_kernels for n, v in k.parameters.items()}

    @property
    def parameter_names(self):
        return list(self.parameters)

    def _fused_pair(self, X, X2, params, op):
        """Two stationary sub-kernels on the same active dimensions (the deep-GP config's Matern52 + RBF): ONE Gram pass with a two-kernel
        epilogue (mxf_gram2) instead of one materialised Gram per sub-kernel plus the combining pass.  None when the pair does not qualify."""
        ks = self.sub_kernels
        if len(ks) != 2 or not X.is_cuda or any(getattr(k, '_kind', None) is None or not hasattr(k, 'ARD') for k in ks):
            return None
        if ks[0].active_dims != ks[1].active_dims:
            return None
        Xs, X2s = ks[0]._slice(X), ks[0]._slice(X2)
        if Xs.shape[-1] > 16:
            return None
        if Xs.shape[-2] > 16 * 65535:         # one launch of mxf_gram2 covers 65 535 row blocks of 16: larger Grams take the per-kernel path
            return None
        p0, p1 = ks[0]._strip(params), ks[1]._strip(params)
        spec = (ks[0]._kind, bool(ks[0].ARD), ks[1]._kind, bool(ks[1].ARD), op)
        return _Gram2Fn.apply(spec, Xs, X2s, p0['lengthscale'], p0['variance'], p1['lengthscale'], p1['variance'])


class AddKernel(CombinationKernel):
    def __init__(self, sub_kernels, name='add', dtype=None, ctx=None):
        flat = []
        for k in sub_kernels:          # add_kernel.py:36-43: a sum of sums is ONE sum (rbf + (rbf + linear) -> add(rbf, rbf0, linear))
            flat.extend(k.sub_kernels if isinstance(k, AddKernel) else [k])
        super(AddKernel, self).__init__(flat, name, dtype, ctx)

    def _compute_K(self, F, X, X2=None, **params):
        """add_kernel.py:44-68."""
        K = self._fused_pair(X, X2, params, ops.ACC_ADD)
        if K is not None:
            return K
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
        K = self._fused_pair(X, X2, params, ops.ACC_MUL)
        if K is not None:
            return K
        K = self.sub_kernels[0].K(F, X, X2, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.K(F, X, X2, **params)
        return K

    def _compute_Kdiag(self, F, X, **params):
        K = self.sub_kernels[0].Kdiag(F, X, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.Kdiag(F, X, **params)
        return K
