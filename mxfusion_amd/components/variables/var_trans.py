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


class _SoftplusManyFn(torch.autograd.Function):
    """softplus of several tensors as ONE kernel each way (concatenate -> mxf_softplus_fwd -> views): a model has five to ten positive
    parameters, and every one of them cost a launch forward and two backward in the step's launch-paced head and tail."""

    @staticmethod
    def forward(ctx, *xs):
        sizes = [x.numel() for x in xs]
        flat = torch.cat([x.reshape(-1) for x in xs]) if len(xs) > 1 else xs[0].reshape(-1)
        ctx.save_for_backward(flat)
        ctx.meta = (sizes, [x.shape for x in xs])
        y = ops.softplus(flat)
        return tuple(p.view(sh) for p, sh in zip(torch.split_with_sizes(y, sizes), ctx.meta[1]))

    @staticmethod
    def backward(ctx, *dys):
        (flat,) = ctx.saved_tensors
        sizes, shapes = ctx.meta
        parts = [(d.reshape(-1) if d is not None else flat.new_zeros(n)) for d, n in zip(dys, sizes)]
        dy = torch.cat(parts) if len(parts) > 1 else parts[0]
        dx = ops.softplus_bwd_(flat, dy, torch.zeros_like(flat))
        return tuple(p.view(sh) for p, sh in zip(torch.split_with_sizes(dx, sizes), shapes))


def transform_many(transformations, tensors):
    """[t.transform(x) for t, x in zip(...)] with every plain softplus (offset 0, device tensor) of the list batched into one kernel call."""
    out = list(tensors)
    idx = [i for i, (t, x) in enumerate(zip(transformations, tensors))
           if type(t) in (Softplus, PositiveTransformation) and not t._offset and isinstance(x, torch.Tensor) and x.is_cuda]
    if len(idx) > 1 and len({(tensors[i].dtype, tensors[i].device) for i in idx}) == 1:
        for i, y in zip(idx, _SoftplusManyFn.apply(*[tensors[i] for i in idx])):
            out[i] = y
        rest = [i for i in range(len(out)) if i not in set(idx)]
    else:
        rest = range(len(out))
    for i in rest:
        out[i] = transformations[i].transform(tensors[i])
    return out


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
