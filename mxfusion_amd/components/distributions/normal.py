"""Normal distribution (mxfusion/components/distributions/normal.py:24-116).  log_pdf and reparameterised
sampling run as HIP kernels (mxf_normal_logpdf / mxf_normal_reparam, elementwise.hip) with fused reverse
mode; this is q(X) and p(X) of the Monte-Carlo ELBO path."""
import torch

from ... import ops
from .distribution import Distribution
from ..variables.variable import Variable


class _NormalLogPdfSumFn(torch.autograd.Function):
    """sum_i mean_s log N(x[s,i] | mean[i], var[i]) * scaling  -- the quantity FactorGraph.log_pdf adds
    (models/factor_graph.py:221-224); value and gradients in one kernel pass."""

    @staticmethod
    def forward(ctx, x, mean, var, scaling):
        S = x.shape[0]
        need = [ctx.needs_input_grad[i] for i in range(3)]
        m1, v1 = mean.reshape(-1), var.reshape(-1)
        # the kernel accumulates into its outputs: ONE zero-filled buffer carved into (out, dm, dv, dx) instead of four fills per call
        # (the step's head and tail are paced by the number of launches, ~5 us each)
        sizes = [1, m1.numel() if need[1] else 0, v1.numel() if need[2] else 0, x.numel() if need[0] else 0]
        buf = torch.zeros(sum(sizes) + 12, dtype=x.dtype, device=x.device)
        off = [0, 4, 4 + (sizes[1] + 3) // 4 * 4]                       # 16-byte aligned starts
        off.append(off[2] + (sizes[2] + 3) // 4 * 4)
        out = buf[0:1]
        dm = buf[off[1]:off[1] + sizes[1]] if need[1] else None
        dv = buf[off[2]:off[2] + sizes[2]] if need[2] else None
        dx = buf[off[3]:off[3] + sizes[3]].view(x.shape) if need[0] else None
        ops.normal_logpdf_(x, m1, v1, float(scaling) / S, out, dx, dm, dv)
        ctx.grads = (dx, dm, dv, mean.shape, var.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        dx, dm, dv, ms, vs = ctx.grads
        have = [t for t in (dx, dm, dv) if t is not None]
        prod = iter(torch._foreach_mul(have, g) if have else [])          # one multi-tensor launch instead of one per gradient
        dx, dm, dv = [next(prod) if t is not None else None for t in (dx, dm, dv)]
        return (dx, None if dm is None else dm.reshape(ms), None if dv is None else dv.reshape(vs), None)


class _NormalReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, eps):
        ctx.save_for_backward(var, eps)
        ctx.shapes = (mean.shape, var.shape)
        return ops.normal_reparam(mean.reshape(-1), var.reshape(-1), eps)

    @staticmethod
    def backward(ctx, dx):
        var, eps = ctx.saved_tensors
        ms, vs = ctx.shapes
        n = var.numel()
        buf = torch.zeros(2 * ((n + 3) // 4 * 4), dtype=var.dtype, device=var.device)
        dm, dv = buf[:n], buf[(n + 3) // 4 * 4:(n + 3) // 4 * 4 + n]
        ops.normal_reparam_bwd_(var.reshape(-1), eps, dx.contiguous(), dm, dv)
        return dm.reshape(ms), dv.reshape(vs), None


class Normal(Distribution):
    def __init__(self, mean, variance, rand_gen=None, dtype=None, ctx=None):
        mean = self._as_variable(mean)
        variance = self._as_variable(variance)
        super(Normal, self).__init__([('mean', mean), ('variance', variance)], None, ['mean', 'variance'], ['random_variable'],
                                     rand_gen, dtype, ctx)

    @staticmethod
    def _per_element(p, x):
        """mean/variance with the sample axis: (1, ...) broadcastable against x (S, ...) -> single element or per element."""
        if p.numel() == 1:
            return p.reshape(1)
        if p.shape[0] == 1 and tuple(p.shape[1:]) == tuple(x.shape[1:]):
            return p[0]
        return None

    def log_pdf_sum(self, F, variables):
        """sum(mean_S(log_pdf)) fused (normal.py:52-70 + factor_graph.py:223)."""
        x = variables[self.random_variable.uuid]
        mean, var = variables[self.inputs[0][1].uuid], variables[self.inputs[1][1].uuid]
        m, v = self._per_element(mean, x), self._per_element(var, x)
        if m is None or v is None:   # sampled mean/variance: generic (elementwise torch, still on device)
            return self.log_pdf(F, variables).mean(dim=0).sum()
        return _NormalLogPdfSumFn.apply(x.contiguous(), m, v, self.log_pdf_scaling)

    def log_pdf_impl(self, mean, variance, random_variable, F=None):
        import math
        logvar = math.log(2 * math.pi) / -2 + torch.log(variance) / -2
        return (logvar + (random_variable - mean) ** 2 / (-2 * variance)) * self.log_pdf_scaling

    def draw_samples_impl(self, mean, variance, rv_shape, num_samples=1, F=None):
        out_shape = (num_samples,) + tuple(rv_shape)
        dtype = mean.dtype if isinstance(mean, torch.Tensor) else None
        dev = mean.device if isinstance(mean, torch.Tensor) else None
        eps = self._rand_gen.sample_normal(shape=out_shape, dtype=dtype, ctx=dev)
        if mean.shape[0] == 1 and variance.shape[0] == 1:
            n = 1
            for s in rv_shape:
                n *= int(s)
            m = mean[0].expand(tuple(rv_shape)).contiguous() if mean.numel() != n else mean[0]
            v = variance[0].expand(tuple(rv_shape)).contiguous() if variance.numel() != n else variance[0]
            return _NormalReparamFn.apply(m, v, eps.reshape(num_samples, -1).contiguous()).reshape(out_shape)
        return eps * torch.sqrt(variance) + mean

    @staticmethod
    def define_variable(mean=0., variance=1., shape=None, rand_gen=None, dtype=None, ctx=None):
        normal = Normal(mean=mean, variance=variance, rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        normal._generate_outputs(shape=shape)
        return normal.random_variable
