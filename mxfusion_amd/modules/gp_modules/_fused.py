"""autograd bridges to the fused C-ABI composites (value + reverse mode in one call)."""
import torch

from ... import ops


def _uniform_weight(g):
    """sum(g) for an upstream gradient that weights every sample equally -- the only reduction the reference applies to a module's
    log-pdf is mean_S (factor_graph.py:233), and the fused composites return the gradients of gscale * sum_s logL[s] with ONE gscale.
    A non-uniform weighting cannot be represented by that call: instead of returning silently wrong gradients the result is poisoned
    with NaN (checked on the device: no host synchronisation, safe inside a hipGraph capture)."""
    if g.numel() == 1:
        return g.reshape(())
    if g.is_cuda and g.dtype in (torch.float32, torch.float64):
        return ops.uniform_sum(g)           # one launch instead of ~8 element-wise ones in the step's tail
    c = g.sum()
    spread = (g - c / g.numel()).abs().max()
    return torch.where(spread <= 1e-6 * c.abs() / g.numel(), c, torch.full_like(c, float('nan')))


def _scaled(grads, shapes, needs, c):
    """(grad.reshape(shape) * c for the inputs that need a gradient, None for the others) -- all products in ONE multi-tensor launch
    (the step's tail is paced by its number of launches)."""
    idx = [i for i, n in enumerate(needs) if n]
    prod = torch._foreach_mul([grads[i] for i in idx], c) if idx else []
    out = [None] * len(needs)
    for i, p in zip(idx, prod):
        out[i] = p.reshape(shapes[i])
    return tuple(out)


class GPLogPdfFn(torch.autograd.Function):
    """mxf_gp_logpdf: logL (S,), and the posterior side products L, LinvY (gp_regression.py:72-75)."""

    @staticmethod
    def forward(ctx, kind, ard, jitter, X, Y, noise, ls, var):
        want = any(ctx.needs_input_grad[3:])
        r = ops.gp_logpdf(kind, X, Y, noise, ls, var, ard, jitter=jitter, want_grad=want)
        ctx.want = want
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dnoise'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, noise, ls, var))
        ctx.mark_non_differentiable(r['L'], r['LinvY'], r['info'])
        return r['logL'], r['L'], r['LinvY'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        out = []
        for grad, shp, need in zip(ctx.grads, ctx.shapes, ctx.needs_input_grad[3:]):
            if not need:
                out.append(None)
                continue
            gg = grad.reshape((grad.shape[0],) + tuple(shp[1:])) * g.reshape((-1,) + (1,) * (len(shp) - 1))
            if shp[0] == 1 and gg.shape[0] > 1:      # primal broadcast over S: its gradient sums over samples
                gg = gg.sum(0, keepdim=True)
            out.append(gg)
        return (None, None, None) + tuple(out)


class SVGPLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf.  Gradients are produced for mean_S(logL) (the only reduction the reference applies to a
    module's log-pdf, factor_graph.py:233) and scaled by sum(grad_output) in backward."""

    @staticmethod
    def forward(ctx, kind, ard, jitter, scaling, X, Y, Z, noise, mu, W, sdiag, ls, var):
        want = any(ctx.needs_input_grad[4:])
        S = max(X.shape[0], Y.shape[0])
        r = ops.svgp_logpdf(kind, X, Y, Z[0], noise[0] if noise.dim() == 3 else noise.reshape(-1), mu[0], W[0], sdiag[0], ls.reshape(-1), var.reshape(-1), ard,
                            jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dZ'], r['dnoise'], r['dmu'], r['dW'], r['dSdiag'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, Z, noise, mu, W, sdiag, ls, var))
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)
        return (None, None, None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[4:], c)


class SVGPMatLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf_mat: the bound from materialised Kuu / Kuf / Kdiag (one sample of the inputs: unit sample axes) for S >= 1 samples of Y;
    the gradients of mean_S(logL) flow on into the kernels' own reverse mode (combination kernels)."""

    @staticmethod
    def forward(ctx, jitter, scaling, Kuu, Kuf, Kdiag, Y, noise, mu, W, sdiag):
        want = any(ctx.needs_input_grad[2:])
        S = Y.shape[0]
        r = ops.svgp_logpdf_mat(Kuu[0], Kuf[0], Kdiag[0], Y if S > 1 else Y[0], noise[0] if noise.dim() == 3 else noise.reshape(-1), mu[0], W[0],
                                sdiag[0], jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
        if want:
            ctx.grads = (r['dKuu'], r['dKuf'], r['dKdiag'], r['dY'], r['dnoise'], r['dmu'], r['dW'], r['dSdiag'])
            ctx.shapes = tuple(t.shape for t in (Kuu, Kuf, Kdiag, Y, noise, mu, W, sdiag))
            ctx.S = S
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)          # gradients were produced for mean_S(logL) (gscale = 1/S): scale by sum(grad_output), as SVGPLogPdfFn does
        return (None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[2:], c)


class SGPLogPdfFn(torch.autograd.Function):
    """mxf_sgp_logpdf for ONE sample (arrays carry a unit sample axis); returns logL (1,), wv, L, LA."""

    @staticmethod
    def forward(ctx, kind, ard, jitter, X, Y, Z, noise, ls, var):
        want = any(ctx.needs_input_grad[3:])
        r = ops.sgp_logpdf(kind, X[0], Y[0], Z[0], noise.reshape(-1), ls.reshape(-1), var.reshape(-1), ard, jitter=jitter, gscale=1.0,
                           want_grad=want)
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dZ'], r['dnoise'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, Z, noise, ls, var))
        ctx.mark_non_differentiable(r['wv'], r['L'], r['LA'], r['info'])
        return r['logL'], r['wv'], r['L'], r['LA'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = g.sum()
        out = [(grad.reshape(shp) * c) if need else None for grad, shp, need in zip(ctx.grads, ctx.shapes, ctx.needs_input_grad[3:])]
        return (None, None, None) + tuple(out)
