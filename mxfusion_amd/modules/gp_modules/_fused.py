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


class Float32Guard(object):
    """Validity of the float32 streaming form of the SVGP bound.  The float32 training call applies
    H0 = Kuu^-1 - Kuu^-1 Su Kuu^-1 explicitly, so its rounding error grows like cond_1(Kuu + jitter I) 2^-24: 5e-8 ... 5e-6 on the ELBO up to
    cond ~ 1.4e3, 2e-3 at 5e4 (tests/probes/f32_accuracy.py) -- beyond LIMIT the 1e-5 parity bar is out of reach in float32.  (A factorised
    float32 form was costed instead of guessed: the cancellation sits in k_n^T (H0 k_n) itself, so only q_n = v_n^T (I - A_s A_s^T) v_n with
    v_n = L^-1 k_n avoids it, which needs three full-width GEMMs plus two passes over them, ~2.5x the step, against the reference's own
    remedy -- float64, svgp_regression.py:83-92 solves in the model's dtype -- at 5.8x.)

    Every training call publishes its condition number into pinned host memory as its last launch (mxf_svgp_cond_nowait); the guard polls
    that word before each float32 call -- no synchronisation, it lags by the calls still in flight -- and once the limit is crossed every
    further float32 SVGP call of the process runs its streaming stage in FLOAT64 (inputs widened, outputs narrowed; same C-ABI call with
    dtype = f64), with one warning.  The first training call of a process is checked synchronously, so a model that STARTS ill-conditioned is
    never evaluated in float32 at all."""
    LIMIT = 3e3
    enabled = True
    active = False          # sticky: the float64 fallback is on
    _checked_first = False

    @classmethod
    def reset(cls):
        cls.active, cls._checked_first = False, False

    @classmethod
    def _trip(cls, cond):
        if not cls.active:
            import warnings
            warnings.warn('mxfusion_amd: cond_1(Kuu + jitter I) = %.2e exceeds %.0e: the float32 streaming form of the SVGP bound loses '
                          'accuracy there (error ~ cond * 2^-24); its streaming stage runs in float64 from now on.' % (cond, cls.LIMIT))
        cls.active = True

    @classmethod
    def use_f64(cls, dev):
        """Called before a float32 training call; True -> run it in float64."""
        if not cls.enabled:
            return False
        if not cls.active:
            c = ops.svgp_cond_nowait(dev)
            if c > cls.LIMIT:
                cls._trip(c)
        return cls.active

    @classmethod
    def after_first_call(cls, dev):
        """Synchronous check after the very first float32 training call; True -> its result must be recomputed in float64."""
        if not cls.enabled or cls._checked_first:
            return False
        cls._checked_first = True
        c = ops.svgp_last_cond(dev)
        if c > cls.LIMIT:
            cls._trip(c)
            return True
        return False


class SVGPLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf.  Gradients are produced for mean_S(logL) (the only reduction the reference applies to a
    module's log-pdf, factor_graph.py:233) and scaled by sum(grad_output) in backward."""

    @staticmethod
    def forward(ctx, kind, ard, jitter, scaling, X, Y, Z, noise, mu, W, sdiag, ls, var):
        want = any(ctx.needs_input_grad[4:])
        S = max(X.shape[0], Y.shape[0])

        def run(cast):
            a = [cast(t) for t in (X, Y, Z[0], noise[0] if noise.dim() == 3 else noise.reshape(-1), mu[0], W[0], sdiag[0], ls.reshape(-1), var.reshape(-1))]
            return ops.svgp_logpdf(kind, *a, ard, jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
        guard = want and X.is_cuda and X.dtype == torch.float32
        if guard and Float32Guard.use_f64(X.device):
            r = {k: (v.float() if v.is_floating_point() else v) for k, v in run(lambda t: t.double()).items()}
        else:
            r = run(lambda t: t)
            if guard and Float32Guard.after_first_call(X.device):
                r = {k: (v.float() if v.is_floating_point() else v) for k, v in run(lambda t: t.double()).items()}
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dZ'], r['dnoise'], r['dmu'], r['dW'], r['dSdiag'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, Z, noise, mu, W, sdiag, ls, var))
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)
        return (None, None, None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[4:], c)


class SVGPSampledLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf_sampled: ONE call for S samples of any of the operands (sampled hyper-parameters, inducing inputs, q(u); the
    reference broadcasts them to S, runtime_variable.py:96-118).  The call returns per-sample gradients; an operand that was shared
    (sample axis 1) receives their sum."""

    @staticmethod
    def forward(ctx, kind, ard, jitter, scaling, X, Y, Z, noise, mu, W, sdiag, ls, var):
        want = any(ctx.needs_input_grad[4:])
        ins = (X, Y, Z, noise, mu, W, sdiag, ls, var)
        S = max(t.shape[0] for t in ins)
        r = ops.svgp_logpdf_sampled(kind, X, Y, Z, noise.reshape(noise.shape[0], 1), mu, W, sdiag, ls, var.reshape(var.shape[0], 1), ard, jitter=jitter,
                                    scaling=scaling, gscale=1.0 / S, want_grad=want)
        if want:
            keys = ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')
            ctx.grads = tuple(r[k].reshape((S,) + tuple(t.shape[1:])) if t.shape[0] == S else r[k].sum(0).reshape(t.shape) for k, t in zip(keys, ins))
            ctx.shapes = tuple(t.shape for t in ins)
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
