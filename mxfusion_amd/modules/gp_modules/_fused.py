"""autograd bridges to the fused C-ABI composites (value + reverse mode in one call)."""
import torch

from ... import _lib, ops


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
        # r04: a float32 call is EVALUATED in float64 (inputs widened, outputs narrowed).  The reference factors K + noise I in the model's dtype
        # (gp_regression.py:61); in float32 that costs cond(K + noise I) 2^-24 of the bound -- 1e-3 at N = 2048, noise 1e-4 -- and here it is not
        # even faster: the tile-dataflow Cholesky is a float64 kernel, the float32 call took 14.5 ms at N = 8192 against 14.7 in float64.
        if X.is_cuda and X.dtype == torch.float32:
            r = ops.gp_logpdf(kind, X.double(), Y.double(), noise.double(), ls.double(), var.double(), ard, jitter=jitter, want_grad=want)
            r = _narrow(r)
        else:
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
    """Validity of the float32 forms of the SVGP bound -- three levels, PER MODULE.

      EXPLICIT  cond_1(Kuu + jitter I) <= LIMIT            T = H0 Kuf with the explicit inverse H0 = Kuu^-1 - Kuu^-1 Su Kuu^-1: two full-width
                                                           split GEMMs, the fastest form; rounding error ~ cond 2^-24 (ELBO 5e-8 .. 5e-6 up to
                                                           cond ~ 1.4e3, 2e-3 at 5e4: tests/probes/f32_accuracy.py)
      WHITENED  LIMIT < cond <= LIMIT_WHITENED (1e6)       the factorised form the reference evaluates (svgp_regression.py:83-92) on the split
                                                           GEMMs: V = L^-1 Kuf, Phi = V V^T, T = L^-T (I - A_s A_s^T) V (mxf_svgp_configure,
                                                           csrc/whiten.hip): three full-width products, error ~ sqrt(cond) 2^-24
      F64       above, or where the whitened form does     the streaming stage in float64 (inputs widened, outputs narrowed; same C-ABI call with
                not cover the call (shape, combination     dtype = f64) -- the reference's own remedy
                kernels, no-grad evaluations)

    Every SVGP call publishes cond_1 of ITS Kuu into the condition slot its owner holds (pinned host memory, written by the call's last
    launch); the owner polls the slot before its next call -- no synchronisation, it lags by the calls still in flight -- and moves
    between the levels (up at once, down with a factor-4 hysteresis).  The first call of an owner is checked synchronously, so a model
    that STARTS ill-conditioned is never evaluated in a form that cannot hold the 1e-5 bar.  One instance per module algorithm object: a
    two-layer model whose second layer is ill-conditioned keeps its first layer on the fast form.  Calls without an owner share
    Float32Guard.default."""
    EXPLICIT, WHITENED, F64 = 0, 1, 2
    NAMES = ('explicit-inverse float32', 'whitened float32', 'float64')
    LIMIT = 1e3          # (3e3 until r04 late: small / low-rank problems -- M = 64 ... 100, Matern -- showed 2-4e-5 on the bound at 1e3 ... 3e3)
    LIMIT_WHITENED = 1e6
    HYSTERESIS = 0.25
    enabled = True          # class-wide switch (bench.py --no-f32-guard, tests): False = always the explicit float32 form
    force = None            # measurements (bench.py --f32-form): every float32 call at this level, whatever its condition number
    default = None
    _instances = None
    _counter = 0
    epoch = 0               # bumped on every level change of any owner: a holder of a captured hipGraph re-captures when it moved (batch_loop.py)

    def __deepcopy__(self, memo):
        """A cloned module (FactorGraph.clone) gets a guard -- and a condition slot -- of its own."""
        return Float32Guard(self.name)

    def __init__(self, name='svgp'):
        import weakref
        cls = Float32Guard
        if cls._instances is None:
            cls._instances = weakref.WeakSet()
        cls._counter += 1
        # a condition slot no LIVE guard holds (slot 0: calls that never configured one).  More live guards than slots: the surplus share the
        # last slot and are pinned to float64 (`shared_slot`) -- a shared slot's readings cannot be attributed
        used = {g.slot for g in cls._instances}
        free = [k for k in range(1, _lib.COND_SLOTS) if k not in used]
        self.shared_slot = not free
        self.slot = free[0] if free else 0          # surplus guards publish into slot 0 (the un-configured calls' slot, which no guard polls)
        self.name = name
        self.tier = self.EXPLICIT
        self.cond_max = 0.0         # largest condition number this owner has seen
        self.cond_last = 0.0
        self._checked_first = False
        self.switches = 0
        self._range_wide, self._range_by_module, self.range_radius = False, False, 0.0
        cls._instances.add(self)

    # ---- class-level views (reports, tests, backwards compatibility) ----------------------------------------------------------------
    @classmethod
    def instances(cls):
        return list(cls._instances) if cls._instances is not None else []

    @classmethod
    def reset(cls):
        """Every owner back to the explicit form, un-checked; the slots are cleared on their next poll."""
        for g in cls.instances():
            g.tier, g._checked_first, g.cond_max, g.cond_last, g.switches = cls.EXPLICIT, False, 0.0, 0.0, 0
            g._stale = True

    @classmethod
    def poll_all(cls, dev):
        """Fold in what every owner's finished calls have published (no synchronisation); returns the level epoch.  A replayed hipGraph runs
        no Python, so its holder calls this before each replay and re-captures when the epoch moved (the captured launches carry the
        form that was current at capture time)."""
        if cls.enabled and cls.force is None:
            for g in cls.instances():
                g.poll(dev)
        return cls.epoch

    @classmethod
    def report(cls, dev=None):
        """What the guards have seen.  dev given: fold in what the finished calls have published first (the caller has synchronised)."""
        if dev is not None:
            for g in cls.instances():
                if g.shared_slot:             # a surplus guard reads nobody's slot (pinned to float64)
                    continue
                last, mx = _lib.svgp_cond_slot(ops._device_index(dev), g.slot, reset=False)
                if mx > 0 and not getattr(g, '_stale', False):
                    g.cond_max, g.cond_last = max(g.cond_max, mx), last
        gs = [g for g in cls.instances() if g.cond_max > 0]
        # the level an owner's calls really RUN at: an owner without a whitened form (combination kernels: the materialised path) evaluates in
        # float64 as soon as it leaves the explicit form
        eff = lambda g: cls.F64 if ((g.tier != cls.EXPLICIT and getattr(g, 'no_whitened_form', False)) or getattr(g, 'small_f64', False)) else g.tier
        return {'kuu_cond_max': max([g.cond_max for g in gs] or [0.0]),
                'float32_fallback_active': any(eff(g) == cls.F64 for g in gs),
                'float32_whitened_active': any(eff(g) == cls.WHITENED for g in gs),
                'float32_tiers': {('%s#%d' % (g.name, g.slot)): ('float64 inputs' if getattr(g, 'f64_inputs', False) else
                                                                    cls.NAMES[eff(g) if cls.force is None else cls.force] + ('' if cls.force is None else ' (forced)'))
                                  for g in gs},
                'float32_guard': bool(cls.enabled)}

    # ---- per owner -------------------------------------------------------------------------------------------------------------------
    def _target(self, cond):
        t = self.tier
        up = self.EXPLICIT if cond <= self.LIMIT else (self.WHITENED if cond <= self.LIMIT_WHITENED else self.F64)
        if up > t:
            return up
        down = self.EXPLICIT if cond <= self.HYSTERESIS * self.LIMIT else (self.WHITENED if cond <= self.HYSTERESIS * self.LIMIT_WHITENED else self.F64)
        return min(t, down)

    def _move(self, cond):
        self.cond_last = cond
        self.cond_max = max(self.cond_max, cond)
        t = self._target(cond)
        if t != self.tier:
            import warnings
            if t > self.tier:
                warnings.warn('mxfusion_amd: cond_1(Kuu + jitter I) = %.2e of %s: the %s form of the SVGP bound loses accuracy there; its '
                              'streaming stage runs in %s from now on.' % (cond, self.name, self.NAMES[self.tier], self.NAMES[t]))
            self.tier = t
            self.switches += 1
            Float32Guard.epoch += 1
            return True
        return False

    def poll(self, dev):
        """Before a call: fold in what this owner's finished calls have published since the last poll (no synchronisation)."""
        if self.shared_slot:                      # more live guards than slots: this one runs float64 and never consumes an owner's readings
            return
        idx = ops._device_index(dev)
        last, mx = _lib.svgp_cond_slot(idx, self.slot, reset=True)
        if getattr(self, '_stale', False):          # after reset(): whatever was published before does not count
            self._stale = False
            return
        if mx > 0:
            self._move(mx)

    # ---- input range (RBF): the matrix-pipe reverse pass forms r2 = |x|^2 + |z|^2 - 2 x.z from coordinates centred on the inducing inputs and
    # scaled by the length-scales; its absolute error ~1e-7 (|x|^2 + |z|^2) costs the gradients 1e-4 at a radius of ~80 length-scales -- a long one-dimensional series
    # with a short length-scale.  Checked on an owner's first call (synchronously) and every RANGE_EVERY calls after that (device reduction,
    # copied to pinned host memory without synchronising, read by a later call); too wide -> the owner's float32 calls run in float64.
    RANGE_LIMIT = 100.0         # (tests/probes/range_accuracy.py: gradients 3.5e-5 at a radius of 40 length-scales, 1.4e-4 at 80, 6e-4 at 160; the bound itself stays at 1e-7)
    RANGE_EVERY = 32

    def range_too_wide(self, X, Z, ls):
        if not (Float32Guard.enabled and Float32Guard.force is None and X.is_cuda):
            self._range_wide = False
            return False
        n = getattr(self, '_range_calls', 0)
        self._range_calls = n + 1
        if getattr(self, '_range_host', None) is None:
            self._range_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._range_wide = False
        if n % self.RANGE_EVERY == 0:
            with torch.no_grad():
                c = Z[..., :64, :].mean(-2, keepdim=True)
                r = (((X - c) / ls.reshape(ls.shape[0], 1, -1)) ** 2).sum(-1).amax().sqrt().float().reshape(1)      # Euclidean radius in length-scales
            if n == 0:
                self._range_host[0] = float(r)           # (an owner's first call synchronises anyway: first_call_needs_rerun)
            else:
                self._range_host.copy_(r, non_blocking=True)
        val = float(self._range_host[0])
        wide = val > (0.8 * self.RANGE_LIMIT if self._range_wide else self.RANGE_LIMIT)
        if wide and not self._range_wide:
            import warnings
            warnings.warn('mxfusion_amd: the inputs of %s span %.0f length-scales around the inducing inputs: the float32 reverse pass of the RBF '
                          'kernel loses accuracy there (DESIGN.md section 5); its calls run in float64 from now on.' % (self.name, val))
            Float32Guard.epoch += 1
        self._range_wide = wide
        self.range_radius = val
        return wide

    def form(self, dev, whitened_ok):
        """The level this call runs at (enabled guards only; the caller holds float32 CUDA inputs)."""
        if not Float32Guard.enabled:
            return self.EXPLICIT
        if getattr(self, 'shared_slot', False):          # no condition slot of its own: cannot be watched, runs where every call is valid
            return self.F64
        self.poll(dev)
        if Float32Guard.force is not None:
            self._checked_first = True
            return self.F64 if (Float32Guard.force == self.WHITENED and not whitened_ok) else Float32Guard.force
        if self.tier == self.WHITENED and not whitened_ok:
            return self.F64
        return self.tier

    def configure(self, dev, tier):
        _lib.svgp_configure(ops._device_index(dev), _lib.FORM_WHITENED if tier == self.WHITENED else _lib.FORM_EXPLICIT, self.slot)

    def first_call_needs_rerun(self, dev, ran_at, whitened_ok):
        """Synchronous check after an owner's very first call; True -> its result must be recomputed at the (higher) level now set."""
        if not Float32Guard.enabled or self._checked_first:
            return False
        self._checked_first = True
        c = ops.svgp_last_cond(dev)                  # (synchronises the device: the slot below is complete)
        # a call over S samples of the hyper-parameters publishes one condition number per sample: act on their MAXIMUM, not on the last one
        _, mx = _lib.svgp_cond_slot(ops._device_index(dev), self.slot, reset=True)
        self._move(max(c, mx))
        need = self.F64 if (self.tier == self.WHITENED and not whitened_ok) else self.tier
        return need > ran_at


def _guarded(guard, dev, is_f32, whitened_ok, run, wide=False):
    """Run `run(tier)` under `guard`: picks the level, configures the handle (form + condition slot), re-runs an owner's first call when its
    synchronous check asks for a higher level.  run(tier) evaluates the call in float32 (EXPLICIT / WHITENED) or widened to float64 (F64)."""
    g = guard if guard is not None else Float32Guard.default
    g.f64_inputs = not is_f32 and not getattr(g, '_widened_by_owner', False)
    try:
        if not is_f32:
            g.configure(dev, g.EXPLICIT)
            return run(g.EXPLICIT)
        tier = g.form(dev, whitened_ok)
        if wide:                    # (range_too_wide: float64 whatever the condition number says)
            tier = g.F64
        g.configure(dev, tier)
        r = run(tier)
        if g.first_call_needs_rerun(dev, tier, whitened_ok) and not getattr(g, '_owner_reruns', False):
            tier = g.F64 if (wide or (g.tier == g.WHITENED and not whitened_ok)) else g.tier
            g.configure(dev, tier)
            r = run(tier)
        return r
    finally:
        # the form and the condition slot are state of the per-(thread, device) HANDLE: left behind, a later direct ops.svgp_logpdf call
        # (bench, tests, user code) would run the whitened form -- or fail on a shape it does not cover -- and publish its condition number
        # into this owner's slot
        _lib.svgp_configure(ops._device_index(dev), _lib.FORM_EXPLICIT, 0)


def _narrow(r):
    return {k: (v.float() if v.is_floating_point() else v) for k, v in r.items()}


class SVGPLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf.  Gradients are produced for mean_S(logL) (the only reduction the reference applies to a
    module's log-pdf, factor_graph.py:233) and scaled by sum(grad_output) in backward.  `guard`: the owner's Float32Guard (None = the
    shared default)."""

    @staticmethod
    def forward(ctx, guard, kind, ard, jitter, scaling, X, Y, Z, noise, mu, W, sdiag, ls, var):
        want = any(ctx.needs_input_grad[5:])
        S = max(X.shape[0], Y.shape[0])

        def run(tier):
            cast = (lambda t: t.double()) if tier == Float32Guard.F64 else (lambda t: t)
            a = [cast(t) for t in (X, Y, Z[0], noise[0] if noise.dim() == 3 else noise.reshape(-1), mu[0], W[0], sdiag[0], ls.reshape(-1), var.reshape(-1))]
            r = ops.svgp_logpdf(kind, *a, ard, jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
            return _narrow(r) if tier == Float32Guard.F64 else r
        if X.is_cuda:
            is_f32 = X.dtype == torch.float32
            homo = noise.numel() == 1
            # the whitened form covers float32 TRAINING calls of the streaming (homoscedastic) path on split-capable shapes
            wok = is_f32 and want and homo and _lib.svgp_whitened_ok(_lib.F32, S, X.shape[-2], Z.shape[-2], X.shape[-1], Y.shape[-1],
                                                                      0 if X.shape[0] == 1 else X.shape[-2] * X.shape[-1])
            g_ = guard if guard is not None else Float32Guard.default
            wide = bool(is_f32 and want and kind == 'rbf' and g_ is not None and (g_._range_wide if getattr(g_, '_range_by_module', False) else g_.range_too_wide(X, Z, ls)))
            r = _guarded(guard, X.device, is_f32, wok, run, wide=wide)
        else:
            r = run(Float32Guard.EXPLICIT)
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dZ'], r['dnoise'], r['dmu'], r['dW'], r['dSdiag'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, Z, noise, mu, W, sdiag, ls, var))
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)
        return (None, None, None, None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[5:], c)


class SVGPSampledLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf_sampled: ONE call for S samples of any of the operands (sampled hyper-parameters, inducing inputs, q(u); the
    reference broadcasts them to S, runtime_variable.py:96-118).  The call returns per-sample gradients; an operand that was shared
    (sample axis 1) receives their sum.  Every sample publishes the condition number of its own Kuu into the owner's slot (the guard acts
    on their maximum)."""

    @staticmethod
    def forward(ctx, guard, kind, ard, jitter, scaling, X, Y, Z, noise, mu, W, sdiag, ls, var):
        want = any(ctx.needs_input_grad[5:])
        ins = (X, Y, Z, noise, mu, W, sdiag, ls, var)
        S = max(t.shape[0] for t in ins)

        def run(tier):
            cast = (lambda t: t.double()) if tier == Float32Guard.F64 else (lambda t: t)
            r = ops.svgp_logpdf_sampled(kind, cast(X), cast(Y), cast(Z), cast(noise.reshape(noise.shape[0], 1)), cast(mu), cast(W), cast(sdiag), cast(ls),
                                        cast(var.reshape(var.shape[0], 1)), ard, jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
            return _narrow(r) if tier == Float32Guard.F64 else r
        if X.is_cuda:
            is_f32 = X.dtype == torch.float32
            wok = is_f32 and want and _lib.svgp_whitened_ok(_lib.F32, 1, X.shape[-2], Z.shape[-2], X.shape[-1], Y.shape[-1], 0)   # (one sample per inner call)
            g_ = guard if guard is not None else Float32Guard.default
            wide = bool(is_f32 and want and kind == 'rbf' and g_ is not None and (g_._range_wide if getattr(g_, '_range_by_module', False) else g_.range_too_wide(X, Z, ls)))
            r = _guarded(guard, X.device, is_f32, wok, run, wide=wide)
        else:
            r = run(Float32Guard.EXPLICIT)
        if want:
            keys = ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')
            ctx.grads = tuple(r[k].reshape((S,) + tuple(t.shape[1:])) if t.shape[0] == S else r[k].sum(0).reshape(t.shape) for k, t in zip(keys, ins))
            ctx.shapes = tuple(t.shape for t in ins)
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)
        return (None, None, None, None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[5:], c)


class SVGPMatLogPdfFn(torch.autograd.Function):
    """mxf_svgp_logpdf_mat: the bound from materialised Kuu / Kuf / Kdiag (one sample of the inputs: unit sample axes) for S >= 1 samples of Y;
    the gradients of mean_S(logL) flow on into the kernels' own reverse mode (combination kernels).  This path has the explicit float32
    form and float64 only: above Float32Guard.LIMIT the guard widens it (the Grams it receives stay float32 values)."""

    @staticmethod
    def forward(ctx, guard, jitter, scaling, Kuu, Kuf, Kdiag, Y, noise, mu, W, sdiag):
        want = any(ctx.needs_input_grad[3:])
        S = Y.shape[0]

        def run(tier):
            cast = (lambda t: t.double()) if tier == Float32Guard.F64 else (lambda t: t)
            r = ops.svgp_logpdf_mat(cast(Kuu[0]), cast(Kuf[0]), cast(Kdiag[0]), cast(Y if S > 1 else Y[0]), cast(noise[0] if noise.dim() == 3 else noise.reshape(-1)),
                                    cast(mu[0]), cast(W[0]), cast(sdiag[0]), jitter=jitter, scaling=scaling, gscale=1.0 / S, want_grad=want)
            return _narrow(r) if tier == Float32Guard.F64 else r
        if Kuu.is_cuda:
            r = _guarded(guard, Kuu.device, Kuu.dtype == torch.float32, False, run)
        else:
            r = run(Float32Guard.EXPLICIT)
        if want:
            ctx.grads = (r['dKuu'], r['dKuf'], r['dKdiag'], r['dY'], r['dnoise'], r['dmu'], r['dW'], r['dSdiag'])
            ctx.shapes = tuple(t.shape for t in (Kuu, Kuf, Kdiag, Y, noise, mu, W, sdiag))
            ctx.S = S
        ctx.mark_non_differentiable(r['info'])
        return r['logL'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = _uniform_weight(g)          # gradients were produced for mean_S(logL) (gscale = 1/S): scale by sum(grad_output), as SVGPLogPdfFn does
        return (None, None, None) + _scaled(ctx.grads, ctx.shapes, ctx.needs_input_grad[3:], c)


class SGPLogPdfFn(torch.autograd.Function):
    """mxf_sgp_logpdf for ONE sample (arrays carry a unit sample axis); returns logL (1,), wv, L, LA.  The float32 form of the Titsias
    bound has the same conditioning limit as the explicit SVGP form (a float32 Psi2 inside C = Kuu + Psi2 / s2 and K^-1 Psi2 K^-1: ELBO
    7e-6 at cond_1(Kuu) 3e4, a non-PD C at 1e6) and no whitened form: above Float32Guard.LIMIT the owner's guard widens the call."""

    @staticmethod
    def forward(ctx, guard, kind, ard, jitter, X, Y, Z, noise, ls, var):
        want = any(ctx.needs_input_grad[4:])

        def run(tier):
            cast = (lambda t: t.double()) if tier == Float32Guard.F64 else (lambda t: t)
            r = ops.sgp_logpdf(kind, cast(X[0]), cast(Y[0]), cast(Z[0]), cast(noise.reshape(-1)), cast(ls.reshape(-1)), cast(var.reshape(-1)), ard, jitter=jitter,
                               gscale=1.0, want_grad=want)
            return _narrow(r) if tier == Float32Guard.F64 else r
        r = _guarded(guard, X.device, X.dtype == torch.float32, False, run) if X.is_cuda else run(Float32Guard.EXPLICIT)
        if want:
            ctx.grads = (r['dX'], r['dY'], r['dZ'], r['dnoise'], r['dls'], r['dvar'])
            ctx.shapes = tuple(t.shape for t in (X, Y, Z, noise, ls, var))
        ctx.mark_non_differentiable(r['wv'], r['L'], r['LA'], r['info'])
        return r['logL'], r['wv'], r['L'], r['LA'], r['info']

    @staticmethod
    def backward(ctx, g, *_):
        c = g.sum()
        out = [(grad.reshape(shp) * c) if need else None for grad, shp, need in zip(ctx.grads, ctx.shapes, ctx.needs_input_grad[4:])]
        return (None, None, None, None) + tuple(out)


Float32Guard.default = Float32Guard('svgp (shared default)')
