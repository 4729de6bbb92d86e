"""The float32 guard of the SVGP module (modules/gp_modules/_fused.py: Float32Guard): three levels PER MODULE --
explicit-inverse float32 up to cond_1(Kuu + jitter I) = 1e3, WHITENED float32 (the reference's factorised form, svgp_regression.py:83-92, on
the split GEMMs) up to 1e6, float64 above or where the whitened form does not apply (no-grad evaluations, combination kernels).  The first
call of an owner is checked synchronously; later calls through the condition number every finished call publishes into the owner's slot
of pinned host memory (mxf_svgp_cond_slot, no synchronisation).  Checked against the ORACLE at B = 8192, M = 1024."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('float32_forms_on_small_problems')]

from oracle import gp_oracle as O  # noqa: E402


def _inputs(ell, seed=0):
    rng = np.random.default_rng(seed)
    B, Q, M = 8192, 8, 1024
    X = rng.uniform(-3., 3., (1, B, Q))
    Y = np.sin(X[0] @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((B, 1))
    Z = rng.uniform(-3., 3., (M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, 1)), 0.4 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    return dict(X=X, Y=Y[None], Z=Z[None], noise=np.array([[0.02]]), qm=qm[None], qW=qW[None], qd=qd[None], ls=np.full((1, Q), ell), var=np.array([[1.0]]))


_ORACLE = {}


def _oracle(a, ell):
    if ell not in _ORACLE:
        T = O.T
        _ORACLE[ell] = float(O.svgp_log_pdf(O.RBF(8, ARD=True), T(a['X']), T(a['Y']), T(a['Z']), T(a['noise']), T(a['qm']), T(a['qW']), T(a['qd']),
                                            {'rbf_lengthscale': T(a['ls']), 'rbf_variance': T(a['var'])}, jitter=1e-6)[0])
    return _ORACLE[ell]


def _module_call(a, guard=None, grad=True):
    """The module's bridge (SVGPRegressionLogPdf.compute -> SVGPLogPdfFn) in float32, as a training step (grad) or an evaluation calls it."""
    from mxfusion_amd.modules.gp_modules._fused import SVGPLogPdfFn
    t = {k: torch.as_tensor(v, dtype=torch.float32).cuda().requires_grad_(grad and k not in ('Y',)) for k, v in a.items()}
    logL, info = SVGPLogPdfFn.apply(guard, 'rbf', True, 1e-6, 1.0, t['X'], t['Y'], t['Z'], t['noise'], t['qm'], t['qW'], t['qd'], t['ls'], t['var'])
    if grad:
        logL.sum().backward()
        assert t['Z'].grad.dtype == torch.float32
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0 and logL.dtype == torch.float32
    return float(logL[0])


def test_ill_conditioned_start_is_never_evaluated_in_a_form_that_cannot_hold_it():
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    G = Float32Guard
    # length-scale 2.6, cond_1 ~ 2e5 (3.0 gives 1.1e6, beyond the whitened limit of 1e6 since r04 late): the explicit form fails, the whitened float32 form holds north_star's bar -- through the float32 API, first call
    a = _inputs(2.6)
    ref = _oracle(a, 2.6)
    g = G('t1')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = _module_call(a, g)
    assert g.tier == G.WHITENED and any('whitened' in str(x.message) for x in w), (g.tier, g.cond_max)
    assert G.LIMIT < g.cond_max < G.LIMIT_WHITENED
    assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    # cond_1 ~ 3e7: float64
    a4 = _inputs(4.0)
    ref4 = _oracle(a4, 4.0)
    g4 = G('t2')
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        got4 = _module_call(a4, g4)
    assert g4.tier == G.F64 and g4.cond_max > G.LIMIT_WHITENED
    assert abs(got4 - ref4) <= 1e-5 * abs(ref4), (got4, ref4)
    # the plain float32 call (guard bypassed) is what the guard protects from: it misses the bar at both
    G.enabled = False
    try:
        raw = _module_call(a, G('t3'))
    finally:
        G.enabled = True
    assert abs(raw - ref) > 1e-5 * abs(ref)
    rep = G.report()
    assert rep['float32_fallback_active'] and rep['float32_whitened_active'] and rep['kuu_cond_max'] > G.LIMIT_WHITENED


def test_guard_moves_without_synchronising_and_comes_back():
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G
    good, bad = _inputs(1.0), _inputs(2.6)
    ref_good, ref_bad = _oracle(good, 1.0), _oracle(bad, 2.6)
    g = G('drift')
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        v = _module_call(good, g)                                        # first call: checked synchronously, well conditioned
        assert g.tier == G.EXPLICIT and abs(v - ref_good) <= 1e-5 * abs(ref_good) and 0 < g.cond_max < G.LIMIT
        _module_call(bad, g)                                             # the parameters have drifted: this call still runs in the explicit form ...
        assert g.tier == G.EXPLICIT
        b = _module_call(bad, g)                                         # ... the next one sees what it published (no host sync) and runs whitened
        assert g.tier == G.WHITENED and abs(b - ref_bad) <= 1e-5 * abs(ref_bad), (g.tier, b, ref_bad)
        _module_call(good, g)                                            # back to a benign Kuu: one whitened call publishes it ...
        v2 = _module_call(good, g)                                       # ... and the owner returns to the fast form
        assert g.tier == G.EXPLICIT and abs(v2 - ref_good) <= 1e-5 * abs(ref_good)
    assert g.switches == 2


def test_guard_is_per_module_and_covers_evaluations():
    """Two owners on one handle (the two layers of a deep GP): the ill-conditioned one moves to the whitened form, the other stays on the
    fast form (the r03 guard was process-wide and sticky).  A no-grad evaluation above the limit runs in float64."""
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G
    good, bad = _inputs(1.0, seed=1), _inputs(2.6)
    ref_bad = _oracle(bad, 2.6)
    g1, g2 = G('layer1'), G('layer2')
    assert g1.slot != g2.slot
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        for _ in range(3):
            _module_call(good, g1)
            b = _module_call(bad, g2)
        assert g1.tier == G.EXPLICIT and g2.tier == G.WHITENED
        assert g1.cond_max < G.LIMIT < g2.cond_max
        assert abs(b - ref_bad) <= 1e-5 * abs(ref_bad)
        ev = _module_call(bad, g2, grad=False)                           # evaluation: no whitened form -> float64
        assert abs(ev - ref_bad) <= 1e-5 * abs(ref_bad)
        g3 = G('eval-first')
        ev3 = _module_call(bad, g3, grad=False)                          # ... also as an owner's very first call
        assert abs(ev3 - ref_bad) <= 1e-5 * abs(ref_bad) and g3.tier == G.WHITENED


def test_two_layer_model_keeps_the_first_layer_on_the_fast_form():
    """A two-layer deep GP through the API in float32: the second layer's 256 inducing points in a 2-D hidden space
    make its Kuu ill-conditioned (cond_1 ~ 2e5: the whitened form), the first layer's Kuu is benign (cond_1 ~ 8e2: stays on the
    explicit form).  The float32 objective matches the oracle's to 1e-5; the inference reports the level of each module."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, BatchInferenceLoop, create_Gaussian_meanfield
    DT = 'float32'
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    t = lambda a: torch.as_tensor(f32(a)).cuda()
    rng = np.random.RandomState(5)
    N, Q, Dh, M, S = 1024, 8, 2, 256, 2
    Z1 = f32(rng.uniform(-2, 2, (M, Dh)))
    Z0 = f32(rng.uniform(-2, 2, (M, Q)))
    X = f32(rng.uniform(-2, 2, (N, Q)))
    Y = f32(np.sin(X[:, :1]) + 0.1 * rng.randn(N, 1))
    hm, hv = f32(rng.uniform(-2, 2, (N, Dh))), f32(np.full((N, Dh), 0.01))
    eps = f32(rng.randn(S, N, Dh))
    qm0, qW0, qd0 = f32(rng.randn(M, Dh) * 0.3), f32(rng.randn(M, M) * 0.02), f32(rng.rand(M) * 0.3 + 0.2)
    qm1, qW1, qd1 = f32(rng.randn(M, 1) * 0.3), f32(rng.randn(M, M) * 0.02), f32(rng.rand(M) * 0.3 + 0.2)
    p = dict(ls0=f32([1.5]), v0=f32([1.0]), ls1=f32([0.2, 0.2]), v1=f32([1.0]), n0=f32([0.05]), n1=f32([0.05]))
    k0 = RBF(Q, variance=t(p['v0']), lengthscale=t(p['ls0']), name='rbf_bottom', dtype=DT)
    k1 = RBF(Dh, ARD=True, variance=t(p['v1']), lengthscale=t(p['ls1']), name='rbf_top', dtype=DT)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z0 = Variable(shape=(M, Q), initial_value=t(Z0))
    m.Z1 = Variable(shape=(M, Dh), initial_value=t(Z1))
    m.noise0 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t(p['n0']))
    m.noise1 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t(p['n1']))
    m.H = SVGPRegression.define_variable(X=m.X, kernel=k0, noise_var=m.noise0, inducing_inputs=m.Z0, shape=(m.N, Dh), dtype=DT)
    m.Y = SVGPRegression.define_variable(X=m.H, kernel=k1, noise_var=m.noise1, inducing_inputs=m.Z1, shape=(m.N, 1), dtype=DT)
    g0, g1 = m.H.factor, m.Y.factor
    g0.svgp_log_pdf.jitter = g1.svgp_log_pdf.jitter = 1e-6
    q = create_Gaussian_meanfield(model=m, observed=[m.X, m.Y], dtype=DT)
    qH = q[m.H].factor
    losses = []

    class Rec(BatchInferenceLoop):
        def run(self, infr_executor, data, **kw):
            def wrapped(*a):
                qH._rand_gen = MockRandomGenerator(t(eps.reshape(-1)))        # the same draw every step
                out = infr_executor(*a)
                losses.append(float(out[0].detach()))
                return out
            return super(Rec, self).run(wrapped, data, **kw)
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.X, m.Y]), grad_loop=Rec(), dtype=DT)
    infr.initialize(X=X.shape, Y=Y.shape)
    for gp, (a, b, c) in ((g0, (qm0, qW0, qd0)), (g1, (qm1, qW1, qd1))):
        infr.params[gp._extra_graphs[0].qU_mean], infr.params[gp._extra_graphs[0].qU_cov_W], infr.params[gp._extra_graphs[0].qU_cov_diag] = t(a), t(b), t(c)
    infr.params[qH.mean], infr.params[qH.variance] = t(hm), t(hv)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        infr.run(X=t(X), Y=t(Y), max_iter=3, learning_rate=0.0)       # three evaluations of the SAME objective (learning rate 0)
    # ---- the oracle on the float32 values the modules received (positive parameters pass through softplus(inverse softplus(.)) in float32)
    T = O.T
    rt = lambda a: T(np.asarray(torch.nn.functional.softplus(torch.log(torch.expm1(torch.as_tensor(f32(a))))).numpy(), dtype=np.float64))
    # (the hidden draw itself in float32, as the module forms it: at length-scale 0.2 on [-2, 2] the top layer's Gram amplifies the float32
    #  rounding of its INPUTS by r / l ~ 1e2 -- an input effect no float32 evaluation can avoid, kept out of the comparison)
    sp32 = torch.nn.functional.softplus(torch.log(torch.expm1(torch.as_tensor(hv))))
    Hs = T((torch.as_tensor(hm)[None] + torch.as_tensor(eps) * torch.sqrt(sp32)[None]).numpy())
    l0 = O.svgp_log_pdf(O.RBF(Q, name='rbf_bottom'), T(X)[None], Hs, T(Z0)[None], rt(p['n0'])[None], T(qm0)[None], T(qW0)[None], rt(qd0)[None],
                        {'rbf_bottom_lengthscale': rt(p['ls0'])[None], 'rbf_bottom_variance': rt(p['v0'])[None]}, jitter=1e-6)
    l1 = O.svgp_log_pdf(O.RBF(Dh, ARD=True, name='rbf_top'), Hs, T(Y)[None], T(Z1)[None], rt(p['n1'])[None], T(qm1)[None], T(qW1)[None], rt(qd1)[None],
                        {'rbf_top_lengthscale': rt(p['ls1'])[None], 'rbf_top_variance': rt(p['v1'])[None]}, jitter=1e-6)
    logq = O.normal_log_pdf(T(hm)[None], rt(hv)[None], Hs).reshape(S, -1).sum(-1)
    ref = float(-(l0 + l1 - logq).mean())
    tiers = infr.float32_tiers
    by_name = {k.split('#')[0]: v for k, v in tiers.items()}
    assert len(tiers) == 2 and sorted(tiers.values()) == sorted([G.NAMES[G.EXPLICIT], G.NAMES[G.WHITENED]]), tiers
    assert not infr.float32_fallback_active and G.LIMIT < infr.last_kuu_condition < G.LIMIT_WHITENED
    # step 1 is already evaluated in the right forms (first calls are checked synchronously); so are the later ones
    for l in losses[:3]:
        assert abs(l - ref) <= 1e-5 * abs(ref), (losses, ref, by_name)


def test_replayed_hipgraph_follows_the_guard():
    """BatchInferenceLoop(use_graph=True) replays captured launches without running any Python, so a captured step keeps the float32 form
    that was current at capture time.  The loop polls the guards before every replay (Float32Guard.poll_all: no synchronisation) and
    re-captures when a level changed: after the length-scale moves from 1 to 2.6 (cond_1 14 -> 2e5) the replayed step is the WHITENED one
    and its loss matches the oracle again."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
    DT = 'float32'
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    t = lambda a: torch.as_tensor(f32(a)).cuda()
    rng = np.random.default_rng(3)
    B, Q, M = 2048, 8, 256
    X = f32(rng.uniform(-3., 3., (B, Q)))
    Y = f32(np.sin(X @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((B, 1)))
    Z = f32(rng.uniform(-3., 3., (M, Q)))
    qm, qW, qd = f32(0.3 * rng.standard_normal((M, 1))), f32(0.4 * rng.standard_normal((M, M)) / np.sqrt(M)), f32(rng.uniform(0.05, 0.5, M))
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=t(Z))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t([0.02]))
    kern = RBF(input_dim=Q, ARD=True, variance=t([1.0]), lengthscale=t(np.ones(Q)), dtype=DT)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=DT)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    loop = BatchInferenceLoop(use_graph=True)
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=DT)
    infr.initialize(X=X.shape, Y=Y.shape)
    post = gp._extra_graphs[0]
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t(qm), t(qW), t(qd)
    ex = infr.create_executor()
    data = [t(X), t(Y)]

    def oracle(ell):
        T = O.T
        val = lambda v_: infr.params[v_].double().cpu().numpy()
        return -float(O.svgp_log_pdf(O.RBF(Q, ARD=True), T(X)[None], T(Y)[None], T(Z)[None], T(val(m.noise_var))[None], T(qm)[None], T(qW)[None],
                                     T(val(post.qU_cov_diag))[None], {'rbf_lengthscale': T(val(kern.lengthscale))[None], 'rbf_variance': T(val(kern.variance))[None]},
                                     jitter=1e-6)[0])

    def step():
        infr.params.zero_grad()
        loss = loop.step(ex, data, infr.params)
        torch.cuda.synchronize()
        return float(loss)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        for _ in range(4):
            l1 = step()                                   # two eager warm-ups, capture, replay
        assert 'graph' in loop._gstate
        g = gp.svgp_log_pdf._f32_guard()
        assert g.tier == G.EXPLICIT and abs(l1 - oracle(1.0)) <= 1e-5 * abs(l1)
        infr.params[kern.lengthscale] = t(np.full(Q, 2.6))      # written in place: the captured graph reads the new values
        ref3 = oracle(2.6)
        step()                                            # the OLD (explicit) graph runs once more and publishes cond ~ 1e6
        losses = [step() for _ in range(3)]               # poll -> level change -> eager whitened warm-up -> capture -> replay
        assert g.tier == G.WHITENED and 'graph' in loop._gstate
        for l in losses:
            assert abs(l - ref3) <= 1e-5 * abs(ref3), (losses, ref3)


def test_inputs_spanning_many_length_scales_run_in_float64():
    """A long one-dimensional series with a short length-scale (inputs spanning +-200 length-scales around the inducing inputs): the float32
    reverse pass of the RBF kernel forms r2 from norms and loses accuracy there, so the owner's calls run in float64 (checked synchronously on
    its first call); the bound and the gradients hold the float64 bar against the oracle.  The same series with a length-scale that covers
    it stays in float32."""
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G, SVGPLogPdfFn
    rng = np.random.default_rng(2)
    B, M, Q = 1024, 128, 1
    X = np.sort(rng.uniform(0., 400., (1, B, Q)), axis=1)
    Y = (np.sin(X[0] / 7.0) + 0.05 * rng.standard_normal((B, 1)))[None]
    Z = np.linspace(0., 400., M)[:, None]
    qm, qW, qd = 0.3 * rng.standard_normal((M, 1)), 0.1 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    for ell, expect_wide in ((1.0, True), (40.0, False)):          # (radius 200 / 5 length-scales; the limit is 100)
        vals = {k: r32(v) for k, v in dict(X=X, Y=Y, Z=Z[None], noise=[[0.05]], qm=qm[None], qW=qW[None], qd=qd[None], ls=[[ell]], var=[[1.1]]).items()}
        g = G('range-%g' % ell)
        t = {k: torch.as_tensor(v, dtype=torch.float32).cuda().requires_grad_(k != 'Y') for k, v in vals.items()}
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            logL, info = SVGPLogPdfFn.apply(g, 'rbf', True, 1e-6, 1.0, t['X'], t['Y'], t['Z'], t['noise'], t['qm'], t['qW'], t['qd'], t['ls'], t['var'])
            logL.sum().backward()
        torch.cuda.synchronize()
        assert g._range_wide == expect_wide, (ell, g.range_radius)
        assert any('length-scales' in str(x.message) for x in w) == expect_wide
        if expect_wide:
            assert g.range_radius > 150
            k = O.RBF(Q, ARD=True)
            lv = {n: O.T(v).clone().requires_grad_(True) for n, v in vals.items()}
            ref = O.svgp_log_pdf(k, lv['X'], lv['Y'], lv['Z'], lv['noise'], lv['qm'], lv['qW'], lv['qd'], {'rbf_lengthscale': lv['ls'], 'rbf_variance': lv['var']}, jitter=1e-6)
            gX, gZ = torch.autograd.grad(ref.sum(), [lv['X'], lv['Z']])
            assert abs(float(logL[0]) - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
            nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / np.linalg.norm(b.ravel()))
            assert nrm(t['X'].grad.double().cpu().numpy(), gX.numpy()) <= 1e-4 and nrm(t['Z'].grad.double().cpu().numpy(), gZ.numpy()) <= 1e-4


@pytest.mark.parametrize('mode', ['disabled', 'forced-whitened', 'forced-explicit'])
def test_module_call_with_the_guard_disabled_or_forced(mode):
    """bench.py --no-f32-guard / --f32-form: the module's training call with the guard switched off or pinned to one form (the input-range check
    is skipped there as well) -- runs, and agrees with the guarded call on a benign problem."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard as G
    rng = np.random.default_rng(3)
    B, M, Q = 512, 128, 4
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()[None]
    X, Z = rng.uniform(-2, 2, (B, Q)), rng.uniform(-2, 2, (M, Q))
    Y = np.sin(X.sum(1))[:, None]
    args = lambda: (t(X).requires_grad_(True), t(Y), t(Z), t([0.05]), t(0.1 * rng.standard_normal((M, 1))), t(np.zeros((M, M))), t(np.ones(M)))
    kern = RBF(input_dim=Q, ARD=True, dtype='float32')
    params = {kern.name + '_lengthscale': t(np.full(Q, 0.5)), kern.name + '_variance': t([1.0])}          # (cond ~ 10: every form holds the bar)

    def call():
        fn = SVGPRegressionLogPdf.__new__(SVGPRegressionLogPdf)
        fn.jitter, fn.log_pdf_scaling, fn._guard = 1e-6, 1.0, G('mode-' + mode)
        a = args()
        out = fn._compute_columns(None, *a, kern, params)
        out.sum().backward()
        return float(out[0]), a[0].grad.clone()
    rng = np.random.default_rng(3)
    ref, gref = call()
    rng = np.random.default_rng(3)
    old = (G.enabled, G.force)
    try:
        if mode == 'disabled':
            G.enabled = False
        else:
            G.force = G.WHITENED if mode == 'forced-whitened' else G.EXPLICIT
        got, g = call()
    finally:
        G.enabled, G.force = old
    assert abs(got - ref) <= 1e-5 * abs(ref)
    assert float((g - gref).norm() / gref.norm()) <= 1e-3
