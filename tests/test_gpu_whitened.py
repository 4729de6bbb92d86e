"""The WHITENED float32 tier of the SVGP training call (mxf_svgp_configure(MXF_SVGP_WHITENED); csrc/whiten.hip, composite.hip): the
factorised form the reference evaluates (svgp_regression.py:83-92: trsm with the Cholesky factor) on the split GEMMs --
V = L^-1 Kuf (triangular product written directly as f16 planes), Phi = V V^T, T = L^-T (I - A_s A_s^T) V, U = (L^-1 mu)^T V.
Checked against the ORACLE where the explicit-inverse float32 form fails (cond_1(Kuu) 1e6 .. 3e7), and piecewise (the chained split
products) against float64."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('float32_forms_on_small_problems')]

from oracle import gp_oracle as O  # noqa: E402


def _dev(a, dt=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda()


@pytest.mark.parametrize('M,N,K,lower', [(256, 512, 256, False), (256, 512, 256, True), (128, 256, 128, True), (512, 768, 512, True), (384, 256, 384, False),
                                         # r05: triangular products walk PAIRS of column strips (ascending / descending k): four row tiles, three 128-row
                                         # tiles, an odd strip count (the last pair has one strip)
                                         (1024, 2048, 1024, True), (384, 1024, 384, True), (1024, 1280, 1024, True)])
def test_planes_output_product_and_transposition(M, N, K, lower):
    """mxf_gemm_f16x2_planes_out: alpha A B^T as f16 planes == the float64 product to f32 accuracy (also with a triangular A whose k loop
    is cut short); mxf_f16x2_planes_transpose: the planes of the transpose hold the same values, and its fused U = scale a^T X."""
    from mxfusion_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K))
    if lower:
        A = np.tril(A)
    Bm = rng.uniform(0, 1, (N, K)) ** 3
    ref = A @ Bm.T
    alpha = 8192.0 / np.abs(ref).max()
    pl = ops.gemm_f16x2_planes_out(ops.f16x2_split(_dev(A)), ops.f16x2_split(_dev(Bm)), M, N, K, alpha=alpha, a_lower=lower)
    got = ops.planes_to_dense(pl, M, N).cpu().numpy() / alpha
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 3e-6, err
    a = rng.standard_normal(M).astype(np.float32)
    plT, U = ops.f16x2_planes_transpose(pl, M, N, a=_dev(a), scale=_dev([0.5]))
    gotT = ops.planes_to_dense(plT, N, M).cpu().numpy()
    assert np.array_equal(gotT, ops.planes_to_dense(pl, M, N).cpu().numpy().T)
    Uref = 0.5 * a.astype(np.float64) @ (got * alpha)
    assert np.abs(U.cpu().numpy() - Uref).max() <= 1e-5 * np.abs(Uref).max()
    # the same launch can write the transposed planes and U itself (what the SVGP step uses)
    pl2, plT2, U2 = ops.gemm_f16x2_planes_out(ops.f16x2_split(_dev(A)), ops.f16x2_split(_dev(Bm)), M, N, K, alpha=alpha, a_lower=lower, a=_dev(a))
    assert torch.equal(pl2, pl) and torch.equal(plT2, plT)
    assert np.abs(U2.cpu().numpy() - 2.0 * Uref).max() <= 1e-5 * np.abs(2.0 * Uref).max()
    # the planes feed a following split product: Phi = X X^T
    w = torch.full((1,), 8192.0, dtype=torch.float32).cuda().view(torch.int32)      # a max word in [2^13, 2^14) = scale 1: the planes are unscaled
    Phi = ops.gemm_f16x2_planes((pl, w), (pl, w), M, M, N).cpu().numpy()
    Pref = (got * alpha) @ (got * alpha).T
    assert np.abs(np.tril(Phi) - np.tril(Pref)).max() <= 3e-6 * np.abs(Pref).max()


def _inputs(B, M, Q, P, ell, S=1, seed=0, kind='rbf'):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3., 3., (S, B, Q))
    Y = np.sin(X[0] @ rng.standard_normal((Q, P))) + 0.05 * rng.standard_normal((B, P))
    Z = rng.uniform(-3., 3., (M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, P)), 0.4 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    return dict(X=X, Y=Y[None], Z=Z, noise=np.array([0.02]), qm=qm, qW=qW, qd=qd, ls=np.full(Q, ell), var=np.array([1.3]))


def _oracle(a, kind='rbf'):
    T = O.T
    k = {'rbf': O.RBF, 'matern52': O.Matern52, 'matern32': O.Matern32, 'matern12': O.Matern12}[kind](a['X'].shape[-1], ARD=True)
    names = ('X', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    lv = {n: T(a[n]).clone().requires_grad_(True) for n in names}
    logL = O.svgp_log_pdf(k, lv['X'], T(a['Y']), lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6)
    g = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    return logL.detach().numpy(), dict(zip(('dX', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar'), [x.numpy() for x in g]))


def _run(a, form, kind='rbf', dt=torch.float32):
    from mxfusion_amd import _lib, ops
    _lib.svgp_configure(torch.cuda.current_device(), form, 5)
    try:
        d = lambda x: _dev(x, dt)
        S = a['X'].shape[0]
        r = ops.svgp_logpdf(kind, d(a['X']), d(a['Y']), d(a['Z']), d(a['noise']), d(a['qm']), d(a['qW']), d(a['qd']), d(a['ls']), d(a['var']), True,
                            jitter=1e-6, gscale=1.0 / S, want_grad=True)
        torch.cuda.synchronize()
    finally:
        _lib.svgp_configure(torch.cuda.current_device(), _lib.FORM_EXPLICIT, 0)
    assert int(r['info'].abs().sum()) == 0
    return {k: v.double().cpu().numpy() for k, v in r.items()}


def _nrm(a, b):
    return float(np.linalg.norm(a.ravel() - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


@pytest.mark.parametrize('B,M,Q,P,S,kind', [(512, 128, 8, 1, 1, 'rbf'), (256, 256, 3, 1, 2, 'rbf'), (512, 128, 8, 2, 1, 'rbf'), (768, 128, 12, 1, 1, 'rbf'),
                                            (512, 256, 8, 1, 1, 'matern52'), (256, 128, 5, 3, 2, 'matern32'), (768, 384, 8, 1, 1, 'rbf'), (128, 128, 4, 1, 2, 'matern12'),
                                            (1024, 640, 6, 1, 1, 'rbf')])
def test_whitened_form_matches_the_oracle_small(B, M, Q, P, S, kind):
    """Values and all gradients of the whitened float32 call against the oracle's autograd, across the launch shapes (one / several output
    columns: fused / separate U; Q <= 8 / > 8: blocked / row-major T; 128- and 256-row tiles; sampled inputs), at a length-scale where Kuu
    is moderately ill-conditioned."""
    from mxfusion_amd import _lib
    a = _inputs(B, M, Q, P, 2.0 if Q > 4 else 0.6, S=S, seed=B + M + Q + P, kind=kind)
    ref, gref = _oracle(a, kind)
    got = _run(a, _lib.FORM_WHITENED, kind)
    assert np.abs(got['logL'] - ref).max() <= 1e-5 * np.abs(ref).max(), (got['logL'], ref)
    for k, g in gref.items():
        assert _nrm(got[k], g) <= 2e-3, (k, _nrm(got[k], g))
    # and the float64 call is untouched by the configured form
    g64 = _run(a, _lib.FORM_WHITENED, kind, torch.float64)
    assert np.abs(g64['logL'] - ref).max() <= 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize('ell', [2.2, 3.0, 4.0])
def test_whitened_form_holds_the_bar_where_the_explicit_form_fails(ell):
    """B = 8192, M = 1024, Q = 8 (the inputs of test_gpu_f32_guard.py): cond_1(Kuu) ~ 3e4 / 1e6 / 3e7.  Whitened float32: ELBO to 1e-5 (north_star)
    and every gradient to 1e-3 normwise against the oracle (5e-3 at 3e7, beyond the guard's whitened range); the explicit-inverse form misses the ELBO bar from ell = 3 on and its dW is two
    digits worse already at 2.2 (Ki G Ki with a float32 Psi2: error ~ cond^2)."""
    from mxfusion_amd import _lib, ops
    a = _inputs(8192, 1024, 8, 1, ell, seed=0)
    a['var'] = np.array([1.0])
    ref, gref = _oracle(a)
    w = _run(a, _lib.FORM_WHITENED)
    cond = ops.svgp_last_cond()
    e = _run(a, _lib.FORM_EXPLICIT)
    rel_w, rel_e = abs(w['logL'][0] - ref[0]) / abs(ref[0]), abs(e['logL'][0] - ref[0]) / abs(ref[0])
    errs_w = {k: _nrm(w[k], g) for k, g in gref.items()}
    errs_e = {k: _nrm(e[k], g) for k, g in gref.items()}
    print('ell %.1f cond_1 %.2e  ELBO rel: whitened %.2e explicit %.2e\n  whitened %s\n  explicit %s' % (
        ell, cond, rel_w, rel_e, {k: '%.1e' % v for k, v in errs_w.items()}, {k: '%.1e' % v for k, v in errs_e.items()}))
    assert rel_w <= 1e-5, (rel_w, rel_e)
    for k, v in errs_w.items():      # (ell = 4, cond 3e7, lies beyond Float32Guard.LIMIT_WHITENED = 1e6: the guard would run float64 there)
        assert v <= (1e-3 if ell <= 3.0 else 5e-3), (k, v, errs_e[k])
    if ell >= 3.0:
        assert rel_e > 1e-5
    assert errs_w['dW'] < errs_e['dW']


def test_whitened_form_at_the_bench_shape_vs_oracle():
    """BASELINE configs[2]'s shape for one sample (N = 65 536, M = 1 024, Q = 8) at length-scale 3 (cond_1(Kuu) ~ 1e6): the whitened float32
    call against the oracle's value and autograd gradients -- ELBO to 1e-5, every gradient to 1e-3 normwise."""
    from mxfusion_amd import _lib
    a = _inputs(65536, 1024, 8, 1, 3.0, seed=0)
    a['var'] = np.array([1.0])
    ref, gref = _oracle(a)
    w = _run(a, _lib.FORM_WHITENED)
    rel = abs(w['logL'][0] - ref[0]) / abs(ref[0])
    errs = {k: _nrm(w[k], g) for k, g in gref.items()}
    print('N 65536: whitened ELBO rel %.2e' % rel, {k: '%.1e' % v for k, v in errs.items()})
    assert rel <= 1e-5, rel
    for k, v in errs.items():
        assert v <= 1e-3, (k, v)


@pytest.mark.parametrize('M,ell,B', [(1000, 1.0, 4096), (1000, 2.2, 4096), (200, 1.5, 4096), (512, 1.0, 4001), (1000, 2.2, 3000)])
def test_inducing_point_counts_that_are_not_tile_multiples_are_padded_onto_the_split_path(M, ell, B):
    """M = 1000 (or 200) inducing points through the module in float32: the bridge pads them to 1024 (256) with decoupled points (far away,
    q(u) = prior on them), the call runs the split-GEMM path (explicit or whitened by the guard) -- and the bound and the gradients of the
    REAL parameters equal the oracle's for the unpadded model (1e-5 / 2e-3)."""
    import warnings
    from mxfusion_amd import Model, Variable, _lib
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    t = lambda a: torch.as_tensor(f32(a)).cuda()
    rng = np.random.default_rng(M)
    Q = 8            # (B = 4001 / 3000: the data rows are padded too -- to the next multiple of 256, with the padded rows' closed-form contribution subtracted)
    X = f32(rng.uniform(-3., 3., (B, Q)))
    Y = f32(np.sin(X @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((B, 1)))
    Z = f32(rng.uniform(-3., 3., (M, Q)))
    qm, qW, qd = f32(0.3 * rng.standard_normal((M, 1))), f32(0.4 * rng.standard_normal((M, M)) / np.sqrt(M)), f32(rng.uniform(0.05, 0.5, M))
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=t(Z))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t([0.02]))
    kern = RBF(input_dim=Q, ARD=True, variance=t([1.3]), lengthscale=t(np.full(Q, ell)), dtype='float32')
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype='float32')
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=BatchInferenceLoop(), dtype='float32')
    infr.initialize(X=X.shape, Y=Y.shape)
    post = gp._extra_graphs[0]
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t(qm), t(qW), t(qd)
    ex = infr.create_executor()
    dev = torch.cuda.current_device()
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        ex(t(X), t(Y))                                   # (first call: the guard's synchronous check may move the level)
        infr.params.zero_grad()
        _lib.svgp_timing(dev, True)
        try:
            loss, lfg = ex(t(X), t(Y))
            lfg.backward()
            stages = _lib.svgp_timing_read(dev)
        finally:
            _lib.svgp_timing(dev, False)
    g = gp.svgp_log_pdf._f32_guard()
    assert 'planes_a' in stages and 't_gemm' in stages, (stages, g.tier, g.cond_max, g.cond_last)          # the split path, although M % 16 != 0
    # the oracle on the UNPADDED model, at the values the module holds
    val = lambda v_: infr.params[v_].double().cpu().numpy()
    names = ('Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in zip(names, (Z, val(m.noise_var), qm, qW, val(post.qU_cov_diag), val(kern.lengthscale), val(kern.variance)))}
    ref = -O.svgp_log_pdf(O.RBF(Q, ARD=True), O.T(X)[None], O.T(Y)[None], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {'rbf_lengthscale': lv['ls'][None], 'rbf_variance': lv['var'][None]}, jitter=1e-6)[0]
    gref = dict(zip(names, torch.autograd.grad(ref, [lv[n] for n in names])))
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref)), (float(loss), float(ref))
    fg = infr.params.flat.grad

    def grad_of(var):
        o, n, shape = infr.params._slices[var.uuid]
        return fg[o:o + n].view(shape).double().cpu().numpy()
    for var, n in ((m.Z, 'Z'), (post.qU_mean, 'qm'), (post.qU_cov_W, 'qW')):
        assert grad_of(var).shape == gref[n].shape
        assert _nrm(grad_of(var), gref[n].numpy()) <= 2e-3, (n, _nrm(grad_of(var), gref[n].numpy()))
    for var, n in ((kern.lengthscale, 'ls'), (kern.variance, 'var'), (m.noise_var, 'noise'), (post.qU_cov_diag, 'qd')):     # softplus-transformed: d/draw = d/dvalue * sigmoid(raw)
        raw = infr.params.raw(var).double().cpu().numpy()
        assert _nrm(grad_of(var), gref[n].numpy().reshape(raw.shape) / (1.0 + np.exp(-raw))) <= 2e-3, n


def test_padding_keeps_the_gradient_of_latent_inputs():
    """Latent inputs (X requires grad) with both paddings in play (M = 200 -> 256, B = 1000 -> 1024): the padded points' coordinates stay inside
    the f16 range of the reverse pass (a 1e6-style offset gives inf * 0 there), dX of the real rows equals the oracle's and has the unpadded shape."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    rng = np.random.default_rng(77)
    B, M, Q = 1000, 200, 4
    X = rng.uniform(-2., 2., (B, Q))
    Y = np.sin(X.sum(1))[:, None] + 0.05 * rng.standard_normal((B, 1))
    Z = rng.uniform(-2., 2., (M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, 1)), 0.3 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    ls, var, noise = np.full(Q, 1.2), np.array([1.1]), np.array([0.05])
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()[None]
    kern = RBF(input_dim=Q, ARD=True, dtype='float32')
    fn = SVGPRegressionLogPdf.__new__(SVGPRegressionLogPdf)
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    fn.jitter, fn.log_pdf_scaling, fn._guard = 1e-6, 1.0, Float32Guard('padding-test')
    Xd, Zd = t(X).requires_grad_(True), t(Z).requires_grad_(True)
    params = {kern.name + '_lengthscale': t(ls), kern.name + '_variance': t(var)}
    got = fn._compute_columns(None, Xd, t(Y), Zd, t(noise), t(qm), t(qW), t(qd), kern, params)
    gX, gZ = torch.autograd.grad(got.sum(), [Xd, Zd])
    Xo, Zo = O.T(X).clone().requires_grad_(True), O.T(Z).clone().requires_grad_(True)
    ref = O.svgp_log_pdf(O.RBF(Q, ARD=True), Xo[None], O.T(Y)[None], Zo[None], O.T(noise)[None], O.T(qm)[None], O.T(qW)[None], O.T(qd)[None],
                         {'rbf_lengthscale': O.T(ls)[None], 'rbf_variance': O.T(var)[None]}, jitter=1e-6)[0]
    rX, rZ = torch.autograd.grad(ref, [Xo, Zo])
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (float(got), float(ref))
    assert gX.shape == (1, B, Q) and gZ.shape == (1, M, Q)
    assert torch.isfinite(gX).all() and torch.isfinite(gZ).all()
    assert _nrm(gX[0].double().cpu().numpy(), rX.numpy()) <= 2e-3
    assert _nrm(gZ[0].double().cpu().numpy(), rZ.numpy()) <= 2e-3


@pytest.mark.parametrize('kind,S,P,B,M', [('matern32', 1, 1, 1000, 200), ('matern12', 1, 2, 700, 130), ('rbf', 2, 1, 900, 300), ('matern52', 2, 3, 520, 100)])
def test_padding_with_other_kernels_output_columns_and_sampled_parameters(kind, S, P, B, M):
    """Both paddings for every stationary kind (the padded points must be far enough for the slowest-decaying kernel, exp(-r): 128
    length-scales), several output columns, and sampled inputs / length-scales / variance / inducing inputs (S = 2: the sampled-operand call;
    the padded coordinates are shared by the samples, placed beyond the largest range and length-scale): bound and gradients vs the oracle."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern12, Matern32, Matern52
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    rng = np.random.default_rng(B + M)
    Q = 3
    X = rng.uniform(-2., 2., (S, B, Q))
    Y = np.sin(X[0] @ rng.standard_normal((Q, P))) + 0.05 * rng.standard_normal((B, P))
    Z = rng.uniform(-2., 2., (S, M, Q)) if S > 1 else rng.uniform(-2., 2., (1, M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, P)), 0.3 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    ls = rng.uniform(0.8, 1.4, (S, Q))
    var, noise = rng.uniform(0.9, 1.3, (S, 1)), np.array([[0.05]])
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()
    kern = {'rbf': RBF, 'matern12': Matern12, 'matern32': Matern32, 'matern52': Matern52}[kind](input_dim=Q, ARD=True, dtype='float32')
    fn = SVGPRegressionLogPdf.__new__(SVGPRegressionLogPdf)
    fn.jitter, fn.log_pdf_scaling, fn._guard = 1e-6, 1.0, Float32Guard('padding-test-' + kind)
    leaves = {n: t(v).requires_grad_(True) for n, v in (('X', X), ('Z', Z), ('noise', noise), ('qm', qm[None]), ('qW', qW[None]), ('qd', qd[None]), ('ls', ls), ('var', var))}
    got = fn._compute_columns(None, leaves['X'], t(Y)[None], leaves['Z'], leaves['noise'], leaves['qm'], leaves['qW'], leaves['qd'], kern,
                              {kern.name + '_lengthscale': leaves['ls'], kern.name + '_variance': leaves['var']})
    names = ('X', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    ggot = torch.autograd.grad(got.mean(), [leaves[n] for n in names])
    ok = {'rbf': O.RBF, 'matern52': O.Matern52, 'matern32': O.Matern32, 'matern12': O.Matern12}[kind](Q, ARD=True)
    lo = {n: O.T(v).clone().requires_grad_(True) for n, v in (('X', X), ('Z', Z), ('noise', noise), ('qm', qm[None]), ('qW', qW[None]), ('qd', qd[None]), ('ls', ls), ('var', var))}
    ref = O.svgp_log_pdf(ok, lo['X'], O.T(Y)[None], lo['Z'], lo['noise'], lo['qm'], lo['qW'], lo['qd'],
                         {ok.name + '_lengthscale': lo['ls'], ok.name + '_variance': lo['var']}, jitter=1e-6)
    gref = torch.autograd.grad(ref.mean(), [lo[n] for n in names])
    assert got.shape == ref.shape
    assert np.abs(got.detach().double().cpu().numpy() - ref.detach().numpy()).max() <= 1e-5 * np.abs(ref.detach().numpy()).max()
    for n, a, b in zip(names, ggot, gref):
        assert a.shape == b.shape, (n, a.shape, b.shape)
        assert torch.isfinite(a).all(), n
        assert _nrm(a.double().cpu().numpy(), b.numpy()) <= 2e-3, (n, _nrm(a.double().cpu().numpy(), b.numpy()))


@pytest.mark.parametrize('kind,Q', [('matern12', 3), ('matern12', 8), ('matern32', 5), ('matern52', 5), ('rbf', 5)])
def test_float32_gradients_with_inducing_inputs_next_to_data_points(kind, Q):
    """Z = (a subset of X) + 1e-3 noise -- what Z = X[:M] looks like after a few optimiser steps: many (x_n, z_m) pairs at distance ~1e-3.
    The Matern slopes are singular / kinked at r = 0 (dk/dr2 = -k / 2r for Matern12): with expansion-form distances in float32 (the matrix-pipe
    reverse pass, r2 = |x|^2 + |z|^2 - 2 x.z) dX and dZ were 10-20 % off for Matern12; the Matern kinds therefore take the difference-form pass
    (r04).  Every gradient against the oracle's autograd, normwise."""
    from mxfusion_amd import _lib
    rng = np.random.default_rng(5)
    B, M = 2048, 128
    X = rng.uniform(-2., 2., (1, B, Q))
    Y = (np.sin(X[0] @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1)))[None]
    Z = X[0, rng.permutation(B)[:M]] + 1e-3 * rng.standard_normal((M, Q))
    a = dict(X=X, Y=Y, Z=Z, noise=np.array([0.05]), qm=0.3 * rng.standard_normal((M, 1)), qW=0.3 * rng.standard_normal((M, M)) / np.sqrt(M),
             qd=rng.uniform(0.05, 0.5, M), ls=np.full(Q, 0.3 * np.sqrt(Q)), var=np.array([1.1]))
    a = {k: np.asarray(v, dtype=np.float32).astype(np.float64) for k, v in a.items()}          # the oracle sees the float32-representable inputs
    ref, gref = _oracle(a, kind)
    got = _run(a, _lib.FORM_EXPLICIT, kind)
    assert abs(got['logL'][0] - ref[0]) <= 1e-5 * abs(ref[0])
    for k, g in gref.items():
        assert _nrm(got[k], g) <= 5e-4, (k, _nrm(got[k], g))


@pytest.mark.parametrize('offset', [100.0, 1000.0])
def test_float32_training_call_does_not_depend_on_where_the_inputs_sit(offset):
    """Inputs at an offset (years, raw sensor readings: X, Z = offset + U(-2, 2)).  The matrix-pipe reverse pass forms r2 = |x|^2 + |z|^2 - 2 x.z
    in float32; un-centred that gave 1e-2 gradient errors at an offset of 100 and 98 % (2e-2 on the bound) at 1000.  Both operands are centred on
    the inducing inputs now (translation invariance): bound and gradients against the oracle as for centred data."""
    from mxfusion_amd import _lib
    rng = np.random.default_rng(9)
    B, M, Q = 2048, 128, 5
    X = offset + rng.uniform(-2., 2., (1, B, Q))
    Y = (np.sin((X[0] - offset) @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1)))[None]
    Z = offset + rng.uniform(-2., 2., (M, Q))
    a = dict(X=X, Y=Y, Z=Z, noise=np.array([0.05]), qm=0.3 * rng.standard_normal((M, 1)), qW=0.3 * rng.standard_normal((M, M)) / np.sqrt(M),
             qd=rng.uniform(0.05, 0.5, M), ls=np.full(Q, 0.3 * np.sqrt(Q)), var=np.array([1.1]))
    a = {k: np.asarray(v, dtype=np.float32).astype(np.float64) for k, v in a.items()}
    ref, gref = _oracle(a)
    got = _run(a, _lib.FORM_EXPLICIT)
    assert abs(got['logL'][0] - ref[0]) <= 1e-5 * abs(ref[0])
    for k, g in gref.items():
        assert _nrm(got[k], g) <= 2e-4, (k, _nrm(got[k], g))


def test_padding_with_inputs_at_a_large_offset():
    """Both paddings with inputs at an offset of 1e5 units (M = 200 -> 256, B = 1000 -> 1024): the padded points are placed from the edge of
    the data, not from the origin, so that they stay inside the f16 range of the (centred) reverse pass -- bound and gradients against the
    oracle on the centred inputs."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    rng = np.random.default_rng(78)
    B, M, Q, off = 1000, 200, 4, 1.0e5
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = r32(off + rng.uniform(-2., 2., (B, Q)))
    Y = r32(np.sin((X - off).sum(1))[:, None] + 0.05 * rng.standard_normal((B, 1)))
    Z = r32(off + rng.uniform(-2., 2., (M, Q)))
    qm, qW, qd = r32(0.3 * rng.standard_normal((M, 1))), r32(0.3 * rng.standard_normal((M, M)) / np.sqrt(M)), r32(rng.uniform(0.05, 0.5, M))
    ls, var, noise = r32(np.full(Q, 1.2)), r32([1.1]), r32([0.05])
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()[None]
    kern = RBF(input_dim=Q, ARD=True, dtype='float32')
    fn = SVGPRegressionLogPdf.__new__(SVGPRegressionLogPdf)
    fn.jitter, fn.log_pdf_scaling, fn._guard = 1e-6, 1.0, Float32Guard('padding-offset')
    Xd, Zd = t(X).requires_grad_(True), t(Z).requires_grad_(True)
    got = fn._compute_columns(None, Xd, t(Y), Zd, t(noise), t(qm), t(qW), t(qd), kern, {kern.name + '_lengthscale': t(ls), kern.name + '_variance': t(var)})
    gX, gZ = torch.autograd.grad(got.sum(), [Xd, Zd])
    Xo, Zo = O.T(X - off).clone().requires_grad_(True), O.T(Z - off).clone().requires_grad_(True)
    ref = O.svgp_log_pdf(O.RBF(Q, ARD=True), Xo[None], O.T(Y)[None], Zo[None], O.T(noise)[None], O.T(qm)[None], O.T(qW)[None], O.T(qd)[None],
                         {'rbf_lengthscale': O.T(ls)[None], 'rbf_variance': O.T(var)[None]}, jitter=1e-6)[0]
    rX, rZ = torch.autograd.grad(ref, [Xo, Zo])
    assert torch.isfinite(gX).all() and torch.isfinite(gZ).all()
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (float(got), float(ref))
    assert _nrm(gX[0].double().cpu().numpy(), rX.numpy()) <= 2e-3 and _nrm(gZ[0].double().cpu().numpy(), rZ.numpy()) <= 2e-3
