"""mxf_kdiag / mxf_gp_predict / mxf_svgp_predict: the prediction algorithms (gp_regression.py:146-196, svgp_regression.py:121-189) and
Kernel.Kdiag as single C-ABI calls, against the oracle -- diagonal and full covariance, with and without the noise term, several samples of
the test inputs against one posterior, every stationary kind, float64 tight and float32 at float32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

KINDS = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}


def _t(a, dt=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda()


def _kp(name, ls, var):
    return {name + '_lengthscale': O.T(ls)[None], name + '_variance': O.T(var)[None]}


@pytest.mark.parametrize('kind', sorted(KINDS))
@pytest.mark.parametrize('ard', [False, True])
@pytest.mark.parametrize('dt,tol', [(torch.float64, 1e-9), (torch.float32, 2e-4)])
def test_gp_predict_matches_oracle(kind, ard, dt, tol):
    from mxfusion_amd import ops
    rng = np.random.RandomState(3)
    N, Nt, Q, P, S = 70, 33, 3, 2, 3
    X, Y, Xt = rng.rand(N, Q), rng.randn(N, P), rng.rand(S, Nt, Q)
    ls, var, noise = (rng.rand(Q) + 0.5 if ard else np.array([0.8])), np.array([1.3]), np.array([0.05])
    k = KINDS[kind](Q, ARD=ard)
    kp = _kp(k.name, ls, var)
    _, post = O.gp_log_pdf(k, O.T(X)[None], O.T(Y)[None], O.T(noise)[None], kp, return_posterior=True)
    for noise_free in (True, False):
        for full in (False, True):
            mr, vr = O.gp_predict(k, O.T(Xt), O.T(noise)[None], post[0][None], post[1][None], post[2][None], kp, noise_free=noise_free,
                                  diagonal_variance=not full)
            m, v = ops.gp_predict(kind, _t(X, dt), _t(Xt, dt), _t(ls, dt), _t(var, dt), ard, _t(post[1].numpy(), dt), _t(post[2].numpy(), dt),
                                  _t(noise, dt), noise_free=noise_free, full_cov=full)
            assert m.shape == (S, Nt, P) and v.shape == ((S, Nt, Nt) if full else (S, Nt))
            assert np.allclose(m.double().cpu().numpy(), mr.numpy(), rtol=tol, atol=tol), (noise_free, full)
            assert np.allclose(v.double().cpu().numpy(), vr.numpy(), rtol=tol, atol=tol), (noise_free, full)


@pytest.mark.parametrize('kind', sorted(KINDS))
@pytest.mark.parametrize('dt,tol', [(torch.float64, 1e-8), (torch.float32, 5e-4)])
def test_svgp_predict_matches_oracle(kind, dt, tol):
    from mxfusion_amd import ops
    rng = np.random.RandomState(5)
    M, Nt, Q, P, S = 40, 50, 2, 2, 2
    # inducing points on a jittered 5 x 8 grid with length-scales below the spacing: a Kuu whose float32 solve is meaningful
    gx, gy = np.meshgrid(np.arange(5.), np.arange(8.))
    Z = np.stack([gx.ravel(), gy.ravel()], -1) + 0.1 * rng.randn(M, Q)
    Xt = rng.rand(S, Nt, Q) * np.array([4., 7.])
    ls, var, noise = 0.3 * rng.rand(Q) + 0.5, np.array([0.9]), np.array([0.02])
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.2
    k = KINDS[kind](Q, ARD=True)
    kp = _kp(k.name, ls, var)
    for noise_free in (True, False):
        for full in (False, True):
            mr, vr = O.svgp_predict(k, O.T(Xt), O.T(Z)[None], O.T(noise)[None], O.T(qm)[None], O.T(qW)[None], O.T(qd)[None], kp, jitter=1e-6,
                                    noise_free=noise_free, diagonal_variance=not full)
            m, v, info = ops.svgp_predict(kind, _t(Z, dt), _t(Xt, dt), _t(ls, dt), _t(var, dt), True, _t(qm, dt), _t(qW, dt), _t(qd, dt),
                                          _t(noise, dt), jitter=1e-6, noise_free=noise_free, full_cov=full)
            assert int(info.abs().sum()) == 0
            # Matern-1/2: sqrt at r = 0 turns the round-off of the reference's expansion-form r^2 (~1e-14 here) into ~1e-7 on the diagonal
            # of Kuu; the library's difference form has none, so the two agree to that level only
            tl = max(tol, 1e-6) if kind == 'matern12' else tol
            assert np.allclose(m.double().cpu().numpy(), mr.numpy(), rtol=tl, atol=tl), (noise_free, full)
            assert np.allclose(v.double().cpu().numpy(), vr.numpy()[..., 0], rtol=tl, atol=tl), (noise_free, full)


def test_predict_composites_agree_with_the_module_algorithms_at_size():
    """N = 1500 conditioning points (the blocked Cholesky / solve paths), 4 samples of 700 test points: the C-ABI composite against the
    module algorithm built from the same posterior."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(11)
    N, Nt, Q, P, S = 1500, 700, 4, 1, 4
    X, Xt = _t(rng.rand(N, Q) * 4), _t(rng.rand(S, Nt, Q) * 4)
    Y = torch.sin(X.sum(-1, keepdim=True))
    ls, var, noise = _t(np.full(Q, 1.2)), _t([1.0]), _t([0.01])
    r = ops.gp_logpdf('rbf', X[None], Y[None], noise[None], ls[None], var[None], True)
    L, LinvY = r['L'][0], r['LinvY'][0]
    m, v = ops.gp_predict('rbf', X, Xt, ls, var, True, L, LinvY, noise, noise_free=False)
    Kxt = ops.gram('rbf', X[None], Xt.reshape(1, S * Nt, Q), ls[None], var[None], True)[0]
    V = torch.linalg.solve_triangular(L.tril(), Kxt, upper=False)
    mref = (V.T @ LinvY).reshape(S, Nt, P)
    vref = (1.0 - (V * V).sum(0) + 0.01).reshape(S, Nt)
    assert torch.allclose(m, mref, rtol=1e-9, atol=1e-10) and torch.allclose(v, vref, rtol=1e-9, atol=1e-10)
    assert float(v.min()) > 0.0


@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_kdiag_every_kind(dt):
    from mxfusion_amd import ops
    rng = np.random.RandomState(2)
    S, N, Q = 3, 37, 4
    X = rng.randn(S, N, Q)
    ls, var = rng.rand(Q) + 0.3, np.array([1.7])
    for kind in ('rbf', 'matern12', 'matern32', 'matern52', 'bias', 'white'):
        out = ops.kdiag(kind, _t(X, dt), _t(ls, dt), _t(var, dt), True)
        assert out.shape == (S, N) and torch.all(out == _t(var, dt))
    out = ops.kdiag('linear', _t(X, dt), _t(ls, dt), None, True)
    ref = O.Linear(Q, ARD=True).Kdiag(O.T(X), linear_variances=O.T(ls)[None]).numpy()
    assert np.allclose(out.double().cpu().numpy(), ref, rtol=1e-6 if dt == torch.float32 else 1e-13)
    # per-sample variances (strideS_var = 1)
    vs = rng.rand(S) + 0.5
    out = ops.kdiag('rbf', _t(X, dt), _t(ls, dt), _t(vs, dt), True)
    assert torch.equal(out, _t(vs, dt)[:, None].expand(S, N))


def test_predict_argument_errors():
    from mxfusion_amd import ops, _lib
    d = lambda *s: torch.zeros(*s, dtype=torch.float64, device='cuda')
    with pytest.raises(_lib.MXFError, match='stationary'):
        ops.gp_predict('linear', d(8, 2), d(1, 4, 2), d(2), d(1), True, d(8, 8), d(8, 1), d(1))
    with pytest.raises(_lib.MXFError, match='null'):
        ops.gp_predict('rbf', d(8, 2), d(1, 4, 2), d(2), d(1), True, d(8, 8), d(8, 1), None, noise_free=False)


@pytest.mark.parametrize('ell', [2.2, 3.0])
def test_float32_svgp_prediction_holds_1e5_at_trained_like_conditioning(ell):
    """north_star: posterior mean / variance within 1e-5 of the reference.  At the condition numbers a trained model has (cond_1(Kuu) 3e4 at
    length-scale 2.2, 1e6 at 3.0; M = 1024, Q = 8) a float32 factorisation of Kuu -- what the reference does in the model's dtype,
    svgp_regression.py:146-154 -- loses cond 2^-24 of the moments.  Since r04 float32 predictions are evaluated in float64 internally:
    the C-ABI call (mxf_svgp_predict) and the module path (SVGPRegressionMeanVariancePrediction) both hold 1e-5 (of the prior variance /
    the largest mean) against the oracle, in all four noise / covariance variants."""
    from mxfusion_amd import ops
    rng = np.random.default_rng(0)
    M, Nt, Q, P, S = 1024, 200, 8, 1, 2
    Z = rng.uniform(-3., 3., (M, Q))
    Xt = rng.uniform(-3., 3., (S, Nt, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, P)), 0.4 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    ls, var, noise = np.full(Q, ell), np.array([1.0]), np.array([0.02])
    f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)            # the oracle sees the float32 VALUES the call receives
    k = O.RBF(Q, ARD=True)
    kp = _kp(k.name, f(ls), f(var))
    dt = torch.float32
    for noise_free in (True, False):
        for full in (False, True):
            mr, vr = O.svgp_predict(k, O.T(f(Xt)), O.T(f(Z))[None], O.T(f(noise))[None], O.T(f(qm))[None], O.T(f(qW))[None], O.T(f(qd))[None], kp, jitter=1e-6,
                                    noise_free=noise_free, diagonal_variance=not full)
            m, v, info = ops.svgp_predict('rbf', _t(Z, dt), _t(Xt, dt), _t(ls, dt), _t(var, dt), True, _t(qm, dt), _t(qW, dt), _t(qd, dt), _t(noise, dt),
                                          jitter=1e-6, noise_free=noise_free, full_cov=full)
            assert int(info.abs().sum()) == 0 and m.dtype == torch.float32
            vr = vr.numpy().reshape(v.shape)
            assert np.abs(m.double().cpu().numpy() - mr.numpy()).max() <= 1e-5 * np.abs(mr.numpy()).max(), (ell, noise_free, full)
            assert np.abs(v.double().cpu().numpy() - vr).max() <= 1e-5 * max(1.0, np.abs(vr).max()), (ell, noise_free, full)
    # the module path
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    t32 = lambda a: _t(a, dt)
    m_ = Model()
    m_.N = Variable()
    m_.X = Variable(shape=(m_.N, Q))
    m_.Z = Variable(shape=(M, Q), initial_value=t32(Z))
    m_.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t32(noise))
    kern = RBF(input_dim=Q, ARD=True, variance=t32(var), lengthscale=t32(ls), dtype='float32')
    m_.Y = SVGPRegression.define_variable(X=m_.X, kernel=kern, noise_var=m_.noise_var, inducing_inputs=m_.Z, shape=(m_.N, P), dtype='float32')
    gp = m_.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    gp.svgp_predict.jitter = 1e-6
    infr = Inference(MAP(model=m_, observed=[m_.X, m_.Y]), dtype='float32')
    infr.initialize(X=(Nt, Q), Y=(Nt, P))
    post = gp._extra_graphs[0]
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t32(qm), t32(qW), t32(qd)
    infr2 = TransferInference(ModulePredictionAlgorithm(m_, observed=[m_.X], target_variables=[m_.Y]), infr_params=infr.params, dtype='float32')
    res = infr2.run(X=t32(Xt[0]))[0]
    # positive parameters pass through softplus(inverse softplus(.)) in float32: compare with the oracle at the values the module really holds
    val = lambda v_: infr.params[v_].double().cpu().numpy()
    kp2 = _kp(k.name, val(kern.lengthscale), val(kern.variance))
    mr, vr = O.svgp_predict(k, O.T(f(Xt[:1])), O.T(f(Z))[None], O.T(val(m_.noise_var))[None], O.T(f(qm))[None], O.T(f(qW))[None], O.T(val(post.qU_cov_diag))[None],
                            kp2, jitter=1e-6, noise_free=True, diagonal_variance=True)
    assert res[0].dtype == torch.float32
    assert np.abs(res[0].double().cpu().numpy() - mr.numpy()).max() <= 1e-5 * np.abs(mr.numpy()).max()
    assert np.abs(res[1].double().cpu().numpy() - vr.numpy()).max() <= 1e-5
