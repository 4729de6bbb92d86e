"""Mean-function parity of the three GP modules: the reference's test_log_pdf_w_mean / test_prediction_w_mean
(testing/modules/gpregression_test.py:98-118,228-254, svgpregression_test.py:117-141,257-300, sparsegpregression_test.py:101-135,198-240):
a mean function with TRAINABLE parameters in front of the module -- there a gluon Dense(D, tanh), here the same map as an
MXFusionFunction of the inputs and two parameter Variables -- on the reference tests' seeded inputs (tests/golden/kat_*.npz).
Against the oracle's `mean=` arguments in float64: the log-pdf through Inference(MAP), its gradient with respect to the mean's parameters
(and the kernel's) through the inference's own flat gradient, and the predictive mean / variance through TransferInference in all four
noise / covariance variants."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

DT = 'float64'


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()


def _mean_fn(x, W, b):
    """Dense(D, activation='tanh', flatten=False) on arrays with a leading sample axis: x (S,N,3), W (1|S,3,D), b (1|S,D)."""
    return torch.tanh(torch.matmul(x, W) + b.unsqueeze(-2))


def _model(which, g, D, W0, b0):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.components.functions import MXFusionFunction
    from mxfusion_amd.modules.gp_modules import GPRegression, SVGPRegression, SparseGPRegression
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(g['noise']))
    m.mean_W = Variable(shape=(3, D), initial_value=_t(W0))
    m.mean_b = Variable(shape=(D,), initial_value=_t(b0))
    m.mean_func = MXFusionFunction(_mean_fn)
    mean = m.mean_func(m.X, m.mean_W, m.mean_b)
    kernel = RBF(input_dim=3, ARD=True, variance=_t(g['var']), lengthscale=_t(g['ls']), dtype=DT)
    if which == 'gp':
        m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, mean=mean, noise_var=m.noise_var, shape=(m.N, D), dtype=DT)
        m.Y.factor.gp_log_pdf.jitter = 1e-6
    else:
        m.Z = Variable(shape=(3, 3), initial_value=_t(g['Z']))
        cls = SVGPRegression if which == 'svgp' else SparseGPRegression
        m.Y = cls.define_variable(X=m.X, kernel=kernel, mean=mean, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, D), dtype=DT)
        getattr(m.Y.factor, 'svgp_log_pdf' if which == 'svgp' else 'sgp_log_pdf').jitter = 1e-8
    return m


def _oracle_logpdf(which, g, W, b, lead):
    """lead: dict of float64 leaves (requires_grad) the oracle differentiates: W, b, ls, var, noise."""
    T = O.T
    k = O.RBF(3, ARD=True)
    X, Y = T(g['X'])[None], T(g['Y'])[None]
    kp = {'rbf_lengthscale': lead['ls'][None], 'rbf_variance': lead['var'][None]}
    mean = _mean_fn(X, lead['W'][None], lead['b'][None])
    if which == 'gp':
        return O.gp_log_pdf(k, X, Y, lead['noise'][None], kp, jitter=1e-6, mean=mean)
    if which == 'svgp':
        return O.svgp_log_pdf(k, X, Y, T(g['Z'])[None], lead['noise'][None], T(g['qm'])[None], T(g['qW'])[None], T(g['qd'])[None], kp, jitter=1e-8, mean=mean)
    return O.sgp_log_pdf(k, X, Y, T(g['Z'])[None], lead['noise'][None], kp, jitter=1e-8, mean=mean)


@pytest.mark.parametrize('which', ['gp', 'svgp', 'sgp'])
def test_log_pdf_and_prediction_with_a_trainable_mean_function(golden_dir, which):
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    g = dict(np.load(os.path.join(golden_dir, {'gp': 'kat_gp.npz', 'svgp': 'kat_svgp.npz', 'sgp': 'kat_sgp.npz'}[which])))
    D = g['Y'].shape[1]
    rng = np.random.RandomState(7)
    W0, b0 = rng.randn(3, D) * 0.8, rng.randn(D) * 0.3
    m = _model(which, g, D, W0, b0)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.initialize(X=g['X'].shape, Y=g['Y'].shape)
    gp = m.Y.factor
    if which == 'svgp':
        post = gp._extra_graphs[0]
        infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = _t(g['qm']), _t(g['qW']), _t(g['qd'])
    # ---- value ------------------------------------------------------------------------------------------------------------------
    lead = {n: O.T(v).clone().requires_grad_(True) for n, v in (('W', W0), ('b', b0), ('ls', g['ls']), ('var', g['var']), ('noise', g['noise']))}
    ref = _oracle_logpdf(which, g, W0, b0, lead)
    loss, _ = infr.run(X=_t(g['X']), Y=_t(g['Y']))
    assert abs(float(-loss) - float(ref[0])) <= 1e-9 * abs(float(ref[0])), (float(-loss), float(ref[0]))
    # the mean function matters on these inputs (a test that passes with the mean ignored proves nothing)
    ref0 = _oracle_logpdf(which, g, W0, b0, dict(lead, W=lead['W'] * 0, b=lead['b'] * 0 + 10.0))
    assert abs(float(ref0[0]) - float(ref[0])) > 1e-2 * abs(float(ref[0]))
    # ---- gradient with respect to the mean's parameters (and the kernel's), through the inference's own executor ------------------
    gref = dict(zip(lead, torch.autograd.grad(ref.sum(), list(lead.values()))))
    ex = infr.create_executor()
    infr.params.zero_grad()
    loss2, lfg = ex(_t(g['X']), _t(g['Y']))
    lfg.backward()
    flat_grad = infr.params.flat.grad
    def grad_of(var):
        o, n, shape = infr.params._slices[var.uuid]
        return flat_grad[o:o + n].view(shape).cpu().numpy()
    # loss = -logL; W, b are unconstrained; noise / ls / var are optimised through softplus: d/draw = d/dvalue * sigmoid(raw)
    assert np.allclose(-grad_of(m.mean_W), gref['W'].numpy(), rtol=1e-7, atol=1e-10), (grad_of(m.mean_W), gref['W'])
    assert np.allclose(-grad_of(m.mean_b), gref['b'].numpy(), rtol=1e-7, atol=1e-10)
    kern = gp.kernel
    for var, name in ((kern.lengthscale, 'ls'), (kern.variance, 'var'), (m.noise_var, 'noise')):
        raw = infr.params.raw(var).cpu().numpy()
        assert np.allclose(-grad_of(var).ravel(), (gref[name].numpy() / (1.0 + np.exp(-raw))).ravel(), rtol=1e-6, atol=1e-9), name
    # ---- prediction: the mean of the test inputs is added, the variance is that of the zero-mean model on Y - m(X) -----------------
    T = O.T
    k = O.RBF(3, ARD=True)
    kp = {'rbf_lengthscale': T(g['ls'])[None], 'rbf_variance': T(g['var'])[None]}
    Xt = T(g['Xt'])[None]
    mean_t = _mean_fn(Xt, T(W0)[None], T(b0)[None])
    mean_x = _mean_fn(T(g['X'])[None], T(W0)[None], T(b0)[None])
    alg = {'gp': 'gp_predict', 'svgp': 'svgp_predict', 'sgp': 'sgp_predict'}[which]
    for nf in (True, False):
        for dg in (True, False):
            if which == 'gp':
                _, (Xc, L, LinvY) = O.gp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['noise'])[None], kp, jitter=1e-6, mean=mean_x, return_posterior=True)
                mu_r, var_r = O.gp_predict(k, Xt, T(g['noise'])[None], Xc[None], L[None], LinvY[None], kp, mean=mean_t, noise_free=nf, diagonal_variance=dg)
            elif which == 'svgp':
                mu_r, var_r = O.svgp_predict(k, Xt, T(g['Z'])[None], T(g['noise'])[None], T(g['qm'])[None], T(g['qW'])[None], T(g['qd'])[None], kp, jitter=0.,
                                             mean=mean_t, noise_free=nf, diagonal_variance=dg)
            else:
                _, (wv, L, LA) = O.sgp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['Z'])[None], T(g['noise'])[None], kp, jitter=1e-8, mean=mean_x,
                                               return_posterior=True)
                mu_r, var_r = O.sgp_predict(k, Xt, T(g['Z'])[None], T(g['noise'])[None], L[None], LA[None], wv[None], kp, mean=mean_t, noise_free=nf,
                                            diagonal_variance=dg)
            infr2 = TransferInference(ModulePredictionAlgorithm(m, observed=[m.X], target_variables=[m.Y]), infr_params=infr.params, dtype=DT)
            getattr(gp, alg).noise_free = nf
            getattr(gp, alg).diagonal_variance = dg
            res = infr2.run(X=_t(g['Xt']))[0]
            tag = which + (' nf' if nf else ' noisy') + (' diag' if dg else ' full')
            assert np.allclose(res[0].cpu().numpy(), mu_r.numpy(), rtol=1e-8, atol=1e-10), tag
            assert np.allclose(res[1].cpu().numpy(), var_r.numpy(), rtol=1e-8, atol=1e-10), tag
    getattr(gp, alg).noise_free, getattr(gp, alg).diagonal_variance = True, True


def test_gp_log_pdf_baseline_config_1_literal_shape():
    """BASELINE.json configs[0]: GPRegression, RBF, N = 512, D = 2, exact marginal log-likelihood -- the literal shape (r03 bracketed it with
    N = 200 / 700 / 2240), float64 to 1e-9 and float32 to 1e-5 against the oracle, values and gradients."""
    from mxfusion_amd import ops
    rng = np.random.default_rng(512)
    N, Q = 512, 2
    X = rng.uniform(-3., 3., (1, N, Q))
    Y = np.sin(X[0] @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((N, 1))
    noise, ls, var = np.array([[0.01]]), np.array([[1.0]]), np.array([[1.0]])
    T = O.T
    names = ('X', 'Y', 'noise', 'ls', 'var')
    lv = {n: T(v).clone().requires_grad_(True) for n, v in zip(names, (X, Y[None], noise, ls, var))}
    ref = O.gp_log_pdf(O.RBF(Q, ARD=False), lv['X'], lv['Y'], lv['noise'], {'rbf_lengthscale': lv['ls'], 'rbf_variance': lv['var']}, jitter=0.)
    gref = dict(zip(('dX', 'dY', 'dnoise', 'dls', 'dvar'), torch.autograd.grad(ref.sum(), [lv[n] for n in names])))
    for dt, tol, gtol in ((torch.float64, 1e-9, 1e-7), (torch.float32, 1e-5, 5e-3)):
        d = lambda a: torch.as_tensor(np.asarray(a), dtype=dt).cuda()
        r = ops.gp_logpdf('rbf', d(X), d(Y[None]), d(noise), d(ls), d(var), False, jitter=0., want_grad=True)
        torch.cuda.synchronize()
        assert int(r['info'].abs().sum()) == 0
        got = float(r['logL'][0])
        assert abs(got - float(ref[0])) <= tol * abs(float(ref[0])), (dt, got, float(ref[0]))
        for kk, gg in gref.items():
            a, b = r[kk].double().cpu().numpy().ravel(), gg.numpy().ravel()
            assert np.linalg.norm(a - b) <= gtol * np.linalg.norm(b), (dt, kk, np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize('which', ['gp', 'svgp', 'sgp'])
@pytest.mark.parametrize('diagonal', [True, False])
def test_sampling_prediction_with_a_mean_function(golden_dir, which, diagonal):
    """The reference's test_sampling_prediction_w_mean (gpregression_test.py:309-350, svgpregression_test.py, sparsegpregression_test.py):
    the sampling-prediction algorithm of each module behind a trainable mean function, with injected noise -- the draws equal the oracle's
    (predictive moments of the zero-mean model on Y - m(X), plus m(X*))."""
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    from mxfusion_amd.modules.gp_modules.gp_regression import GPRegressionSamplingPrediction
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionSamplingPrediction
    from mxfusion_amd.modules.gp_modules.sparsegp_regression import SparseGPRegressionSamplingPrediction
    g = dict(np.load(os.path.join(golden_dir, {'gp': 'kat_gp.npz', 'svgp': 'kat_svgp.npz', 'sgp': 'kat_sgp.npz'}[which])))
    D = g['Y'].shape[1]
    rng = np.random.RandomState(11)
    W0, b0 = rng.randn(3, D) * 0.8, rng.randn(D) * 0.3
    m = _model(which, g, D, W0, b0)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.initialize(X=g['X'].shape, Y=g['Y'].shape)
    gp = m.Y.factor
    if which == 'svgp':
        post = gp._extra_graphs[0]
        infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = _t(g['qm']), _t(g['qW']), _t(g['qd'])
    infr.run(X=_t(g['X']), Y=_t(g['Y']))
    S, Nt = 3, g['Xt'].shape[0]
    eps = rng.randn(S, Nt, D)
    jit = 0. if diagonal else 1e-8
    cls, name = {'gp': (GPRegressionSamplingPrediction, 'gp_predict'), 'svgp': (SVGPRegressionSamplingPrediction, 'svgp_predict'),
                 'sgp': (SparseGPRegressionSamplingPrediction, 'sgp_predict')}[which]
    alg = cls(gp._module_graph, gp._extra_graphs[0], [gp._module_graph.X], rand_gen=MockRandomGenerator(_t(eps)), diagonal_variance=diagonal, jitter=jit)
    gp.attach_prediction_algorithms(targets=gp.output_names, conditionals=gp.input_names, algorithm=alg, alg_name=name)
    infr2 = TransferInference(ModulePredictionAlgorithm(model=m, observed=[m.X], target_variables=[m.Y], num_samples=S), infr_params=infr.params, dtype=DT)
    ys = infr2.run(X=_t(g['Xt']))[0]
    T = O.T
    k = O.RBF(3, ARD=True)
    kp = {'rbf_lengthscale': T(g['ls'])[None], 'rbf_variance': T(g['var'])[None]}
    Xt = T(g['Xt'])[None]
    mean_t, mean_x = _mean_fn(Xt, T(W0)[None], T(b0)[None]), _mean_fn(T(g['X'])[None], T(W0)[None], T(b0)[None])
    if which == 'gp':
        _, (Xc, L, LinvY) = O.gp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['noise'])[None], kp, jitter=1e-6, mean=mean_x, return_posterior=True)
        ref = O.gp_predict_sample(k, Xt, T(g['noise'])[None], Xc[None], L[None], LinvY[None], kp, T(eps), mean=mean_t, diagonal_variance=diagonal, jitter=jit)
    elif which == 'svgp':
        ref = O.svgp_predict_sample(k, Xt, T(g['Z'])[None], T(g['noise'])[None], T(g['qm'])[None], T(g['qW'])[None], T(g['qd'])[None], kp, T(eps),
                                    jitter=jit, mean=mean_t, diagonal_variance=diagonal)
    else:
        _, (wv, L, LA) = O.sgp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['Z'])[None], T(g['noise'])[None], kp, jitter=1e-8, mean=mean_x,
                                       return_posterior=True)
        ref = O.sgp_predict_sample(k, Xt, T(g['Z'])[None], T(g['noise'])[None], L[None], LA[None], wv[None], kp, T(eps), mean=mean_t,
                                   diagonal_variance=diagonal, jitter=jit)
    assert ys.shape == (S, Nt, D)
    assert np.allclose(ys.cpu().numpy(), ref.numpy(), atol=1e-7, rtol=1e-7), (which, diagonal)
    # the mean matters: the same draws without it differ
    assert float((ys.cpu() - (ref - mean_t)).abs().max()) > 1e-2
