"""GPU parity of the sparse (Titsias) GP path (row a13): C-ABI composite and the SparseGPRegression module vs the oracle /
golden fixture built from testing/modules/sparsegpregression_test.py:38-47,58 (reference asserts the value vs GPy at :99)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def test_sgp_composite_golden(golden_dir):
    from mxfusion_amd import ops
    g = np.load(os.path.join(golden_dir, 'kat_sgp.npz'))
    r = ops.sgp_logpdf('rbf', _t(g['X']), _t(g['Y']), _t(g['Z']), _t(g['noise']), _t(g['ls']), _t(g['var']), True, jitter=1e-8, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    assert abs(float(r['logL'][0]) - (-20.731336414403)) < 1e-8
    for k in ('wv', 'L', 'LA'):
        assert np.allclose(r[k].cpu().numpy(), g[k], atol=1e-8), k
    for n, k in (('dX', 'd_X'), ('dY', 'd_Y'), ('dZ', 'd_Z'), ('dnoise', 'd_noise'), ('dls', 'd_ls'), ('dvar', 'd_var')):
        assert np.allclose(r[n].cpu().numpy().reshape(g[k].shape), g[k], rtol=1e-8, atol=1e-8 * max(1, np.abs(g[k]).max())), n


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 2e-5)])
@pytest.mark.parametrize('kind', ['rbf', 'matern32'])
def test_sgp_composite_vs_oracle(dtype, tol, kind):
    from mxfusion_amd import ops
    rng = np.random.RandomState(3)
    B, M, Q, P = 700, 90, 5, 2
    X = rng.uniform(-2, 2, (B, Q))
    Y = np.sin(X @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    ls = rng.rand(Q) * 0.5 + (1.0 if dtype == torch.float64 else 0.3)
    var, noise = np.array([1.3]), np.array([0.05])
    k = {'rbf': O.RBF, 'matern32': O.Matern32}[kind](Q, ARD=True)
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=X, Y=Y, Z=Z, noise=noise, ls=ls, var=var).items()}
    ref = O.sgp_log_pdf(k, lv['X'][None], lv['Y'][None], lv['Z'][None], lv['noise'][None],
                        {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6)[0]
    grads = torch.autograd.grad(ref, [lv[n] for n in ('X', 'Y', 'Z', 'noise', 'ls', 'var')])
    r = ops.sgp_logpdf(kind, _t(X, dtype), _t(Y, dtype), _t(Z, dtype), _t(noise, dtype), _t(ls, dtype), _t(var, dtype), True, jitter=1e-6,
                       want_grad=True)
    assert abs(float(r['logL'][0]) - float(ref)) <= tol * abs(float(ref))
    gtol = tol * 50 if dtype == torch.float64 else 5e-3
    for key, gr in zip(('dX', 'dY', 'dZ', 'dnoise', 'dls', 'dvar'), grads):
        a, b = r[key].cpu().numpy().reshape(gr.shape), gr.numpy()
        assert np.allclose(a, b, rtol=gtol, atol=gtol * max(1., np.abs(b).max())), key


def test_sgp_module_like_reference_tests(golden_dir):
    """testing/modules/sparsegpregression_test.py:80-99 (test_log_pdf) and :137-196 (test_prediction)."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SparseGPRegression
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    g = np.load(os.path.join(golden_dir, 'kat_sgp.npz'))
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.Z = Variable(shape=(3, 3), initial_value=_t(g['Z']))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(g['noise']))
    kernel = RBF(input_dim=3, ARD=True, variance=_t(g['var']), lengthscale=_t(g['ls']), dtype='float64')
    m.Y = SparseGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 2), dtype='float64')
    m.Y.factor.sgp_log_pdf.jitter = 1e-8
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype='float64')
    loss, _ = infr.run(X=_t(g['X']), Y=_t(g['Y']))
    assert abs(float(-loss) - (-20.731336414403)) < 1e-8
    gp = m.Y.factor
    for nf in (True, False):
        for dg in (True, False):
            infr2 = TransferInference(ModulePredictionAlgorithm(m, observed=[m.X], target_variables=[m.Y]), infr_params=infr.params, dtype='float64')
            gp.sgp_predict.noise_free = nf
            gp.sgp_predict.diagonal_variance = dg
            res = infr2.run(X=_t(g['Xt']))[0]
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            assert np.allclose(res[0].cpu().numpy(), g['mu_' + tag], atol=1e-8), tag
            assert np.allclose(res[1].cpu().numpy(), g['var_' + tag], atol=1e-8), tag


def test_sgp_with_a_combination_kernel_runs_the_materialised_path():
    """SparseGPRegression with Matern52 + RBF (no fused description): the reference's operator sequence (sparsegp_regression.py:67-106) on
    kern.K matrices through the differentiable potrf / trsm / gemm bridges -- loss and every gradient against the oracle, then prediction
    from the stored posterior."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import SparseGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop, TransferInference, ModulePredictionAlgorithm
    rng = np.random.RandomState(5)
    N, Q, M, D = 50, 3, 7, 2
    X, Y, Z, Xt = rng.uniform(-2, 2, (N, Q)), rng.randn(N, D), rng.uniform(-2, 2, (M, Q)), rng.uniform(-2, 2, (9, Q))
    ls1, ls2, v1, v2, noise = np.array([1.3]), np.array([0.7]), np.array([0.9]), np.array([0.4]), np.array([0.2])
    t64 = lambda a: _t(a, torch.float64)
    kern = Matern52(Q, variance=t64(v1), lengthscale=t64(ls1), dtype='float64') + RBF(Q, variance=t64(v2), lengthscale=t64(ls2), dtype='float64')
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=t64(Z))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t64(noise))
    m.Y = SparseGPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, D), dtype='float64')
    m.Y.factor.sgp_log_pdf.jitter = 1e-6
    grads, losses = [], []

    class Rec(BatchInferenceLoop):
        def step(self, ex, data, params):
            loss = super(Rec, self).step(ex, data, params)
            losses.append(float(loss.detach()))
            return loss

        def _exchange(self, param_dict, loss):
            grads.append(param_dict.flat.grad.clone())
            return loss
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=Rec(), dtype='float64')
    infr.run(X=t64(X), Y=t64(Y), max_iter=1, learning_rate=1e-9)
    ok = O.AddKernel([O.Matern52(Q), O.RBF(Q)])
    sp = O.softplus
    raw = {n: O.inv_softplus(O.T(v)).clone().requires_grad_(True) for n, v in dict(ls1=ls1, ls2=ls2, v1=v1, v2=v2, noise=noise).items()}
    Zr = O.T(Z).clone().requires_grad_(True)
    kp = lambda: {'add_matern52_lengthscale': sp(raw['ls1'])[None], 'add_matern52_variance': sp(raw['v1'])[None],
                  'add_rbf_lengthscale': sp(raw['ls2'])[None], 'add_rbf_variance': sp(raw['v2'])[None]}
    logL, post = O.sgp_log_pdf(ok, O.T(X)[None], O.T(Y)[None], Zr[None], sp(raw['noise'])[None], kp(), jitter=1e-6, return_posterior=True)
    (-logL.sum()).backward()
    assert abs(losses[0] - float(-logL.sum())) <= 1e-9 * abs(float(logL.sum()))
    P = infr.params
    sub = {k.name: k for k in kern.sub_kernels}
    for var, ref in ((m.noise_var, raw['noise'].grad), (sub['matern52'].lengthscale, raw['ls1'].grad), (sub['matern52'].variance, raw['v1'].grad),
                     (sub['rbf'].lengthscale, raw['ls2'].grad), (sub['rbf'].variance, raw['v2'].grad), (m.Z, Zr.grad)):
        o, n, _ = P._slices[var.uuid]
        assert np.allclose(grads[0][o:o + n].cpu().numpy(), ref.numpy().ravel(), rtol=1e-7, atol=1e-9), var.name
    infr2 = TransferInference(ModulePredictionAlgorithm(m, observed=[m.X], target_variables=[m.Y]), infr_params=infr.params, dtype='float64')
    mu, var = infr2.run(X=t64(Xt))[0]
    with torch.no_grad():
        rmu, rvar = O.sgp_predict(ok, O.T(Xt)[None], O.T(Z)[None], O.T(noise)[None], post[1][None], post[2][None], post[0][None],
                                  {k: v.detach() for k, v in kp().items()})
    assert np.allclose(mu.cpu().numpy(), rmu.numpy(), atol=1e-7) and np.allclose(var.cpu().numpy(), rvar.numpy(), atol=1e-7)


def test_sparse_gp_float32_guard_widens_above_the_limit():
    """The Titsias bound in float32 has the explicit SVGP form's conditioning limit (ELBO 7e-6 at cond_1(Kuu) 3e4, NaN at 1e6: a float32 Psi2
    inside C = Kuu + Psi2 / s2).  The sparse-GP call publishes its condition number like the SVGP call (r04) and the module's guard
    widens it to float64 above Float32Guard.LIMIT: through the float32 bridge the bound holds 1e-5 at cond ~ 1e6, first call included."""
    import warnings
    from mxfusion_amd.modules.gp_modules._fused import SGPLogPdfFn, Float32Guard as G
    rng = np.random.default_rng(0)
    B, Q, M = 4096, 8, 512
    X = rng.uniform(-3., 3., (B, Q))
    Y = np.sin(X @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((B, 1))
    Z = rng.uniform(-3., 3., (M, Q))
    for ell, explicit in ((1.0, True), (3.0, False)):      # (above the limit the owner's level is 'whitened' or 'float64': a call the whitened form does not cover runs float64 either way)
        ls, var, noise = np.full(Q, ell), np.array([1.0]), np.array([0.02])
        T = O.T
        ref = float(O.sgp_log_pdf(O.RBF(Q, ARD=True), T(X)[None], T(Y)[None], T(Z)[None], T(noise)[None],
                                  {'rbf_lengthscale': T(ls)[None], 'rbf_variance': T(var)[None]}, jitter=1e-6)[0])
        t = {k: torch.as_tensor(v, dtype=torch.float32).cuda().requires_grad_(k != 'Y') for k, v in
             dict(X=X[None], Y=Y[None], Z=Z[None], noise=noise[None], ls=ls[None], var=var[None]).items()}
        g = G('sgp-test')
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            out = SGPLogPdfFn.apply(g, 'rbf', True, 1e-6, t['X'], t['Y'], t['Z'], t['noise'], t['ls'], t['var'])
            out[0].sum().backward()
        torch.cuda.synchronize()
        assert (g.tier == G.EXPLICIT) == explicit and g.cond_max > 0, (ell, g.tier, g.cond_max)
        assert out[0].dtype == torch.float32 and t['Z'].grad.dtype == torch.float32 and bool(torch.isfinite(t['Z'].grad).all())
        assert abs(float(out[0][0]) - ref) <= 1e-5 * abs(ref), (ell, float(out[0][0]), ref)
