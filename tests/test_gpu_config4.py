"""BASELINE.json configs[3]: the SVGP config with minibatches (B = 8192 of N = 65536, rv_scaling = N/B = 8) and the Monte-Carlo samples
sharded over the GPUs.  The model that makes minibatches and MC samples compatible in MXFusion's API is the uncertain-input SVGP: every
factor is a sum over rows (X ~ N(Xobs, s) row-wise, Y ~ SVGP(X), q(X) = N(Xobs, v) with one shared variance), so MinibatchInferenceLoop can
slice the observed rows and scale all three factors by N/B.

  * small: the first minibatch step through the API (model, posterior, StochasticVariationalInference, MinibatchInferenceLoop.step,
    injected noise) against the oracle's autograd -- loss and every gradient;
  * full size (B = 8192, M = 1024, Q = 8, 4 samples = one GPU's share of 32 over 8 GPUs, log_pdf_scaling = 8): exact linearity in
    log_pdf_scaling (float64) and float32-vs-float64 agreement of the ELBO at the initial AND at a trained-like length-scale.  (The same
    shape against the ORACLE's values and autograd gradients, S = 2: tests/test_gpu_fullsize_oracle.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _t(a, dt=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda()


def build_uncertain_input_svgp(N, Q, M, B, S, dtype, Z, loop=None, prior_var=1e-2, lengthscale=1.0, kernel_cls=None):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.models.posterior import Posterior
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, MinibatchInferenceLoop
    m = Model()
    m.N = Variable()
    m.Xobs = Variable(shape=(m.N, Q))
    m.X = Normal.define_variable(mean=m.Xobs, variance=prior_var, shape=(m.N, Q), dtype=dtype)
    m.Z = Variable(shape=(M, Q), initial_value=Z)
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    kernel = (kernel_cls or RBF)(input_dim=Q, ARD=True, variance=1., lengthscale=np.full(Q, float(lengthscale)), dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=dtype)
    m.Y.factor.svgp_log_pdf.jitter = 1e-6
    q = Posterior(m)
    q.qx_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=1e-2)
    q[m.X].set_prior(Normal(mean=q[m.Xobs], variance=q.qx_var, dtype=dtype))
    loop = loop or MinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B, m.X: N / B})
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Xobs, m.Y]), grad_loop=loop,
                              dtype=dtype)
    infr.initialize(Xobs=(B, Q), Y=(B, 1))
    return m, q, infr, loop, kernel


def test_uncertain_input_svgp_minibatch_step_matches_oracle():
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    rng = np.random.RandomState(0)
    N, Q, M, B, S = 24, 2, 4, 8, 4
    X, Y, Z = rng.rand(N, Q), rng.rand(N, 1), rng.rand(M, Q)
    eps = rng.randn(S, B, Q)
    m, q, infr, loop, kernel = build_uncertain_input_svgp(N, Q, M, B, S, 'float64', _t(Z))
    post = m.Y.factor._extra_graphs[0]
    qm, qW, qd = rng.randn(M, 1) * 0.1, rng.randn(M, M) * 0.05, rng.rand(M) + 0.5
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = _t(qm), _t(qW), _t(qd)
    q[m.X].factor._rand_gen = MockRandomGenerator(_t(eps.reshape(-1)))
    ex = infr.create_executor()
    sel = rng.permutation(N)[:B]
    loss = loop.step(ex, [_t(X[sel]), _t(Y[sel])], infr.params)
    assert m.Y.factor.svgp_log_pdf.log_pdf_scaling == N / B and m.X.factor.log_pdf_scaling == N / B and q[m.X].factor.log_pdf_scaling == N / B
    sp, isp = O.softplus, O.inv_softplus
    raw = {'qx_var': isp(O.T([1e-2])), 'noise_var': isp(O.T([0.01])), 'lengthscale': isp(O.T(np.ones(Q))), 'variance': isp(O.T([1.0])),
           'qU_mean': O.T(qm), 'qU_cov_W': O.T(qW), 'qU_cov_diag': isp(O.T(qd)), 'Z': O.T(Z)}
    lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    ref = O.svi_uncertain_input_svgp_loss(O.RBF(Q, ARD=True), O.T(X[sel]), O.T(Y[sel]), lv, O.T(eps), prior_var=1e-2, jitter=1e-6,
                                          log_pdf_scaling=N / B)
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) <= 1e-9 * abs(float(ref.detach()))
    P = infr.params
    g = P.flat.grad
    for var, name in ((q.qx_var, 'qx_var'), (m.noise_var, 'noise_var'), (kernel.lengthscale, 'lengthscale'), (kernel.variance, 'variance'),
                      (post.qU_mean, 'qU_mean'), (post.qU_cov_W, 'qU_cov_W'), (post.qU_cov_diag, 'qU_cov_diag'), (m.Z, 'Z')):
        o, n, _ = P._slices[var.uuid]
        a, b = g[o:o + n].cpu().numpy(), lv[name].grad.numpy().ravel()
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(b).max())), name


def _full_inputs(B, Q, M, S, seed=0):
    rng = np.random.default_rng(seed)
    Xobs = rng.uniform(-3., 3., (B, Q))
    w = rng.standard_normal(Q)
    Y = np.sin(Xobs @ w)[:, None] + 0.05 * rng.standard_normal((B, 1))
    Z = rng.uniform(-3., 3., (M, Q))
    X = Xobs[None] + 0.1 * rng.standard_normal((S, B, Q))
    qm = 0.3 * rng.standard_normal((M, 1))
    qW = 0.4 * rng.standard_normal((M, M)) / np.sqrt(M)
    qd = rng.uniform(0.05, 0.5, M)
    return X, Y, Z, qm, qW, qd


def _svgp(dt, X, Y, Z, qm, qW, qd, ls, scaling, want_grad=True):
    from mxfusion_amd import ops
    d = lambda a: _t(a, dt)
    S = X.shape[0]
    r = ops.svgp_logpdf('rbf', d(X), d(Y[None]), d(Z), d([0.02]), d(qm), d(qW), d(qd), d(ls), d([1.0]), True, jitter=1e-6, scaling=scaling,
                        gscale=1.0 / S, want_grad=want_grad)
    torch.cuda.synchronize()
    assert int(r['info'].abs().sum()) == 0
    return {k: v.double().cpu().numpy() for k, v in r.items()}


def test_config4_shapes_scaling_linearity_and_f32_agreement():
    B, Q, M, S = 8192, 8, 1024, 4
    X, Y, Z, qm, qW, qd = _full_inputs(B, Q, M, S)
    for l in (1.0, 2.2):                      # the notebook initial value, and where tests/probes/train_probe.py ends (trained-like)
        ls = np.full(Q, l)
        r8 = _svgp(torch.float64, X, Y, Z, qm, qW, qd, ls, 8.0)
        r4 = _svgp(torch.float64, X, Y, Z, qm, qW, qd, ls, 4.0, want_grad=False)
        r2 = _svgp(torch.float64, X, Y, Z, qm, qW, qd, ls, 2.0, want_grad=False)
        # log L = scaling * data term + negKL (svgp_regression.py:108): exactly linear in log_pdf_scaling
        d84, d42 = r8['logL'] - r4['logL'], r4['logL'] - r2['logL']
        assert np.allclose(d84, 2.0 * d42, rtol=1e-11), (l, d84, d42)
        f = _svgp(torch.float32, X, Y, Z, qm, qW, qd, ls, 8.0)
        rel = np.abs(f['logL'] - r8['logL']).max() / np.abs(r8['logL']).max()
        assert rel <= 1e-5, (l, rel)          # north_star: 1e-5 relative on the ELBO
        # gradients of the float32 streaming step against float64, normwise
        for k, tol in (('dX', 2e-4), ('dmu', 2e-4), ('dnoise', 2e-4), ('dZ', 5e-3), ('dls', 5e-3)):
            a, b = f[k].ravel(), r8[k].ravel()
            assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b), (l, k, np.linalg.norm(a - b) / np.linalg.norm(b))


def test_float32_validity_guard_reports_the_condition_of_kuu():
    """mxf_svgp_last_cond: the 1-norm condition number of Kuu + jitter I of the last training call, computed on the device next to the
    factorisation -- equal to the float64 value formed with torch, below the float32 limit (3e3) at the initial and at the trained-like
    length-scale, above it where the float32 streaming form is known to fail (tests/probes/f32_accuracy.py: ELBO error 2e-3 at 5e4)."""
    from mxfusion_amd import ops
    from mxfusion_amd.inference import GradBasedInference
    B, Q, M, S = 2048, 8, 1024, 2
    X, Y, Z, qm, qW, qd = _full_inputs(B, Q, M, S)
    conds = {}
    for l in (1.0, 2.2, 3.0):
        _svgp(torch.float32, X, Y, Z, qm, qW, qd, np.full(Q, l), 1.0)
        got = ops.svgp_last_cond()
        Zt = torch.as_tensor(Z, dtype=torch.float64).cuda() / l
        K = torch.exp(-0.5 * torch.cdist(Zt, Zt) ** 2) + 1e-6 * torch.eye(M, dtype=torch.float64, device='cuda')
        ref = float(K.abs().sum(0).max() * torch.linalg.inv(K).abs().sum(0).max())
        assert abs(got - ref) <= 1e-6 * ref, (l, got, ref)
        conds[l] = got
    lim = GradBasedInference.F32_COND_LIMIT
    assert conds[1.0] < lim and conds[3.0] > lim, conds


def test_row_shards_add_up_through_the_c_abi():
    """SURVEY 8(e), second axis, at the C ABI (what INTEGRATION.md tells a binder without the Python layer to do): rank r evaluates its rows with
    scaling = c * world and gscale = 1 / world and scales the bound by 1 / world -- the shards' bounds and gradients then SUM to the one-process call
    (float64, 1e-10); the KL term is counted once."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(3)
    B, M, Q, world, c = 1024, 128, 4, 4, 8.0
    X = rng.uniform(-2, 2, (1, B, Q)); Y = np.sin(X[0, :, :1]) + 0.1 * rng.randn(B, 1)
    Z = rng.uniform(-2, 2, (M, Q))
    qm, qW, qd = 0.3 * rng.randn(M, 1), 0.2 * rng.randn(M, M) / np.sqrt(M), rng.rand(M) * 0.4 + 0.1
    ls, var, noise = np.full(Q, 1.1), np.array([1.3]), np.array([0.05])
    args = lambda rows: (_t(X[:, rows]), _t(Y[None][:, rows]), _t(Z), _t(noise), _t(qm), _t(qW), _t(qd), _t(ls), _t(var), True)
    full = ops.svgp_logpdf('rbf', *args(slice(None)), jitter=1e-6, scaling=c, gscale=1.0, want_grad=True)
    tot = None
    for r in range(world):
        rows = slice(r * B // world, (r + 1) * B // world)
        part = ops.svgp_logpdf('rbf', *args(rows), jitter=1e-6, scaling=c * world, gscale=1.0 / world, want_grad=True)
        contrib = {k: part[k].double() for k in ('dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')}
        contrib['logL'] = part['logL'].double() / world
        tot = contrib if tot is None else {k: tot[k] + v for k, v in contrib.items()}
    torch.cuda.synchronize()
    for k, v in tot.items():
        ref = full[k].double()
        assert float((v - ref).abs().max()) <= 1e-10 * float(ref.abs().max()), k
