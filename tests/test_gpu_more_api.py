"""GPU parity of the remaining algorithm classes: prior / predictive sampling with injected noise (a9, a12), combination
kernels through the generic Cholesky path (a6 + a7), the minibatch loop with rv_scaling (a20)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

DT = 'float64'


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()


def _gp_model(g, kernel=None, rand_gen=None):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(g['noise']))
    kernel = kernel or RBF(input_dim=3, ARD=True, variance=_t(g['var']), lengthscale=_t(g['ls']), dtype=DT)
    m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, 2), dtype=DT, rand_gen=rand_gen)
    return m


def test_gp_prior_and_predictive_sampling_with_injected_noise(golden_dir):
    """testing/modules/gpregression_test.py:131-168 (test_draw_samples) and :256-307 (sampling prediction)."""
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import Inference, MAP, ForwardSamplingAlgorithm, TransferInference, ModulePredictionAlgorithm
    from mxfusion_amd.modules.gp_modules.gp_regression import GPRegressionSamplingPrediction
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    k = O.RBF(3, ARD=True)
    kp = {'rbf_lengthscale': O.T(g['ls'])[None], 'rbf_variance': O.T(g['var'])[None]}
    rng = np.random.RandomState(5)
    eps = rng.randn(2, 10, 2)
    m = _gp_model(g, rand_gen=MockRandomGenerator(_t(eps)))
    infr = Inference(ForwardSamplingAlgorithm(m, [m.X], num_samples=2, target_variables=[m.Y]), dtype=DT)
    samples = infr.run(X=_t(g['X']))[0]
    ref = O.gp_sample_prior(k, O.T(g['X'])[None], O.T(g['noise'])[None], kp, O.T(eps))
    assert np.allclose(samples.cpu().numpy(), ref.numpy(), atol=1e-10)
    # predictive sampling: swap the prediction algorithm through the registry (gp_regression.ipynb cell 24)
    m = _gp_model(g)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.run(X=_t(g['X']), Y=_t(g['Y']))
    post = O.gp_log_pdf(k, O.T(g['X'])[None], O.T(g['Y'])[None], O.T(g['noise'])[None], kp, return_posterior=True)[1]
    gp = m.Y.factor
    for dg, jit in ((True, 0.), (False, 1e-8)):
        eps2 = rng.randn(3, 20, 2)
        alg = GPRegressionSamplingPrediction(gp._module_graph, gp._extra_graphs[0], [gp._module_graph.X], rand_gen=MockRandomGenerator(_t(eps2)),
                                             diagonal_variance=dg, jitter=jit)
        gp.attach_prediction_algorithms(targets=gp.output_names, conditionals=gp.input_names, algorithm=alg, alg_name='gp_predict')
        infr2 = TransferInference(ModulePredictionAlgorithm(model=m, observed=[m.X], target_variables=[m.Y], num_samples=3),
                                  infr_params=infr.params, dtype=DT)
        ys = infr2.run(X=_t(g['Xt']))[0]
        ref = O.gp_predict_sample(k, O.T(g['Xt'])[None], O.T(g['noise'])[None], post[0][None], post[1][None], post[2][None], kp, O.T(eps2),
                                  diagonal_variance=dg, jitter=jit)
        assert np.allclose(ys.cpu().numpy(), ref.numpy(), atol=1e-8), dg


def test_combination_kernel_gp_logpdf_and_gradients():
    """AddKernel(Matern52, RBF) (add_kernel.py:44-68, the deep-GP config's kernel) through the generic path: HIP Gram per sub-kernel
    (autograd via mxf_gram_bwd) + mxf_potrf/trsm/trtri with the closed-form dK."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import GPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
    rng = np.random.RandomState(0)
    N, Q = 60, 4
    X, Y = rng.uniform(-2, 2, (N, Q)), rng.randn(N, 1)
    ls1, ls2, v1, v2, noise = np.array([1.3]), np.array([0.7]), np.array([0.9]), np.array([0.4]), np.array([0.2])
    kern = Matern52(Q, variance=_t(v1), lengthscale=_t(ls1), dtype=DT) + RBF(Q, variance=_t(v2), lengthscale=_t(ls2), dtype=DT)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=_t(noise))
    m.Y = GPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, shape=(m.N, 1), dtype=DT)
    grads = []

    class Rec(BatchInferenceLoop):
        def _exchange(self, param_dict, loss):
            grads.append(param_dict.flat.grad.clone())
            return loss
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=Rec(), dtype=DT)
    infr.run(X=_t(X), Y=_t(Y), max_iter=1, learning_rate=0.01)
    ok = O.AddKernel([O.Matern52(Q), O.RBF(Q)])
    raw = {n: O.inv_softplus(O.T(v)).clone().requires_grad_(True) for n, v in dict(ls1=ls1, ls2=ls2, v1=v1, v2=v2, noise=noise).items()}
    sp = O.softplus
    logL = O.gp_log_pdf(ok, O.T(X)[None], O.T(Y)[None], sp(raw['noise'])[None],
                        {'add_matern52_lengthscale': sp(raw['ls1'])[None], 'add_matern52_variance': sp(raw['v1'])[None],
                         'add_rbf_lengthscale': sp(raw['ls2'])[None], 'add_rbf_variance': sp(raw['v2'])[None]})
    (-logL.sum()).backward()
    P = infr.params
    sub = {k.name: k for k in kern.sub_kernels}
    for var, name in ((m.noise_var, 'noise'), (sub['matern52'].lengthscale, 'ls1'), (sub['matern52'].variance, 'v1'),
                      (sub['rbf'].lengthscale, 'ls2'), (sub['rbf'].variance, 'v2')):
        o, n, _ = P._slices[var.uuid]
        assert np.allclose(grads[0][o:o + n].cpu().numpy(), raw[name].grad.numpy(), rtol=1e-8, atol=1e-9), name


def test_minibatch_loop_with_rv_scaling_runs_and_scales(golden_dir):
    """inference/minibatch_loop.py + rv_scaling -> SVGPRegressionLogPdf.log_pdf_scaling (svgp_regression.py:108); first minibatch loss vs oracle."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, MinibatchInferenceLoop
    rng = np.random.RandomState(1)
    N, Q, M, Bsz = 64, 2, 8, 16
    X, Y, Z = rng.uniform(-2, 2, (N, Q)), rng.randn(N, 1), rng.uniform(-2, 2, (M, Q))
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=_t(Z))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.1)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=RBF(Q, ARD=True, dtype=DT), noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=DT)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    seen = []

    class Loop(MinibatchInferenceLoop):
        def run(self, infr_executor, data, **kw):
            def wrapped(*a):
                out = infr_executor(*a)
                seen.append((a[0].detach().clone(), a[1].detach().clone(), float(out[0].detach())))
                return out
            return super(Loop, self).run(wrapped, data, **kw)
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=Loop(batch_size=Bsz, rv_scaling={m.Y: N / Bsz}), dtype=DT)
    infr.initialize(X=(N, Q), Y=(N, 1))
    post = gp._extra_graphs[0]
    qm, qW, qd = rng.randn(M, 1) * 0.1, rng.randn(M, M) * 0.05, rng.rand(M) + 0.5
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = _t(qm), _t(qW), _t(qd)
    infr.run(X=_t(X), Y=_t(Y), max_iter=1, learning_rate=1e-3)
    assert len(seen) == N // Bsz and gp.svgp_log_pdf.log_pdf_scaling == N / Bsz
    xb, yb, loss0 = seen[0]
    k = O.RBF(Q, ARD=True)
    ref = O.svgp_log_pdf(k, O.T(xb.cpu().numpy())[None], O.T(yb.cpu().numpy())[None], O.T(Z)[None], O.T([[0.1]]), O.T(qm)[None], O.T(qW)[None],
                         O.T(qd)[None], {'rbf_lengthscale': O.T(np.ones((1, Q))), 'rbf_variance': O.T([[1.0]])}, jitter=1e-6,
                         log_pdf_scaling=N / Bsz)
    assert abs(loss0 - float(-ref[0])) < 1e-8 * abs(float(ref[0]))


def test_svgp_heteroscedastic_noise_through_the_api(golden_dir):
    """testing/modules/svgpregression_test.py:142-167 (test_log_pdf_w_samples_of_noise_var): noise_var Variable of shape (N, D), D = 2,
    one MAP iteration.  The reference only smoke-tests it; here the loss of that iteration is also checked against the oracle."""
    from oracle import gp_oracle as O
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import Inference, MAP
    g = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    rng = np.random.RandomState(7)
    D = 2
    X, Z = g['X'], g['Z']
    Y = rng.rand(10, D)
    qm = rng.rand(3, D)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.Z = Variable(shape=(3, 3), initial_value=_t(Z))
    m.noise_var = Variable(transformation=PositiveTransformation(), shape=(m.N, D))
    kernel = RBF(input_dim=3, ARD=True, variance=_t(g['var']), lengthscale=_t(g['ls']), dtype=DT)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, D), dtype=DT)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-8
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.initialize(X=X.shape, Y=Y.shape)
    infr.params[gp._extra_graphs[0].qU_mean] = _t(qm)
    infr.params[gp._extra_graphs[0].qU_cov_W] = _t(g['qW'])
    infr.params[gp._extra_graphs[0].qU_cov_diag] = _t(g['qd'])
    noise0 = infr.params[m.noise_var].detach().cpu().double()          # the randomly initialised (N, D) noise, constrained space
    assert noise0.shape[-2:] == (10, D)
    loss, _ = infr.run(X=_t(X), Y=_t(Y), max_iter=1)
    k = O.RBF(3, ARD=True)
    ref = O.svgp_log_pdf(k, O.T(X)[None], O.T(Y)[None], O.T(Z)[None], noise0.reshape(1, 10, D), O.T(qm)[None], O.T(g['qW'])[None],
                         O.T(g['qd'])[None], {k.name + '_lengthscale': O.T(g['ls'])[None], k.name + '_variance': O.T(g['var'])[None]},
                         jitter=1e-8)
    assert abs(float(-loss) - float(ref[0])) < 1e-8 * max(1.0, abs(float(ref[0])))


@pytest.mark.parametrize('latent', [False, True])
def test_svgp_with_add_kernel_through_the_api(latent):
    """The deep-GP configuration's layer: SVGPRegression with AddKernel(Matern52, RBF) (add_kernel.py:44-68), plain and
    with latent (sampled) inputs as in svgpregression_test.py:357-385; first-step loss and flat gradient vs the oracle."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, StochasticVariationalInference, BatchInferenceLoop, create_Gaussian_meanfield
    rng = np.random.RandomState(3)
    N, Q, M, S = 50, 3, 7, 3
    X, Y, Z = rng.uniform(-2, 2, (N, Q)), rng.randn(N, 1), rng.uniform(-2, 2, (M, Q))
    ls1, ls2, v1, v2, noise = np.array([1.3]), np.array([0.7]), np.array([0.9]), np.array([0.4]), np.array([0.2])
    qm, qW, qd = rng.randn(M, 1) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    eps = rng.randn(S, N, Q)
    kern = Matern52(Q, variance=_t(v1), lengthscale=_t(ls1), dtype=DT) + RBF(Q, variance=_t(v2), lengthscale=_t(ls2), dtype=DT)
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, Q)) if latent else Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=_t(Z))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=_t(noise))
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=DT)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    grads, losses = [], []

    class Rec(BatchInferenceLoop):
        def _exchange(self, param_dict, loss):
            grads.append(param_dict.flat.grad.clone())
            return loss

        def run(self, infr_executor, data, **kw):
            def wrapped(*a):
                out = infr_executor(*a)
                losses.append(float(out[0]))
                return out
            return super(Rec, self).run(wrapped, data, **kw)
    if latent:
        q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=DT)
        qX = q[m.X].factor
        qX._rand_gen = MockRandomGenerator(_t(eps.reshape(-1)))
        alg = StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y])
        data = dict(Y=_t(Y))
    else:
        alg = MAP(model=m, observed=[m.X, m.Y])
        data = dict(X=_t(X), Y=_t(Y))
    infr = GradBasedInference(alg, grad_loop=Rec(), dtype=DT)
    infr.initialize(**{k: v.shape for k, v in data.items()})
    infr.params[gp._extra_graphs[0].qU_mean] = _t(qm)
    infr.params[gp._extra_graphs[0].qU_cov_W] = _t(qW)
    infr.params[gp._extra_graphs[0].qU_cov_diag] = _t(qd)
    if latent:
        infr.params[qX.mean] = _t(X)
        infr.params[qX.variance] = _t(np.full((N, Q), 0.01))
    infr.run(max_iter=1, learning_rate=1e-3, **data)
    loss = losses[0]

    ok = O.AddKernel([O.Matern52(Q), O.RBF(Q)])
    sp = O.softplus
    raw = {n: O.inv_softplus(O.T(v)).clone().requires_grad_(True) for n, v in dict(ls1=ls1, ls2=ls2, v1=v1, v2=v2, noise=noise, qd=qd).items()}
    lin = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(Z=Z, qm=qm, qW=qW).items()}
    kp = {'add_matern52_lengthscale': sp(raw['ls1'])[None], 'add_matern52_variance': sp(raw['v1'])[None],
          'add_rbf_lengthscale': sp(raw['ls2'])[None], 'add_rbf_variance': sp(raw['v2'])[None]}
    if latent:
        xm = O.T(X).clone().requires_grad_(True)
        xv_raw = O.inv_softplus(O.T(np.full((N, Q), 0.01))).clone().requires_grad_(True)
        xv = sp(xv_raw)
        Xs = xm[None] + O.T(eps) * torch.sqrt(xv)[None]
        logq = O.normal_log_pdf(xm[None], xv[None], Xs).reshape(S, -1).sum(-1)
        logp = O.normal_log_pdf(torch.zeros(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64), Xs).reshape(S, -1).sum(-1)
    else:
        Xs, logq, logp = O.T(X)[None], 0.0, 0.0
    logL = O.svgp_log_pdf(ok, Xs, O.T(Y)[None], lin['Z'][None], sp(raw['noise'])[None], lin['qm'][None], lin['qW'][None], sp(raw['qd'])[None],
                          kp, jitter=1e-6)
    obj = -((logL + logp - logq).mean() if latent else logL.sum())
    obj.backward()
    assert abs(float(loss) - float(obj)) < 1e-8 * max(1.0, abs(float(obj)))
    P = infr.params
    sub = {k.name: k for k in kern.sub_kernels}
    checks = [(m.noise_var, raw['noise']), (sub['matern52'].lengthscale, raw['ls1']), (sub['matern52'].variance, raw['v1']),
              (sub['rbf'].lengthscale, raw['ls2']), (sub['rbf'].variance, raw['v2']), (m.Z, lin['Z']),
              (gp._extra_graphs[0].qU_mean, lin['qm']), (gp._extra_graphs[0].qU_cov_W, lin['qW']), (gp._extra_graphs[0].qU_cov_diag, raw['qd'])]
    if latent:
        checks += [(qX.mean, xm), (qX.variance, xv_raw)]
    for var, ref in checks:
        o, n, _ = P._slices[var.uuid]
        assert np.allclose(grads[0][o:o + n].cpu().numpy(), ref.grad.numpy().reshape(-1), rtol=1e-7, atol=1e-8), var


def test_two_layer_deep_gp_svi_step_matches_oracle():
    """SURVEY 8f rank 1 (BASELINE config 5 in miniature): two chained SVGPRegression modules, first layer AddKernel(Matern52, RBF),
    hidden layer H with a mean-field q(H) (inference/meanfield.py:24-44), StochasticVariationalInference with injected noise.
    Loss and flat gradient of the first step vs the oracle."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, BatchInferenceLoop, create_Gaussian_meanfield
    rng = np.random.RandomState(11)
    N, Q, Dh, M, S = 40, 3, 2, 6, 4
    X, Y = rng.uniform(-2, 2, (N, Q)), rng.randn(N, 1)
    Z0, Z1 = rng.uniform(-2, 2, (M, Q)), rng.randn(M, Dh)
    p = dict(ls1=np.array([1.3]), v1=np.array([0.9]), ls2=np.array([0.7]), v2=np.array([0.4]), ls3=np.array([1.1, 0.8]), v3=np.array([1.2]),
             n0=np.array([0.1]), n1=np.array([0.2]))
    qm0, qW0, qd0 = rng.randn(M, Dh) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    qm1, qW1, qd1 = rng.randn(M, 1) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    hm, hv = rng.randn(N, Dh), np.full((N, Dh), 0.05)
    eps = rng.randn(S, N, Dh)
    k0 = Matern52(Q, variance=_t(p['v1']), lengthscale=_t(p['ls1']), dtype=DT) + RBF(Q, variance=_t(p['v2']), lengthscale=_t(p['ls2']), dtype=DT)
    k1 = RBF(Dh, ARD=True, variance=_t(p['v3']), lengthscale=_t(p['ls3']), name='rbf_top', dtype=DT)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z0 = Variable(shape=(M, Q), initial_value=_t(Z0))
    m.Z1 = Variable(shape=(M, Dh), initial_value=_t(Z1))
    m.noise0 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=_t(p['n0']))
    m.noise1 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=_t(p['n1']))
    m.H = SVGPRegression.define_variable(X=m.X, kernel=k0, noise_var=m.noise0, inducing_inputs=m.Z0, shape=(m.N, Dh), dtype=DT)
    m.Y = SVGPRegression.define_variable(X=m.H, kernel=k1, noise_var=m.noise1, inducing_inputs=m.Z1, shape=(m.N, 1), dtype=DT)
    g0, g1 = m.H.factor, m.Y.factor
    g0.svgp_log_pdf.jitter = g1.svgp_log_pdf.jitter = 1e-6
    q = create_Gaussian_meanfield(model=m, observed=[m.X, m.Y], dtype=DT)
    qH = q[m.H].factor
    qH._rand_gen = MockRandomGenerator(_t(eps.reshape(-1)))
    grads, losses = [], []

    class Rec(BatchInferenceLoop):
        def _exchange(self, param_dict, loss):
            grads.append(param_dict.flat.grad.clone())
            return loss

        def run(self, infr_executor, data, **kw):
            def wrapped(*a):
                out = infr_executor(*a)
                losses.append(float(out[0].detach()))
                return out
            return super(Rec, self).run(wrapped, data, **kw)
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.X, m.Y]), grad_loop=Rec(), dtype=DT)
    infr.initialize(X=X.shape, Y=Y.shape)
    for gp, (a, b, c) in ((g0, (qm0, qW0, qd0)), (g1, (qm1, qW1, qd1))):
        infr.params[gp._extra_graphs[0].qU_mean] = _t(a)
        infr.params[gp._extra_graphs[0].qU_cov_W] = _t(b)
        infr.params[gp._extra_graphs[0].qU_cov_diag] = _t(c)
    infr.params[qH.mean] = _t(hm)
    infr.params[qH.variance] = _t(hv)
    infr.run(X=_t(X), Y=_t(Y), max_iter=1, learning_rate=1e-3)

    sp = O.softplus
    raw = {n: O.inv_softplus(O.T(v)).clone().requires_grad_(True) for n, v in dict(p, qd0=qd0, qd1=qd1, hv=hv).items()}
    lin = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(Z0=Z0, Z1=Z1, qm0=qm0, qW0=qW0, qm1=qm1, qW1=qW1, hm=hm).items()}
    ok0, ok1 = O.AddKernel([O.Matern52(Q), O.RBF(Q)]), O.RBF(Dh, ARD=True, name='rbf_top')
    Hs = lin['hm'][None] + O.T(eps) * torch.sqrt(sp(raw['hv']))[None]
    l0 = O.svgp_log_pdf(ok0, O.T(X)[None], Hs, lin['Z0'][None], sp(raw['n0'])[None], lin['qm0'][None], lin['qW0'][None], sp(raw['qd0'])[None],
                        {'add_matern52_lengthscale': sp(raw['ls1'])[None], 'add_matern52_variance': sp(raw['v1'])[None],
                         'add_rbf_lengthscale': sp(raw['ls2'])[None], 'add_rbf_variance': sp(raw['v2'])[None]}, jitter=1e-6)
    l1 = O.svgp_log_pdf(ok1, Hs, O.T(Y)[None], lin['Z1'][None], sp(raw['n1'])[None], lin['qm1'][None], lin['qW1'][None], sp(raw['qd1'])[None],
                        {'rbf_top_lengthscale': sp(raw['ls3'])[None], 'rbf_top_variance': sp(raw['v3'])[None]}, jitter=1e-6)
    logq = O.normal_log_pdf(lin['hm'][None], sp(raw['hv'])[None], Hs).reshape(S, -1).sum(-1)
    obj = -(l0 + l1 - logq).mean()
    obj.backward()
    assert abs(losses[0] - float(obj)) < 1e-8 * max(1.0, abs(float(obj)))
    P = infr.params
    sub = {k.name: k for k in k0.sub_kernels}
    checks = [(m.noise0, raw['n0']), (m.noise1, raw['n1']), (sub['matern52'].lengthscale, raw['ls1']), (sub['matern52'].variance, raw['v1']),
              (sub['rbf'].lengthscale, raw['ls2']), (sub['rbf'].variance, raw['v2']), (k1.lengthscale, raw['ls3']), (k1.variance, raw['v3']),
              (m.Z0, lin['Z0']), (m.Z1, lin['Z1']), (qH.mean, lin['hm']), (qH.variance, raw['hv']),
              (g0._extra_graphs[0].qU_mean, lin['qm0']), (g0._extra_graphs[0].qU_cov_W, lin['qW0']), (g0._extra_graphs[0].qU_cov_diag, raw['qd0']),
              (g1._extra_graphs[0].qU_mean, lin['qm1']), (g1._extra_graphs[0].qU_cov_W, lin['qW1']), (g1._extra_graphs[0].qU_cov_diag, raw['qd1'])]
    for var, ref in checks:
        o, n, _ = P._slices[var.uuid]
        assert np.allclose(grads[0][o:o + n].cpu().numpy(), ref.grad.numpy().reshape(-1), rtol=1e-7, atol=1e-8), var


@pytest.mark.parametrize('square', [True, False])
@pytest.mark.parametrize('S', [1, 3])
@pytest.mark.parametrize('ard', [True, False])
def test_linear_bias_white_kernels_reverse_mode(square, S, ard):
    """Linear / Bias / White (kernels/linear.py:59-103, static.py:56-164) and their combination with a stationary
    kernel (sum and product): K and d<K, G>/d(inputs, parameters) vs the oracle's autograd."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Linear, Bias, White
    rng = np.random.RandomState(17 + S)
    N, N2, Q = 9, 7, 3
    X, X2 = rng.randn(S, N, Q), rng.randn(1, N2, Q)
    lv, bv, wv = rng.rand(1, Q if ard else 1) + 0.3, rng.rand(1, 1) + 0.2, rng.rand(1, 1) + 0.1
    rl, rv = rng.rand(1, 1) + 0.5, rng.rand(1, 1) + 0.5
    G = rng.randn(S, N, N if square else N2)
    kern = (Linear(Q, ARD=ard, dtype=DT) + Bias(Q, dtype=DT) + White(Q, dtype=DT)) * RBF(Q, dtype=DT)
    okern = O.MultiplyKernel([O.AddKernel([O.AddKernel([O.Linear(Q, ARD=ard), O.Bias(Q)]), O.White(Q)]), O.RBF(Q)])   # a sum of sums is ONE AddKernel, add_kernel.py:36-46
    vals = dict(X=X, X2=X2, lv=lv, bv=bv, wv=wv, rl=rl, rv=rv)
    dev = {n: _t(v).requires_grad_(True) for n, v in vals.items()}
    ora = {n: O.T(v).clone().requires_grad_(True) for n, v in vals.items()}
    names = {'mul_add_linear_variances': 'lv', 'mul_add_bias_variance': 'bv', 'mul_add_white_variance': 'wv', 'mul_rbf_lengthscale': 'rl',
             'mul_rbf_variance': 'rv'}
    assert sorted(kern.parameters) == sorted(names)
    K = kern.K(None, dev['X'], None if square else dev['X2'], **{k: dev[v] for k, v in names.items()})
    Ko = okern.K(ora['X'], None if square else ora['X2'], **{k: ora[v] for k, v in names.items()})
    assert np.allclose(K.detach().cpu().numpy(), Ko.detach().numpy(), rtol=1e-11, atol=1e-12)
    (K * _t(G)).sum().backward()
    (Ko * O.T(G)).sum().backward()
    for n in vals:
        if n == 'X2' and square:
            continue
        if ora[n].grad is None:          # White with an explicit X2 is identically zero (static.py:147-150)
            assert dev[n].grad is None or float(dev[n].grad.abs().max()) == 0.0, n
            continue
        assert np.allclose(dev[n].grad.cpu().numpy(), ora[n].grad.numpy(), rtol=1e-9, atol=1e-10), n
    Kd = kern.Kdiag(None, dev['X'], **{k: dev[v].detach() for k, v in names.items()})
    Kdo = okern.Kdiag(ora['X'].detach(), **{k: ora[v].detach() for k, v in names.items()})
    assert np.allclose(Kd.detach().cpu().numpy(), Kdo.numpy(), rtol=1e-11)


def test_graph_captured_step_matches_eager(golden_dir):
    """BatchInferenceLoop(use_graph=True): the forward + reverse pass of the SVI step replayed from a hipGraph (three streams, ~280
    launches) gives the same parameter trajectory as eager execution (deterministic injected noise)."""
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, create_Gaussian_meanfield, BatchInferenceLoop
    g = np.load(os.path.join(golden_dir, 'kat_svi.npz'))
    k = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))

    def run(use_graph):
        from mxfusion_amd import Model, Variable
        from mxfusion_amd.components.variables import PositiveTransformation
        from mxfusion_amd.components.distributions import Normal
        from mxfusion_amd.components.distributions.gp.kernels import RBF
        from mxfusion_amd.modules.gp_modules import SVGPRegression
        m = Model()
        m.N = Variable()
        m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, 3))
        m.Z = Variable(shape=(3, 3), initial_value=_t(k['Z']))
        m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(k['noise']))
        kernel = RBF(input_dim=3, ARD=True, variance=_t(k['var']), lengthscale=_t(k['ls']), dtype=DT)
        m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=DT)
        m.Y.factor.svgp_log_pdf.jitter = 1e-8
        q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=DT)
        # ONE step's worth of noise, replayed every step: host-side state (the mock generator's cursor) is frozen in a captured graph
        q[m.X].factor._rand_gen = MockRandomGenerator(_t(g['eps'][0]))
        S = g['eps'].shape[1]
        infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y]),
                                  grad_loop=BatchInferenceLoop(use_graph=use_graph), dtype=DT)
        infr.initialize(Y=g['Y'].shape)
        infr.params[m.Y.factor._extra_graphs[0].qU_mean] = _t(g['init_qU_mean'])
        infr.params[m.Y.factor._extra_graphs[0].qU_cov_W] = _t(g['init_qU_cov_W'])
        infr.run(Y=_t(g['Y']), max_iter=6, learning_rate=0.05)
        return infr.params.flat.detach().clone()
    torch.manual_seed(0); np.random.seed(0)          # un-set parameters (q(X)) are initialised from the host RNG
    a = run(False)
    torch.manual_seed(0); np.random.seed(0)
    b = run(True)
    assert torch.allclose(a, b, rtol=1e-9, atol=1e-10), float((a - b).abs().max())


def test_active_dims_and_their_gradients():
    """kernel_test.py:302-369: kernels restricted to a subset of the input columns (kernel.py:119-123), alone and inside Add / Multiply;
    the gradient w.r.t. the full input must vanish on the inactive columns."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern32, Linear
    rng = np.random.RandomState(21)
    N, N2, D = 11, 6, 5
    X, X2 = rng.randn(2, N, D), rng.randn(1, N2, D)
    G = rng.randn(2, N, N2)
    kern = RBF(2, ARD=True, active_dims=[0, 3], dtype=DT) + Matern32(1, active_dims=[1], dtype=DT) * Linear(2, ARD=True, active_dims=[2, 4], dtype=DT)
    okern = O.AddKernel([O.RBF(2, ARD=True, active_dims=[0, 3]),
                         O.MultiplyKernel([O.Matern32(1, active_dims=[1]), O.Linear(2, ARD=True, active_dims=[2, 4])])])
    vals = {'add_rbf_lengthscale': rng.rand(1, 2) + 0.6, 'add_rbf_variance': rng.rand(1, 1) + 0.5,
            'add_mul_matern32_lengthscale': rng.rand(1, 1) + 0.8, 'add_mul_matern32_variance': rng.rand(1, 1) + 0.5,
            'add_mul_linear_variances': rng.rand(1, 2) + 0.3}
    assert sorted(kern.parameters) == sorted(vals)
    dX, dX2 = _t(X).requires_grad_(True), _t(X2).requires_grad_(True)
    dp = {k: _t(v).requires_grad_(True) for k, v in vals.items()}
    oX, oX2 = O.T(X).clone().requires_grad_(True), O.T(X2).clone().requires_grad_(True)
    op = {k: O.T(v).clone().requires_grad_(True) for k, v in vals.items()}
    K, Ko = kern.K(None, dX, dX2, **dp), okern.K(oX, oX2, **op)
    assert np.allclose(K.detach().cpu().numpy(), Ko.detach().numpy(), rtol=1e-11, atol=1e-12)
    (K * _t(G)).sum().backward()
    (Ko * O.T(G)).sum().backward()
    assert np.allclose(dX.grad.cpu().numpy(), oX.grad.numpy(), rtol=1e-9, atol=1e-10)
    assert np.allclose(dX2.grad.cpu().numpy(), oX2.grad.numpy(), rtol=1e-9, atol=1e-10)
    for k in vals:
        assert np.allclose(dp[k].grad.cpu().numpy(), op[k].grad.numpy(), rtol=1e-9, atol=1e-10), k
    Kd = kern.Kdiag(None, dX.detach(), **{k: v.detach() for k, v in dp.items()})
    assert np.allclose(Kd.cpu().numpy(), okern.Kdiag(oX.detach(), **{k: v.detach() for k, v in op.items()}).numpy(), rtol=1e-11)


@pytest.mark.parametrize('module', ['gp', 'sgp'])
def test_with_samples_gp_and_sparse_gp(module, golden_dir):
    """gpregression_test.py:309-350 / sparsegpregression_test.py:287-330 (test_with_samples; smoke tests in the reference): latent
    X ~ N(0, 1) with a mean-field q(X), S injected-noise samples, one SVI step -- here loss and flat gradient are checked against the oracle."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression, SparseGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, BatchInferenceLoop, create_Gaussian_meanfield
    rng = np.random.RandomState(5)
    N, Q, D, M, S = 10, 3, 2, 4, 5
    Y, Z = rng.rand(N, D), rng.rand(M, Q)
    ls, var, noise = rng.rand(Q) + 0.5, rng.rand(1) + 0.5, rng.rand(1) * 0.3 + 0.1
    xm, xv, eps = rng.randn(N, Q), rng.rand(N, Q) * 0.1 + 0.01, rng.randn(S, N, Q)
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, Q))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(noise))
    kernel = RBF(input_dim=Q, ARD=True, variance=_t(var), lengthscale=_t(ls), dtype=DT)
    if module == 'gp':
        m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, D), dtype=DT)
    else:
        m.Z = Variable(shape=(M, Q), initial_value=_t(Z))
        m.Y = SparseGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, D), dtype=DT)
        m.Y.factor.sgp_log_pdf.jitter = 1e-8
    q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=DT)
    qX = q[m.X].factor
    qX._rand_gen = MockRandomGenerator(_t(eps.reshape(-1)))
    grads, losses = [], []

    class Rec(BatchInferenceLoop):
        def _exchange(self, param_dict, loss):
            grads.append(param_dict.flat.grad.clone())
            return loss

        def run(self, infr_executor, data, **kw):
            def wrapped(*a):
                out = infr_executor(*a)
                losses.append(float(out[0].detach()))
                return out
            return super(Rec, self).run(wrapped, data, **kw)
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y]), grad_loop=Rec(), dtype=DT)
    infr.initialize(Y=Y.shape)
    infr.params[qX.mean] = _t(xm)
    infr.params[qX.variance] = _t(xv)
    infr.run(Y=_t(Y), max_iter=1, learning_rate=1e-3)

    sp = O.softplus
    raw = {n: O.inv_softplus(O.T(v)).clone().requires_grad_(True) for n, v in dict(ls=ls, var=var, noise=noise, xv=xv).items()}
    lin = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(xm=xm, Z=Z).items()}
    k = O.RBF(Q, ARD=True)
    kp = {'rbf_lengthscale': sp(raw['ls'])[None], 'rbf_variance': sp(raw['var'])[None]}
    Xs = lin['xm'][None] + O.T(eps) * torch.sqrt(sp(raw['xv']))[None]
    if module == 'gp':
        logL = O.gp_log_pdf(k, Xs, O.T(Y)[None], sp(raw['noise'])[None], kp)
    else:
        logL = O.sgp_log_pdf(k, Xs, O.T(Y)[None], lin['Z'][None], sp(raw['noise'])[None], kp, jitter=1e-8)
    logq = O.normal_log_pdf(lin['xm'][None], sp(raw['xv'])[None], Xs).reshape(S, -1).sum(-1)
    logp = O.normal_log_pdf(torch.zeros(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64), Xs).reshape(S, -1).sum(-1)
    obj = -(logL + logp - logq).mean()
    obj.backward()
    assert abs(losses[0] - float(obj)) < 1e-9 * max(1.0, abs(float(obj)))
    P = infr.params
    checks = [(m.noise_var, raw['noise']), (kernel.lengthscale, raw['ls']), (kernel.variance, raw['var']), (qX.mean, lin['xm']), (qX.variance, raw['xv'])]
    if module == 'sgp':
        checks.append((m.Z, lin['Z']))
    for v, ref in checks:
        o, n, _ = P._slices[v.uuid]
        assert np.allclose(grads[0][o:o + n].cpu().numpy(), ref.grad.numpy().reshape(-1), rtol=1e-7, atol=1e-8), v


def test_forward_sampling_wrapper_and_expectation_algorithm(golden_dir):
    """ForwardSampling (forward_sampling.py:40-79) == TransferInference over ForwardSamplingAlgorithm; ExpectationAlgorithm
    (expectation.py:24-60) == the mean over the sample axis of the same forward samples (injected noise)."""
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import Inference, MAP, ForwardSampling, ExpectationAlgorithm, TransferInference
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    rng = np.random.RandomState(3)
    eps = rng.randn(4, 10, 2)
    m = _gp_model(g, rand_gen=MockRandomGenerator(_t(eps)))
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.run(X=_t(g['X']), Y=_t(g['Y']))
    fs = ForwardSampling(num_samples=4, model=m, observed=[m.X], var_tie={}, infr_params=infr.params, target_variables=[m.Y], dtype=DT)
    samples = fs.run(X=_t(g['X']))[0]
    k = O.RBF(3, ARD=True)
    kp = {'rbf_lengthscale': O.T(g['ls'])[None], 'rbf_variance': O.T(g['var'])[None]}
    ref = O.gp_sample_prior(k, O.T(g['X'])[None], O.T(g['noise'])[None], kp, O.T(eps))
    assert np.allclose(samples.cpu().numpy(), ref.numpy(), atol=1e-10)
    m.Y.factor._rand_gen = MockRandomGenerator(_t(eps))
    ex = TransferInference(ExpectationAlgorithm(model=m, observed=[m.X], num_samples=4, target_variables=[m.Y]), infr_params=infr.params, dtype=DT)
    mean = ex.run(X=_t(g['X']))[0]
    assert np.allclose(mean.cpu().numpy(), ref.numpy().mean(0), atol=1e-10)


@pytest.mark.parametrize('ls_samples,var_samples,x_samples,ard', [(False, False, False, True), (True, False, False, True), (False, True, False, True),
                                                                  (False, False, True, True), (False, False, True, False)])
def test_kernel_as_a_function_in_a_model(ls_samples, var_samples, x_samples, ard):
    """kernel_test.py:101-139 (test_kernel_as_MXFusionFunction): rbf(X_var [, X2_var], rbf_lengthscale=l_var, rbf_variance=v_var) gives a
    Variable whose factor evaluates the covariance -- the same numbers as K(...) with fetch_parameters' dictionary, for every combination
    of sampled operands the reference test runs; the factor is a FunctionEvaluation to the factor graph, its inputs are attributes."""
    from mxfusion_amd import Variable
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.components.functions.function_evaluation import FunctionEvaluation
    rng = np.random.RandomState(0)
    S, Q = 3, 2
    X = rng.rand(S if x_samples else 1, 5, Q)
    X2 = rng.rand(S if x_samples else 1, 4, Q)
    ls = rng.rand(S if ls_samples else 1, Q if ard else 1) + 1e-4
    var = rng.rand(S if var_samples else 1, 1) + 1e-4
    okern = O.RBF(Q, ARD=ard)
    for with_x2 in (False, True):
        X_mf, X2_mf, l_mf, v_mf = Variable(shape=X.shape[1:]), Variable(shape=X2.shape[1:]), Variable(shape=ls.shape[1:]), Variable(shape=var.shape[1:])
        rbf = RBF(Q, ard, 1., 1., 'rbf', None, DT)
        ev = (rbf(X_mf, X2_mf, rbf_lengthscale=l_mf, rbf_variance=v_mf) if with_x2 else rbf(X_mf, rbf_lengthscale=l_mf, rbf_variance=v_mf)).factor
        assert isinstance(ev, FunctionEvaluation) and ev.X is X_mf and ev.rbf_lengthscale is l_mf and ev.rbf_variance is v_mf
        variables = {ev.X.uuid: _t(X), ev.rbf_lengthscale.uuid: _t(ls), ev.rbf_variance.uuid: _t(var)}
        if with_x2:
            variables[ev.X2.uuid] = _t(X2)
        res_eval = ev.eval(F=None, variables=variables)
        params = rbf.fetch_parameters({rbf.lengthscale.uuid: _t(ls), rbf.variance.uuid: _t(var)})
        res_direct = rbf.K(None, _t(X), _t(X2) if with_x2 else None, **params)
        assert torch.equal(res_eval, res_direct)
        ref = okern.K(O.T(X), O.T(X2) if with_x2 else None, rbf_lengthscale=O.T(ls), rbf_variance=O.T(var))
        assert np.allclose(res_eval.cpu().numpy(), ref.numpy(), rtol=1e-12, atol=1e-14)
    with pytest.raises(TypeError):
        rbf(X_mf, rbf_lengthscal=l_mf)


@pytest.mark.parametrize('samples', [False, True])
def test_adding_an_add_kernel_flattens_and_renames(samples):
    """kernel_test.py:246-260 (test_adding_add_kernel): rbf + (rbf + linear) is ONE AddKernel of three sub-kernels named rbf, rbf0, linear
    (add_kernel.py:36-46, kernel.py:333-340), parameters add_rbf_*, add_rbf0_*, add_linear_variances, sub-kernels reachable as attributes;
    K, K(X, X2), Kdiag of the replicated kernel equal the sum of the parts (the oracle's kernels), with and without sampled operands."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Linear, AddKernel
    rng = np.random.RandomState(1)
    S, Q = (3 if samples else 1), 6
    X, X2 = rng.rand(S, 5, Q), rng.rand(S, 4, Q)
    ls, var, ls0, var0, lv = rng.rand(S, Q) + 1e-4, rng.rand(S, 1) + 1e-4, rng.rand(S, Q) + 1e-4, rng.rand(S, 1) + 1e-4, rng.rand(S, Q) + 1e-4
    kern = (RBF(Q, True, 1., 1., 'rbf', None, DT) + (RBF(Q, True, 1., 1., 'rbf', None, DT) + Linear(Q, True, 1, 'linear', None, DT))).replicate_self()
    assert isinstance(kern, AddKernel) and [k.name for k in kern.sub_kernels] == ['rbf', 'rbf0', 'linear']
    assert kern.rbf0 is kern.sub_kernels[1] and kern.linear is kern.sub_kernels[2]
    assert sorted(kern.parameter_names) == sorted(['add_rbf_lengthscale', 'add_rbf_variance', 'add_rbf0_lengthscale', 'add_rbf0_variance', 'add_linear_variances'])
    p = {'add_rbf_lengthscale': ls, 'add_rbf_variance': var, 'add_rbf0_lengthscale': ls0, 'add_rbf0_variance': var0, 'add_linear_variances': lv}
    okern = O.RBF(Q, ARD=True) + (O.RBF(Q, ARD=True) + O.Linear(Q, ARD=True))
    assert [k.name for k in okern.sub_kernels] == ['rbf', 'rbf0', 'linear']
    parts = (O.RBF(Q, ARD=True).K(O.T(X), rbf_lengthscale=O.T(ls), rbf_variance=O.T(var)) + O.RBF(Q, ARD=True).K(O.T(X), rbf_lengthscale=O.T(ls0), rbf_variance=O.T(var0))
             + O.Linear(Q, ARD=True).K(O.T(X), linear_variances=O.T(lv)))
    dp = {k: _t(v) for k, v in p.items()}
    op = {k: O.T(v) for k, v in p.items()}
    assert np.allclose(okern.K(O.T(X), **op).numpy(), parts.numpy(), rtol=1e-13)
    assert np.allclose(kern.K(None, _t(X), **dp).cpu().numpy(), parts.numpy(), rtol=1e-11, atol=1e-13)
    assert np.allclose(kern.K(None, _t(X), _t(X2), **dp).cpu().numpy(), okern.K(O.T(X), O.T(X2), **op).numpy(), rtol=1e-11, atol=1e-13)
    assert np.allclose(kern.Kdiag(None, _t(X), **dp).cpu().numpy(), okern.Kdiag(O.T(X), **op).numpy(), rtol=1e-11, atol=1e-13)


def test_float32_exact_gp_holds_the_bar_at_small_noise():
    """GPRegression in float32 at N = 2048 with noise 1e-4 (cond(K + noise I) ~ 2e7): a float32 factorisation leaves 1e-3 on the log-pdf; the
    float32 call is evaluated in float64 inside (same speed: the tile Cholesky is a float64 kernel) -- log-pdf 1e-5, gradients 1e-3 vs the oracle."""
    from mxfusion_amd.modules.gp_modules._fused import GPLogPdfFn
    rng = np.random.RandomState(0)
    N, Q = 2048, 5
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = r32(rng.uniform(-2, 2, (1, N, Q)))
    Y = r32(np.sin(X[0].sum(-1, keepdims=True)) + 0.05 * rng.standard_normal((N, 1)))[None]
    ls, var, nz = r32(np.full((1, Q), 1.5)), r32([[1.1]]), r32([[1e-4]])
    k = O.RBF(Q, ARD=True)
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in (('X', X), ('ls', ls), ('var', var), ('noise', nz))}
    ref = O.gp_log_pdf(k, lv['X'], O.T(Y), lv['noise'], {'rbf_lengthscale': lv['ls'], 'rbf_variance': lv['var']})
    gref = torch.autograd.grad(ref.sum(), list(lv.values()))
    d = lambda a: torch.as_tensor(a, dtype=torch.float32).cuda()
    t = {n: d(v).requires_grad_(True) for n, v in (('X', X), ('ls', ls), ('var', var), ('noise', nz))}
    logL, L, LinvY, info = GPLogPdfFn.apply('rbf', True, 0.0, t['X'], d(Y), t['noise'], t['ls'], t['var'])
    logL.sum().backward()
    assert int(info.abs().sum()) == 0 and logL.dtype == torch.float32 and L.dtype == torch.float32
    assert abs(float(logL[0]) - float(ref[0])) <= 1e-5 * abs(float(ref[0])), (float(logL[0]), float(ref[0]))
    for n, g in zip(lv, gref):
        e = float(np.linalg.norm(t[n].grad.double().cpu().numpy().ravel() - g.numpy().ravel()) / np.linalg.norm(g.numpy().ravel()))
        assert e <= 1e-3, (n, e)


@pytest.mark.parametrize('module', ['gp', 'sgp'])
def test_float32_combination_kernel_modules_hold_the_bar_at_small_noise(module):
    """GPRegression / SparseGPRegression with AddKernel(Matern52, RBF) in FLOAT32 through the API, noise 1e-4 on N = 600 points (GP:
    cond(K + noise I) ~ 1e7): the generic paths factor in the model's dtype as the reference does -- 1e-3 on the log-pdf in float32 -- and
    are evaluated in float64 inside since r04: 1e-5 against the oracle."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import GPRegression, SparseGPRegression
    from mxfusion_amd.inference import Inference, MAP
    rng = np.random.RandomState(0)
    N, Q, M = 600, 4, 40
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = r32(rng.uniform(-2, 2, (N, Q)))
    Y = r32(np.sin(X.sum(-1, keepdims=True)) + 0.01 * rng.randn(N, 1))
    Z = r32(rng.uniform(-2, 2, (M, Q)))
    ls1, ls2, v1, v2, noise = r32([1.3]), r32([1.7]), r32([0.9]), r32([0.4]), r32([1e-4])
    f = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()
    kern = Matern52(Q, variance=f(v1), lengthscale=f(ls1), dtype='float32') + RBF(Q, variance=f(v2), lengthscale=f(ls2), dtype='float32')
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=f(noise))
    if module == 'gp':
        m.Y = GPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, shape=(m.N, 1), dtype='float32')
    else:
        m.Z = Variable(shape=(M, Q), initial_value=f(Z))
        m.Y = SparseGPRegression.define_variable(X=m.X, kernel=kern, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype='float32')
        m.Y.factor.sgp_log_pdf.jitter = 1e-6
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype='float32')
    loss, _ = infr.run(X=f(X), Y=f(Y))
    ok = O.AddKernel([O.Matern52(Q), O.RBF(Q)])
    kp = {'add_matern52_lengthscale': O.T(ls1)[None], 'add_matern52_variance': O.T(v1)[None], 'add_rbf_lengthscale': O.T(ls2)[None], 'add_rbf_variance': O.T(v2)[None]}
    # (the parameters pass through softplus / its inverse in float32: compare at the values the inference holds)
    val = lambda v: infr.params[v].double().cpu().numpy()
    sub = {k.name: k for k in kern.sub_kernels}
    kp = {'add_matern52_lengthscale': O.T(val(sub['matern52'].lengthscale))[None], 'add_matern52_variance': O.T(val(sub['matern52'].variance))[None],
          'add_rbf_lengthscale': O.T(val(sub['rbf'].lengthscale))[None], 'add_rbf_variance': O.T(val(sub['rbf'].variance))[None]}
    nz = O.T(val(m.noise_var))[None]
    if module == 'gp':
        ref = O.gp_log_pdf(ok, O.T(X)[None], O.T(Y)[None], nz, kp)
    else:
        ref = O.sgp_log_pdf(ok, O.T(X)[None], O.T(Y)[None], O.T(val(m.Z))[None], nz, kp, jitter=1e-6)
    assert loss.dtype == torch.float32
    assert abs(float(-loss) - float(ref[0])) <= 1e-5 * abs(float(ref[0])), (float(-loss), float(ref[0]))
