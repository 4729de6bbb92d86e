"""GPU parity of the host-side API mirror (Model / Variable / kernels / modules / Inference): the reference's
own test programs (testing/modules/gpregression_test.py, svgpregression_test.py, the GP notebook) re-run through
mxfusion_amd with device tensors, checked against the golden fixtures and the reference-recorded outputs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = 'float64'


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()


def _gp_model(noise_var, lengthscale, variance, D):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(noise_var))
    kernel = RBF(input_dim=3, ARD=True, variance=_t(variance), lengthscale=_t(lengthscale), dtype=DT)
    m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, D), dtype=DT)
    return m


def test_gp_log_pdf_and_prediction_like_reference_tests(golden_dir):
    """testing/modules/gpregression_test.py:80-96 (test_log_pdf) and :168-226 (test_prediction)."""
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    m = _gp_model(g['noise'], g['ls'], g['var'], 2)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    loss, _ = infr.run(X=_t(g['X']), Y=_t(g['Y']))
    assert abs(float(-loss) - (-18.814420362103)) < 1e-9
    for nf in (True, False):
        for dg in (True, False):
            infr2 = TransferInference(ModulePredictionAlgorithm(m, observed=[m.X], target_variables=[m.Y]), infr_params=infr.params, dtype=DT)
            gp = infr2.inference_algorithm.model.Y.factor
            gp.gp_predict.noise_free = nf
            gp.gp_predict.diagonal_variance = dg
            res = infr2.run(X=_t(g['Xt']))[0]
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            assert np.allclose(res[0].cpu().numpy(), g['mu_' + tag], atol=1e-9), tag     # north_star: 1e-5 rel on mean/variance
            assert np.allclose(res[1].cpu().numpy(), g['var_' + tag], atol=1e-9), tag
            assert res[1].shape == ((1, 20) if dg else (1, 20, 20))


def test_gp_notebook_trajectory_through_the_api(golden_dir):
    """examples/notebooks/gp_regression.ipynb cells 4-14: 100 Adam iterations through GradBasedInference(MAP) on the
    MI355X reproduce the losses MXFusion itself printed."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.common import config
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
    rec = json.load(open(os.path.join(golden_dir, 'reference_recorded.json')))
    np.random.seed(0)
    X = np.random.uniform(-3., 3., (20, 1))
    Y = np.sin(X) + np.random.randn(20, 1) * 0.05
    old = config.DEFAULT_DTYPE
    config.DEFAULT_DTYPE = 'float64'
    try:
        m = Model()
        m.N = Variable()
        m.X = Variable(shape=(m.N, 1))
        m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
        m.kernel = RBF(input_dim=1, variance=1, lengthscale=1)
        m.Y = GPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, shape=(m.N, 1))

        losses = {}

        class Rec(BatchInferenceLoop):
            def run(self, infr_executor, data, **kw):
                def wrapped(*a):
                    out = infr_executor(*a)
                    losses[len(losses) + 1] = float(out[0])
                    return out
                return super(Rec, self).run(wrapped, data, **kw)
        infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.X, m.Y]), grad_loop=Rec())
        infr.run(X=_t(X), Y=_t(Y), max_iter=100, learning_rate=0.05)
        for it, ref in rec['gp_notebook_loss_trajectory'].items():
            assert abs(losses[int(it)] - ref) <= 3e-6 * abs(ref), (it, losses[int(it)], ref)
        assert abs(float(infr.params[m.kernel.variance]) - 0.616992) < 1e-6
        assert abs(float(infr.params[m.kernel.lengthscale]) - 1.649073) < 1e-6
        assert abs(float(infr.params[m.noise_var]) - 0.002251) < 1e-6
    finally:
        config.DEFAULT_DTYPE = old


def _svgp_model(g, latent_X=False):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, 3)) if latent_X else Variable(shape=(m.N, 3))
    m.Z = Variable(shape=(3, 3), initial_value=_t(g['Z']))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=_t(g['noise']))
    kernel = RBF(input_dim=3, ARD=True, variance=_t(g['var']), lengthscale=_t(g['ls']), dtype=DT)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=DT)
    m.Y.factor.svgp_log_pdf.jitter = 1e-8
    return m, m.Y.factor


def test_svgp_log_pdf_and_prediction_like_reference_tests(golden_dir):
    """testing/modules/svgpregression_test.py:91-115 (test_log_pdf) and :170-242 (test_prediction)."""
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    g = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    m, gp = _svgp_model(g)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.initialize(X=g['X'].shape, Y=g['Y'].shape)
    infr.params[gp._extra_graphs[0].qU_mean] = _t(g['qm'])
    infr.params[gp._extra_graphs[0].qU_cov_W] = _t(g['qW'])
    infr.params[gp._extra_graphs[0].qU_cov_diag] = _t(g['qd'])
    loss, _ = infr.run(X=_t(g['X']), Y=_t(g['Y']))
    assert abs(float(-loss) - (-32.725635407458)) < 1e-8
    for nf in (True, False):
        for dg in (True, False):
            infr2 = TransferInference(ModulePredictionAlgorithm(m, observed=[m.X], target_variables=[m.Y]), infr_params=infr.params, dtype=DT)
            gp.svgp_predict.noise_free = nf
            gp.svgp_predict.diagonal_variance = dg
            res = infr2.run(X=_t(g['Xt']))[0]
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            assert np.allclose(res[0].cpu().numpy(), g['mu_' + tag], atol=1e-8), tag
            assert np.allclose(res[1].cpu().numpy(), g['var_' + tag], atol=1e-8), tag
            assert res[1].shape == ((1, 5, 1) if dg else (1, 5, 5, 1))


def test_svi_with_samples_matches_oracle_trajectory(golden_dir):
    """The MC path of testing/modules/svgpregression_test.py:357-385 (the reference only smoke-tests it): latent
    X ~ N(0,1), mean-field q(X), S injected-noise samples, 3 Adam iterations -- losses, first gradient and final
    parameters against the oracle's trajectory (tests/golden/kat_svi.npz)."""
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, create_Gaussian_meanfield, BatchInferenceLoop
    from oracle import gp_oracle as O
    g = np.load(os.path.join(golden_dir, 'kat_svi.npz'))
    k = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    m, gp = _svgp_model(k, latent_X=True)
    q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=DT)
    qX = q[m.X].factor
    qX._rand_gen = MockRandomGenerator(_t(g['eps']))
    S = g['eps'].shape[1]
    losses, grads = [], []

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
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y]), grad_loop=Rec(), dtype=DT)
    infr.initialize(Y=g['Y'].shape)
    sp = lambda raw: O.softplus(O.T(raw)).numpy()
    infr.params[gp._extra_graphs[0].qU_mean] = _t(g['init_qU_mean'])
    infr.params[gp._extra_graphs[0].qU_cov_W] = _t(g['init_qU_cov_W'])
    infr.params[gp._extra_graphs[0].qU_cov_diag] = _t(sp(g['init_qU_cov_diag']))
    infr.params[qX.mean] = _t(g['init_qX_mean'])
    infr.params[qX.variance] = _t(sp(g['init_qX_var']))
    infr.run(Y=_t(g['Y']), max_iter=3, learning_rate=0.1)
    assert np.allclose(losses[:3], g['losses'], rtol=1e-9)
    P = infr.params
    got0 = {n: grads[0][P._slices[v.uuid][0]:P._slices[v.uuid][0] + P._slices[v.uuid][1]].cpu().numpy().reshape(g['g0_' + n].shape)
            for n, v in (('qX_mean', qX.mean), ('qX_var', qX.variance), ('Z', m.Z), ('noise_var', m.noise_var),
                         ('qU_mean', gp._extra_graphs[0].qU_mean), ('qU_cov_W', gp._extra_graphs[0].qU_cov_W),
                         ('qU_cov_diag', gp._extra_graphs[0].qU_cov_diag), ('lengthscale', gp.kernel.lengthscale),
                         ('variance', gp.kernel.variance))}
    for n, a in got0.items():
        assert np.allclose(a, g['g0_' + n], rtol=1e-7, atol=1e-8), n
    for n, v in (('qX_mean', qX.mean), ('Z', m.Z), ('qU_mean', gp._extra_graphs[0].qU_mean), ('qU_cov_W', gp._extra_graphs[0].qU_cov_W)):
        assert np.allclose(P.raw(v).cpu().numpy(), g['final_' + n], rtol=1e-7, atol=1e-8), n
    for n, v in (('qX_var', qX.variance), ('noise_var', m.noise_var), ('lengthscale', gp.kernel.lengthscale), ('variance', gp.kernel.variance),
                 ('qU_cov_diag', gp._extra_graphs[0].qU_cov_diag)):
        assert np.allclose(P.raw(v).cpu().numpy().reshape(g['final_' + n].shape), g['final_' + n], rtol=1e-7, atol=1e-8), n


def test_variational_posterior_forward_sampling_draws_latents_from_q(golden_dir):
    """forward_sampling.py:118-160: after an SVI run, VariationalPosteriorForwardSampling samples the latent variables from the learned q
    (here q(X) = N(mean, variance) with injected noise: X_s = mean + sqrt(variance) eps_s exactly; Y stays observed, so the model's ancestral
    pass has nothing left to draw); a non-variational inherited inference is rejected."""
    from mxfusion_amd.common.exceptions import InferenceError
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.inference import (GradBasedInference, StochasticVariationalInference, create_Gaussian_meanfield, BatchInferenceLoop,
                                        VariationalPosteriorForwardSampling, ForwardSamplingAlgorithm, Inference)
    g = np.load(os.path.join(golden_dir, 'kat_svi.npz'))
    k = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    m, gp = _svgp_model(k, latent_X=True)
    q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=DT)
    qX = q[m.X].factor
    qX._rand_gen = MockRandomGenerator(_t(g['eps']))
    S = g['eps'].shape[1]
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y]), grad_loop=BatchInferenceLoop(), dtype=DT)
    infr.initialize(Y=g['Y'].shape)
    infr.run(Y=_t(g['Y']), max_iter=2, learning_rate=0.1)
    qX._rand_gen = MockRandomGenerator(_t(g['eps'][0]))
    vp = VariationalPosteriorForwardSampling(num_samples=S, observed=[m.Y], inherited_inference=infr, target_variables=[m.X], dtype=DT)
    Xs, = vp.run(Y=_t(g['Y']))
    mean, var = infr.params[qX.mean].double(), infr.params[qX.variance].double()
    eps = _t(g['eps'][0]).double().reshape((S,) + tuple(mean.shape[-2:]))
    want = mean.reshape((1,) + tuple(mean.shape[-2:])) + var.reshape((1,) + tuple(var.shape[-2:])).sqrt() * eps
    assert Xs.shape == want.shape
    assert torch.allclose(Xs.double(), want, rtol=1e-12, atol=1e-12)
    with pytest.raises(InferenceError):
        VariationalPosteriorForwardSampling(num_samples=2, observed=[], inherited_inference=Inference(ForwardSamplingAlgorithm(model=m, observed=[], num_samples=2), dtype=DT))


def test_a_cloned_model_computes_the_same_bound_and_prints_its_parameters(golden_dir):
    """gpregression_test.py:352-377 (test_prediction_print, test_module_clone): the clone of the KAT-GP model evaluates to the reference's
    log-pdf (-18.8144...) through its own Inference, and print_params lists the inference's parameters."""
    from mxfusion_amd.inference import Inference, MAP
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    m = _gp_model(g['noise'], g['ls'], g['var'], 2)
    c = m.clone()
    assert c.Y.factor is not m.Y.factor and c.Y.uuid == m.Y.uuid
    infr = Inference(MAP(model=c, observed=[c.X, c.Y]), dtype=DT)
    loss, _ = infr.run(X=_t(g['X']), Y=_t(g['Y']))
    assert abs(float(-loss) - (-18.814420362103)) < 1e-9
    txt = infr.print_params()
    assert isinstance(txt, str) and len(txt) > 1 and 'Model' in txt


def _normal_normal_model():
    """testing/inference/map_test.py:49-55: x ~ N(mean, var) with free mean / positive var, y ~ N(x, 1), both of length N."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    m = Model()
    m.mean = Variable()
    m.var = Variable(transformation=PositiveTransformation())
    m.N = Variable()
    m.x = Normal.define_variable(mean=m.mean, variance=m.var, shape=(m.N,), dtype=DT)
    m.y = Normal.define_variable(mean=m.x, variance=_t([1.]), shape=(m.N,), dtype=DT)
    return m


def test_map_examples_of_the_reference(golden_dir):
    """map_test.py:74-109: MAP with x latent (test_one_map_example), with x observed (test_function_map_example), and the outcome of a
    MAP inference handed to VariationalPosteriorForwardSampling(10, [m.x], infr, [m.y]) (test_inference_outcome_passing_success):
    the losses fall, and the forward samples of y are N(x, 1) around the observed x."""
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop, VariationalPosteriorForwardSampling
    rng = np.random.RandomState(0)
    D = 10
    y, x = rng.rand(D), rng.rand(D)

    def run(observed_names):
        m = _normal_normal_model()
        losses = []

        class Rec(BatchInferenceLoop):
            def run(self, infr_executor, data, **kw):
                def wrapped(*a):
                    out = infr_executor(*a)
                    losses.append(float(out[0].detach()))
                    return out
                return super(Rec, self).run(wrapped, data, **kw)
        infr = GradBasedInference(MAP(model=m, observed=[getattr(m, n) for n in observed_names]), grad_loop=Rec(), dtype=DT)
        infr.run(max_iter=10, learning_rate=0.05, **{n: _t({'y': y, 'x': x}[n]) for n in observed_names})
        assert len(losses) >= 10 and all(np.isfinite(losses)) and losses[-1] < losses[0]
        return m, infr
    run(['y'])
    m, infr = run(['y', 'x'])
    infr2 = VariationalPosteriorForwardSampling(10, [m.x], infr, [m.y], dtype=DT)
    ys, = infr2.run(x=_t(x))
    assert ys.shape == (10, D) and torch.isfinite(ys).all()
    assert float((ys.mean(0) - _t(x)).abs().max()) < 2.0          # 10 draws of N(x, 1)


def test_change_default_dtype():
    """inference_alg_test.py:65-83: with config.DEFAULT_DTYPE = 'float64' a model built without dtype arguments runs in float64."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.common import config
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.inference import GradBasedInference, MAP
    old = config.DEFAULT_DTYPE
    config.DEFAULT_DTYPE = 'float64'
    try:
        rng = np.random.RandomState(0)
        data = rng.randn(100) * np.sqrt(5.) + 3.
        m = Model()
        m.mu = Variable()
        m.s = Variable(transformation=PositiveTransformation())
        m.Y = Normal.define_variable(mean=m.mu, variance=m.s, shape=(100,))
        infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.Y]))
        infr.run(Y=_t(data), learning_rate=0.1, max_iter=2)
        assert infr.params[m.mu].dtype == torch.float64 and infr.params[m.s].dtype == torch.float64
    finally:
        config.DEFAULT_DTYPE = old


@pytest.mark.parametrize('name,opts', [('rmsprop', {}), ('adagrad', {}), ('adadelta', {'rho': 0.95}), ('nag', {'momentum': 0.9})])
def test_trainer_seam_takes_the_other_mxnet_optimisers_by_name(name, opts):
    """batch_loop.py:46-49 hands any optimiser NAME to gluon.Trainer; here 'rmsprop' / 'adagrad' / 'adadelta' / 'nag' drive the same fused
    flat-buffer update (mxf_opt_step).  The GP notebook model, 5 MAP iterations through GradBasedInference.run: raw parameters against the oracle's
    objective + the oracle's restatement of MXNet's rule (the reference holds no vector for these rules: API knowledge, parity unpinned)."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP
    from oracle import gp_oracle as O
    rng = np.random.RandomState(0)
    X = rng.uniform(-3., 3., (20, 1))
    Y = np.sin(X) + rng.randn(20, 1) * 0.05
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 1))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=_t([0.01]))
    m.kernel = RBF(input_dim=1, variance=_t([1.0]), lengthscale=_t([1.0]), dtype=DT)
    m.Y = GPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, shape=(m.N, 1), dtype=DT)
    infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    infr.run(X=_t(X), Y=_t(Y), max_iter=5, learning_rate=0.05, optimizer=(name, opts) if opts else name)
    kern = O.RBF(1, ARD=False)
    raw = {'lengthscale': O.inv_softplus(O.T([1.0])), 'variance': O.inv_softplus(O.T([1.0])), 'noise_var': O.inv_softplus(O.T([0.01]))}
    p1 = list(opts.values())[0] if opts else None
    rules = {k: O.MXNetRule(name, 0.05, p1=p1) for k in raw}
    for _ in range(5):
        lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        O.map_gp_loss(kern, O.T(X), O.T(Y), lv).backward()
        raw = {k: rules[k].step(lv[k].detach(), lv[k].grad) for k in lv}
    sp = O.softplus
    assert abs(float(infr.params[m.kernel.lengthscale]) - float(sp(raw['lengthscale']))) < 1e-9
    assert abs(float(infr.params[m.kernel.variance]) - float(sp(raw['variance']))) < 1e-9
    assert abs(float(infr.params[m.noise_var]) - float(sp(raw['noise_var']))) < 1e-9
    assert abs(float(sp(raw['lengthscale'])) - 1.0) > 1e-3           # the rule moved the parameters


def test_unknown_optimiser_name_is_refused():
    from mxfusion_amd.inference.batch_loop import _Adam

    class P(object):
        flat = torch.zeros(3, device='cuda')
    with pytest.raises(NotImplementedError):
        _Adam(P(), 0.1, 'ftrl')
