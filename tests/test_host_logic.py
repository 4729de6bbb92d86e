"""CPU tests of the host-side mirror of the reference API (no kernels are launched): graph construction, variable
roles, the Module algorithm registry (modules/module.py:193-302), parameter storage, minibatch 'rollover' indexing."""
import numpy as np
import pytest
import torch

from mxfusion_amd import Model, Variable
from mxfusion_amd.components.variables import PositiveTransformation, VariableType
from mxfusion_amd.components.distributions import Normal
from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
from mxfusion_amd.modules.gp_modules import GPRegression, SVGPRegression
from mxfusion_amd.modules.gp_modules.gp_regression import GPRegressionSamplingPrediction, GPRegressionMeanVariancePrediction
from mxfusion_amd.inference import MAP, StochasticVariationalInference, create_Gaussian_meanfield
from mxfusion_amd.inference.inference_parameters import InferenceParameters
from mxfusion_amd.common.exceptions import ModelSpecificationError


def _gp():
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.kernel = RBF(input_dim=3, ARD=True, variance=1., lengthscale=np.ones(3))
    m.Y = GPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, shape=(m.N, 2))
    return m


def test_model_graph_and_variable_roles():
    m = _gp()
    assert m.Y.type == VariableType.RANDVAR and m.X.type == VariableType.PARAMETER
    gp = m.Y.factor
    assert gp.input_names == ['X', 'noise_var'] and gp.output_names == ['random_variable']
    assert gp.X is m.X and gp.random_variable is m.Y                       # factor.py:76-99 name lookup
    assert set(gp.kernel.parameters) == {'rbf_lengthscale', 'rbf_variance'}
    hidden = gp.extra_parameters()
    assert m.kernel.lengthscale in hidden and m.kernel.variance in hidden      # kernel parameters are hidden module parameters
    assert m.ordered_factors == [gp]
    assert [v.name for v in m.get_latent_variables([m.X, m.Y])] == []
    assert MAP(model=m, observed=[m.X, m.Y]).observed_variable_names == ['X', 'Y']


def test_module_algorithm_registry_is_the_plugin_point():
    m = _gp()
    gp = m.Y.factor
    assert isinstance(gp.gp_predict, GPRegressionMeanVariancePrediction)
    # swap the prediction algorithm exactly like testing/modules/gpregression_test.py:270-278 / the GP notebook cell 24
    alg = GPRegressionSamplingPrediction(gp._module_graph, gp._extra_graphs[0], [gp._module_graph.X])
    gp.attach_prediction_algorithms(targets=gp.output_names, conditionals=gp.input_names, algorithm=alg, alg_name='gp_predict')
    assert gp.gp_predict is alg
    assert len(gp._prediction_algorithms[('X', 'noise_var')]) == 1                # replaced, not appended (module.py:262-302)
    got = gp._get_algorithm_for_target_conditional_pair(gp._prediction_algorithms, ('random_variable',), ('X', 'noise_var'), exact_match=True)
    assert got is alg
    with pytest.raises(ModelSpecificationError):
        gp._get_algorithm_for_target_conditional_pair(gp._prediction_algorithms, ('random_variable',), ('X',), exact_match=True)
    gp.gp_log_pdf.jitter = 1e-6                                                     # algorithm knobs are plain attributes
    assert gp.gp_log_pdf.jitter == 1e-6


def test_combination_kernels_prefix_parameters():
    k = Matern52(4, name='matern52') + RBF(4)
    assert set(k.parameters) == {'add_matern52_lengthscale', 'add_matern52_variance', 'add_rbf_lengthscale', 'add_rbf_variance'}
    k2 = RBF(2) * RBF(2)
    assert len(k2.parameters) == 4            # duplicate sub-kernel names are disambiguated
    assert (RBF(2)).fused_spec() == ('rbf', False) and k.fused_spec() is None


def test_svi_model_posterior_and_parameters_on_cpu():
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, 3))
    m.Z = Variable(shape=(5, 3), initial_value=np.arange(15.).reshape(5, 3))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=RBF(3, ARD=True), noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1))
    gp = m.Y.factor
    post = gp._extra_graphs[0]
    assert post.qU_mean.shape == (5, 1) and post.qU_cov_W.shape == (5, 5) and post.qU_cov_diag.shape == (5,)
    q = create_Gaussian_meanfield(model=m, observed=[m.Y])
    qX = q[m.X].factor
    assert isinstance(qX, Normal) and q[m.X].uuid == m.X.uuid                 # posterior replicas keep the UUID
    alg = StochasticVariationalInference(model=m, posterior=q, num_samples=4, observed=[m.Y])
    params = InferenceParameters(dtype='float64', context=torch.device('cpu'))
    params.update_constants({m.N.uuid: 7})
    params.initialize_params(alg.graphs, alg.observed_variable_UUIDs, seed=0)
    names = {v.uuid for v in (m.Z, m.noise_var, post.qU_mean, post.qU_cov_W, post.qU_cov_diag, qX.mean, qX.variance,
                              gp.kernel.lengthscale, gp.kernel.variance)}
    assert names == set(params._slices)                                      # exactly the trainable parameters, N/X/Y excluded
    assert params.flat.numel() == 15 + 1 + 5 + 25 + 5 + 21 + 21 + 3 + 1 and params.flat.requires_grad
    assert torch.allclose(params.raw(m.Z), torch.arange(15., dtype=torch.float64).reshape(5, 3))
    # positive parameters are stored unconstrained: raw = log(expm1(value)) (var_trans.py:91)
    assert torch.allclose(params.raw(m.noise_var), torch.log(torch.expm1(torch.tensor([0.01], dtype=torch.float64))))
    params[post.qU_mean] = np.ones((5, 1))
    assert torch.all(params.raw(post.qU_mean) == 1)
    params[qX.variance] = np.full((7, 3), 0.5)
    assert torch.allclose(params.raw(qX.variance), torch.log(torch.expm1(torch.tensor(0.5, dtype=torch.float64))))
    views = params.tensors()
    (views[m.Z.uuid].sum() * 2).backward()                                  # views are autograd-connected to the ONE flat leaf
    o, n, _ = params._slices[m.Z.uuid]
    assert torch.all(params.flat.grad[o:o + n] == 2) and params.flat.grad.abs().sum() == 2 * 15


def test_minibatch_rollover_indexing(monkeypatch):
    """minibatch_loop.py:65-69: shuffle=True, last_batch='rollover' -- the remainder opens the next epoch; every batch is full."""
    from mxfusion_amd.inference import minibatch_loop as ml

    class FakeTrainer(object):
        def __init__(self, *a, **k):
            self.steps = []

        def step(self, batch_size=1):
            self.steps.append(batch_size)
    monkeypatch.setattr(ml, '_Adam', FakeTrainer)
    seen = []

    def executor(xb):
        seen.append(xb.clone())
        loss = (xb.sum() * 0).requires_grad_(True)
        return loss, loss
    data = [torch.arange(10.)]
    loop = ml.MinibatchInferenceLoop(batch_size=4, rv_scaling=None)
    loop.run(executor, data, param_dict=None, ctx=None, max_iter=2)
    assert all(b.numel() == 4 for b in seen) and len(seen) == 5          # 10 + 10 samples -> 2 + 3 full batches, 0 left
    assert sorted(torch.cat(seen).tolist()) == sorted(list(range(10)) * 2)


def test_grad_transfer_parameters_fixed_model_trainable_policy_on_cpu():
    """GradTransferInference's parameter handling (grad_based_inference.py:124-140): inherited parameters are carried over and fixed;
    the caller's `train_params` are re-homed into one flat buffer whose .grad is the flat gradient the optimiser / all-reduce see."""
    from mxfusion_amd.inference import GradTransferInference, PILCOAlgorithm
    m = _gp()
    alg0 = MAP(model=m, observed=[m.X, m.Y])
    trained = InferenceParameters(dtype='float64', context=torch.device('cpu'))
    trained.initialize_params(alg0.graphs, alg0.observed_variable_UUIDs, seed=0)
    trained[m.noise_var] = np.array([0.3])
    policy = torch.nn.Linear(3, 1)                                            # float32 module: re-homed in the inference dtype
    w0 = policy.weight.detach().clone().double()
    alg = PILCOAlgorithm(model=m, observed=[m.X, m.Y], cost_function=None, policy=policy, n_time_steps=2,
                         initial_state_generator=None, num_samples=3)
    infr = GradTransferInference(alg, infr_params=trained, train_params=list(policy.parameters()), dtype='float64', context=torch.device('cpu'))
    infr.initialize(X=(5, 3), Y=(5, 2))
    p = infr.params
    assert np.allclose(np.log1p(np.exp(p.raw(m.noise_var).numpy())), 0.3)                 # carried over (stored unconstrained) ...
    assert not any(t.requires_grad for t in p.tensors().values())                          # ... and fixed
    assert p.flat.numel() == 4 and p.flat.dtype == torch.float64 and policy.weight.dtype == torch.float64
    assert torch.allclose(policy.weight.detach(), w0) and policy.weight.data_ptr() == p.flat.data_ptr()
    (policy(torch.ones(2, 3, dtype=torch.float64)).sum() * 3).backward()
    assert torch.allclose(p.flat.grad, torch.tensor([6., 6., 6., 6.], dtype=torch.float64))   # weight grads 3*2 each, bias grad 3*2
    with torch.no_grad():
        p.flat.detach().sub_(1.0)                                                          # an optimiser step on the flat buffer ...
    assert torch.allclose(policy.weight.detach(), w0 - 1.0)                                # ... is a step on the module's tensors
    p.zero_grad()
    assert float(p.flat.grad.abs().sum()) == 0 and policy.weight.grad.data_ptr() == p.flat.grad.data_ptr()


# ---- the float32 guard's level logic (modules/gp_modules/_fused.py), with the library's condition slots replaced by a dict -----------------
class _FakeSlots(object):
    def __init__(self, monkeypatch):
        from mxfusion_amd import _lib, ops
        self.slots, self.last_cond, self.configured = {}, 0.0, []
        monkeypatch.setattr(ops, '_device_index', lambda dev: 0)
        monkeypatch.setattr(ops, 'svgp_last_cond', lambda dev=None: self.last_cond)
        monkeypatch.setattr(_lib, 'svgp_configure', lambda dev, form, slot: self.configured.append((form, slot)))

        def cond_slot(dev, slot, reset=False):
            v = self.slots.get(slot, (0.0, 0.0))
            if reset:
                self.slots[slot] = (0.0, 0.0)
            return v
        monkeypatch.setattr(_lib, 'svgp_cond_slot', cond_slot)

    def publish(self, guard, cond):           # what a finished call's last launch does
        last, mx = self.slots.get(guard.slot, (0.0, 0.0))
        self.slots[guard.slot] = (cond, max(mx, cond))
        self.last_cond = cond


def test_float32_guard_levels_first_call_and_hysteresis(monkeypatch):
    import warnings
    from mxfusion_amd import _lib
    from mxfusion_amd.modules.gp_modules import _fused
    G = _fused.Float32Guard
    fake = _FakeSlots(monkeypatch)
    g = G('unit')
    ran = []

    def call(cond, whitened_ok=True, is_f32=True):
        def run(tier):
            ran.append(tier)
            fake.publish(g, cond)
            return tier
        return _fused._guarded(g, 'cuda', is_f32, whitened_ok, run)

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        # first call, ill-conditioned: runs explicit, the synchronous check re-runs it whitened
        assert call(1e5) == G.WHITENED and ran == [G.EXPLICIT, G.WHITENED] and g.tier == G.WHITENED
        assert fake.configured[-2] == (_lib.FORM_WHITENED, g.slot) and len(w) == 1
        assert fake.configured[-1] == (_lib.FORM_EXPLICIT, 0)     # the handle's form / slot never outlive the guarded call
        # further calls poll the slot (no synchronous check any more); a call the whitened form does not cover runs float64 without moving the owner
        assert call(1e5) == G.WHITENED
        assert call(1e5, whitened_ok=False) == G.F64 and g.tier == G.WHITENED
        # beyond the whitened range: the call AFTER the one that published it moves up
        assert call(1e8) == G.WHITENED
        assert call(1e8) == G.F64 and g.tier == G.F64
        # down only below a quarter of a limit, one level set at a time by what was published
        assert call(0.5 * G.LIMIT_WHITENED) == G.F64            # (published 1e8 before this poll)
        assert call(0.5 * G.LIMIT_WHITENED) == G.F64            # 2.5e6 > 0.25 * 5e6: stays
        assert call(0.2 * G.LIMIT_WHITENED) == G.F64
        assert call(0.2 * G.LIMIT_WHITENED) == G.WHITENED       # 1e6 < 1.25e6: whitened again
        assert call(10.0) == G.WHITENED
        assert call(10.0) == G.EXPLICIT
        assert g.cond_max == 1e8 and g.switches == 4
    # float64 inputs: the form is irrelevant, the slot is still configured; a disabled guard never leaves the explicit form
    n = len(ran)
    assert call(1e9, is_f32=False) == G.EXPLICIT and len(ran) == n + 1
    monkeypatch.setattr(G, 'enabled', False)
    g2 = G('off')
    fake.last_cond = 1e9
    assert _fused._guarded(g2, 'cuda', True, True, lambda t: t) == G.EXPLICIT and g2.tier == G.EXPLICIT
    monkeypatch.setattr(G, 'enabled', True)
    # a forced level (bench.py --f32-form)
    monkeypatch.setattr(G, 'force', G.WHITENED)
    g3 = G('forced')
    assert _fused._guarded(g3, 'cuda', True, True, lambda t: t) == G.WHITENED
    assert _fused._guarded(g3, 'cuda', True, False, lambda t: t) == G.F64
    monkeypatch.setattr(G, 'force', None)
    # two owners never share a slot while fewer than 63 exist, and the report names each
    assert g.slot != g2.slot != g3.slot
    rep = G.report()
    assert rep['kuu_cond_max'] >= 1e8 and any('unit' in k for k in rep['float32_tiers'])


def test_module_clone():
    """gpregression_test.py:369-377, svgpregression_test.py / sparsegpregression_test.py test_module_clone: Model.clone() of a model holding a
    GP module -- same UUIDs, names and topology, new component objects (module, its internal graphs and algorithms, the kernel and its
    parameter Variables), constants shared; a cloned SVGP module has a float32 guard of its own."""
    import torch
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression, SVGPRegression, SparseGPRegression
    for cls, extra in ((GPRegression, {}), (SVGPRegression, {'num_inducing': 4}), (SparseGPRegression, {'num_inducing': 4})):
        m = Model()
        m.N = Variable()
        X0 = torch.zeros(2, 3, dtype=torch.float64)
        kernel = RBF(input_dim=3, ARD=True, variance=torch.ones(1, dtype=torch.float64), lengthscale=torch.ones(3, dtype=torch.float64), dtype='float64')
        m.Y = cls.define_variable(X=X0, kernel=kernel, noise_var=torch.ones(1, dtype=torch.float64), dtype='float64', **extra)
        c = m.clone()
        assert type(c) is type(m) and c is not m
        assert sorted(c._variables) == sorted(m._variables)
        assert c.Y.uuid == m.Y.uuid and c.Y is not m.Y and c.Y.name == 'Y'
        f, g = m.Y.factor, c.Y.factor
        assert type(g) is cls and g is not f
        assert g.kernel is not f.kernel and g.kernel.lengthscale.uuid == f.kernel.lengthscale.uuid and g.kernel.lengthscale is not f.kernel.lengthscale
        assert [n for n, _ in g.inputs] == [n for n, _ in f.inputs]
        alg_f, alg_g = f._log_pdf_algorithms, g._log_pdf_algorithms
        assert len(alg_f) == len(alg_g) and all(a is not b for a, b in zip(alg_f.values(), alg_g.values())) if isinstance(alg_f, dict) else True
        # constants keep their identity (no array is copied)
        cx = [v for _, v in g.inputs if v.isConstant and v._value is not None]
        assert any(v._value is X0 for v in cx)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=RBF(3, ARD=True, dtype='float32'), noise_var=torch.ones(1), num_inducing=8, shape=(m.N, 1), dtype='float32')
    g0 = m.Y.factor.svgp_log_pdf._f32_guard()
    g1 = m.clone().Y.factor.svgp_log_pdf._f32_guard()
    assert g0 is not g1 and g0.slot != g1.slot


def test_set_prior_module_variable_access_and_graph_printing():
    """variable_test.py:28-39 (set_prior puts the distribution and its inputs into the variable's graph), factor_graph_test.py:431-438
    (a module's kernel parameter is reachable from the model by UUID; printing a graph lists its factors)."""
    import torch
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    m = Model(verbose=False)
    m.x = Variable()
    d = Normal(mean=torch.tensor([0.]), variance=torch.tensor([1e6]))
    m.x.set_prior(d)
    assert m.x.factor is d and any(f is d for f in m._factors) and all(v.uuid in m for _, v in d.inputs)
    m1 = Model()
    m1.N = Variable()
    m1.X = Variable(shape=(m1.N, 3))
    m1.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m1.kernel = RBF(input_dim=3, variance=1, lengthscale=1)
    m1.Y = GPRegression.define_variable(X=m1.X, kernel=m1.kernel, noise_var=m1.noise_var, shape=(m1.N, 1))
    l = m1.Y.factor.kernel.lengthscale
    assert m1[l.uuid] == l
    txt = str(m1)
    assert txt.startswith('Model (') and '~ GPRegression(' in txt and 'noise_var=' in txt
    assert '~ Normal(mean=' in str(m)


def test_replicate_and_reconcile_gp_model(tmp_path):
    """factor_graph_test.py:158-165 (test_replicate_gp_model), :288-292 (test_reconcile_gp_model), :368-395
    (test_save_reload_then_reconcile_gp_module), through the reference's own entry points on FactorGraph: the clone holds the same
    components (model and the module's internal graphs, same shapes); two independently built copies of the script reconcile 1:1 over
    every component of the model and of the module's graphs; so does a copy saved to JSON and loaded back."""
    import json
    import torch
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.models import FactorGraph
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression

    def make():
        m = Model()
        m.N = Variable()
        m.X = Variable(shape=(m.N, 3))
        m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=torch.tensor([1.]))
        kernel = RBF(input_dim=3, variance=torch.tensor([1.]), lengthscale=torch.tensor([1.]))
        m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, 2))
        return m

    def uuids(m):
        gp = m.Y.factor
        s = set(m.variables)                                # (the model's variables + those of the module's internal graphs)
        for g in [gp._module_graph] + list(gp._extra_graphs):
            s |= set(g.variables)
        return s
    m = make()
    m2 = m.clone()
    assert uuids(m) == uuids(m2)
    assert all(tuple(getattr(s, 'uuid', s) for s in m[u].shape) == tuple(getattr(s, 'uuid', s) for s in m2[u].shape) for u in m.variables)
    m1, m3 = make(), make()
    cmap = FactorGraph.reconcile_graphs([m1], m3)
    assert len(set(cmap.values())) == len(cmap)                       # 1:1
    assert uuids(m3) <= set(cmap) and set(cmap[u] for u in uuids(m3)) == uuids(m1)
    f = str(tmp_path / 'graph.json')
    FactorGraph.save(f, m3.as_json())
    loaded = FactorGraph.load_graphs([json.load(open(f))])[0]
    cmap2 = FactorGraph.reconcile_graphs([m1], loaded)
    assert {k: v for k, v in cmap2.items() if k in uuids(m3)} == {k: v for k, v in cmap.items() if k in uuids(m3)}


def test_row_sharding_weights_global_factors_and_refuses_non_additive_modules():
    """prepare_executor(rv_scaling, global_weight) of a row-sharded data-parallel loop: factors named in rv_scaling keep their N / B scaling,
    every other distribution factor (a prior of a global variable) carries the weight 1 / world -- restored by a later unsharded run --; an
    SVGP module takes the weight on its KL term only; a module whose bound is not a sum over rows (exact GP) refuses."""
    from mxfusion_amd.common.exceptions import InferenceError
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.w = Normal.define_variable(mean=0., variance=1., shape=(3,))                 # a global latent variable with a prior
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=RBF(3, ARD=True), noise_var=m.noise_var, num_inducing=4, shape=(m.N, 1))
    alg = MAP(model=m, observed=[m.X, m.Y])
    alg.prepare_executor(rv_scaling={m.Y.uuid: 8.0}, global_weight=0.125)
    assert m.w.factor.log_pdf_scaling == 0.125
    assert m.Y.factor.log_pdf_scaling == 8.0 and m.Y.factor.global_weight == 0.125 and m.Y.factor._rows_sharded
    alg.prepare_executor(rv_scaling={m.Y.uuid: 8.0}, global_weight=0.25)           # (weights do not compound)
    assert m.w.factor.log_pdf_scaling == 0.25
    alg.prepare_executor(rv_scaling={m.Y.uuid: 8.0})
    assert m.w.factor.log_pdf_scaling == 1 and m.Y.factor.global_weight == 1.0

    class Probe(object):                                                           # what Module.log_pdf hands its algorithm
        kl_weight, seen = 1.0, None

        def compute(self, F, variables):
            Probe.seen = (self.log_pdf_scaling, self.kl_weight)
            return torch.ones(1)
    gp = m.Y.factor
    alg.prepare_executor(rv_scaling={m.Y.uuid: 8.0}, global_weight=0.125)
    gp._get_algorithm_for_target_conditional_pair = lambda *a, **k: Probe()
    variables = {v.uuid: None for _, v in gp.inputs}
    variables[m.Y.uuid] = None
    assert float(gp.log_pdf(None, variables)) == 1.0 and Probe.seen == (8.0, 0.125)
    alg.prepare_executor(rv_scaling={}, global_weight=0.125)                       # the module's output not sharded: the whole module is global
    gp.log_pdf_scaling = 1
    assert float(gp.log_pdf(None, variables)) == 0.125 and Probe.seen == (1, 1.0)

    g = _gp()
    with pytest.raises(InferenceError):
        MAP(model=g, observed=[g.X, g.Y]).prepare_executor(rv_scaling={g.Y.uuid: 1.0}, global_weight=0.5)
