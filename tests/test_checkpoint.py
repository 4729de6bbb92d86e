"""Checkpoint round trip (inference/inference.py:179-310, util/serialization.py): the reference's zip layout -- version.json,
graphs.json, mxnet_parameters.npz keyed by UUID (unconstrained values), mxnet_constants.npz, variable_constants.json,
configuration.json -- and reconciliation of a freshly built model's UUIDs with the saved ones.
The CPU test needs no kernel call (initialise / set / save / load / compare); the GPU test is the reference's
testing/inference/inference_serialization_test.py:175-220 program plus a prediction from the reloaded parameters."""
import io
import json
import os
import zipfile

import numpy as np
import pytest
import torch

DT = 'float64'


def _model(dev, noise_var, lengthscale, variance):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 3))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=t(noise_var))
    kernel = RBF(input_dim=3, ARD=True, variance=t(variance), lengthscale=t(lengthscale), dtype=DT)
    m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, 1), dtype=DT)
    return m


def test_zip_layout_and_uuid_reconciliation_cpu(tmp_path):
    from mxfusion_amd.inference import Inference, MAP
    from mxfusion_amd.common.exceptions import SerializationError
    rng = np.random.RandomState(0)
    nv, ls, var = rng.rand(1), rng.rand(3), rng.rand(1)
    m = _model('cpu', nv, ls, var)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT, context=torch.device('cpu'))
    infr.initialize(X=(10, 3), Y=(10, 1))
    infr.params[m.noise_var] = torch.tensor([0.37], dtype=torch.float64)
    infr.params[m.Y.factor.kernel.lengthscale] = torch.tensor([0.5, 0.6, 0.7], dtype=torch.float64)
    z = str(tmp_path / 'inference.zip')
    infr.save(z)
    with zipfile.ZipFile(z) as zf:
        assert sorted(zf.namelist()) == sorted(['graphs.json', 'mxnet_parameters.npz', 'mxnet_constants.npz', 'variable_constants.json',
                                                'configuration.json', 'version.json'])
        assert json.loads(zf.read('version.json').decode()) == {'serialization_version': '2.0'}
        assert json.loads(zf.read('configuration.json').decode()) == {'observed': [m.X.uuid, m.Y.uuid]}
        npz = np.load(io.BytesIO(zf.read('mxnet_parameters.npz')))
        assert m.noise_var.uuid in npz.files                     # keyed by UUID, unconstrained (softplus^-1) value
        assert np.allclose(np.log1p(np.exp(npz[m.noise_var.uuid])), 0.37)
        vc = json.loads(zf.read('variable_constants.json').decode())
        assert vc[m.N.uuid] == 10
    m2 = _model('cpu', rng.rand(1), rng.rand(3), rng.rand(1))         # same script, fresh UUIDs, different initial values
    infr2 = Inference(MAP(model=m2, observed=[m2.X, m2.Y]), dtype=DT, context=torch.device('cpu'))
    with pytest.raises(SerializationError):
        infr2.load(z)                                             # must be initialised first
    infr2.initialize(X=(10, 3), Y=(10, 1))
    infr2.load(z)
    assert infr2._uuid_map[m.noise_var.uuid] == m2.noise_var.uuid and infr2._uuid_map[m.X.uuid] == m2.X.uuid
    for u in infr.params._slices:
        assert torch.equal(infr.params.raw(u), infr2.params.raw(infr2._uuid_map[u]))
    sp = lambda r: np.log1p(np.exp(r.numpy()))                   # constrained value = softplus(stored); params[...] itself is a HIP op
    assert abs(float(sp(infr2.params.raw(m2.noise_var)).reshape(-1)[0]) - 0.37) < 1e-12
    assert np.allclose(sp(infr2.params.raw(m2.Y.factor.kernel.lengthscale)), [0.5, 0.6, 0.7])
    # graphs.json has the REFERENCE's layout (factor_graph.py:619-628): a list of networkx node-link dicts, components as serialization.py:42-53
    with zipfile.ZipFile(z) as zf:
        graphs = json.loads(zf.read('graphs.json').decode())
    assert isinstance(graphs, list) and set(graphs[0]) >= {'directed', 'multigraph', 'nodes', 'links', 'name'}
    node = {n['id']['name']: n['id'] for n in graphs[0]['nodes']}
    assert node['noise_var']['uuid'] == m.noise_var.uuid and node['noise_var']['type'] == 'Variable' and node['noise_var']['version'] == '1.0'
    assert node['X']['attributes'] == [m.N.uuid]
    mod = [n['id'] for n in graphs[0]['nodes'] if n['id']['type'] == 'GPRegression'][0]
    assert [g['name'] for g in mod['graphs']] == ['gp_regression', 'gp_regression_posterior']
    assert {(l['source']['name'], l['name'], l['target']['type']) for l in graphs[0]['links']} >= {('X', 'X', 'GPRegression'), ('noise_var', 'noise_var', 'GPRegression')}
    # a structurally different model is refused: a named component of the saved graph that the running script does not have
    for n in graphs[0]['nodes']:
        if n['id']['name'] == 'noise_var':
            n['id']['name'] = 'some_other_name'
    z2 = str(tmp_path / 'broken.zip')
    with zipfile.ZipFile(z) as zin, zipfile.ZipFile(z2, 'w') as zout:
        for n in zin.namelist():
            zout.writestr(n, json.dumps(graphs) if n == 'graphs.json' else zin.read(n))
    with pytest.raises(SerializationError):
        infr2.load(z2)


def test_corrupt_or_incomplete_checkpoint_is_an_error_cpu(tmp_path):
    """A truncated parameter archive or one that lacks a trainable parameter must raise, not leave the model at its initial values."""
    from mxfusion_amd.inference import Inference, MAP
    from mxfusion_amd.common.exceptions import SerializationError
    rng = np.random.RandomState(1)
    m = _model('cpu', rng.rand(1), rng.rand(3), rng.rand(1))
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT, context=torch.device('cpu'))
    infr.initialize(X=(10, 3), Y=(10, 1))
    z = str(tmp_path / 'inference.zip')
    infr.save(z)

    def rewrite(edit):
        out = str(tmp_path / 'edited.zip')
        with zipfile.ZipFile(z) as src, zipfile.ZipFile(out, 'w') as dst:
            for name in src.namelist():
                data = edit(name, src.read(name))
                if data is not None:
                    dst.writestr(name, data)
        return out
    m2 = _model('cpu', rng.rand(1), rng.rand(3), rng.rand(1))
    infr2 = Inference(MAP(model=m2, observed=[m2.X, m2.Y]), dtype=DT, context=torch.device('cpu'))
    infr2.initialize(X=(10, 3), Y=(10, 1))
    with pytest.raises(SerializationError):          # truncated mxnet_parameters.npz
        infr2.load(rewrite(lambda n, d: d[:len(d) // 2] if n == 'mxnet_parameters.npz' else d))

    def drop_one(n, d):                               # a valid archive that lacks one trainable parameter
        if n != 'mxnet_parameters.npz':
            return d
        npz = np.load(io.BytesIO(d))
        keep = {k: npz[k] for k in npz.files if k != m.noise_var.uuid}
        b = io.BytesIO()
        np.savez(b, **keep)
        return b.getvalue()
    before = infr2.params._flat.detach().clone()
    with pytest.raises(SerializationError):
        infr2.load(rewrite(drop_one))
    assert torch.equal(infr2.params._flat.detach(), before)      # validated before the first write: nothing was half-restored
    infr2.load(z)                                     # the intact checkpoint still loads


@pytest.mark.gpu
def test_gp_module_save_and_load_gpu(tmp_path):
    from mxfusion_amd.inference import Inference, MAP, TransferInference, ModulePredictionAlgorithm
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()
    np.random.seed(0)
    X, Xt, Y = np.random.rand(10, 3), np.random.rand(20, 3), np.random.rand(10, 1)
    nv, ls, var = np.random.rand(1), np.random.rand(3), np.random.rand(1)
    m = _model('cuda', nv, ls, var)
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
    loss, _ = infr.run(X=t(X), Y=t(Y))
    z = str(tmp_path / 'inference.zip')
    infr.save(z)
    m2 = _model('cuda', nv * 3, ls * 2, var * 5)
    infr2 = Inference(MAP(model=m2, observed=[m2.X, m2.Y]), dtype=DT)
    infr2.initialize(X=t(X), Y=t(Y))
    infr2.load(z)
    for u in infr.params._slices:
        assert torch.equal(infr.params.raw(u), infr2.params.raw(infr2._uuid_map[u]))
    # the posterior caches (L, LinvY, X of gp_regression.py:72-75) travel too: predict from the reloaded inference without re-running it
    pred = lambda mm, ii: TransferInference(ModulePredictionAlgorithm(mm, observed=[mm.X], target_variables=[mm.Y]), infr_params=ii.params,
                                            dtype=DT).run(X=t(Xt))[0]
    a, b = pred(m, infr), pred(m2, infr2)
    assert torch.allclose(a[0], b[0], rtol=0, atol=1e-12) and torch.allclose(a[1], b[1], rtol=0, atol=1e-12)
    loss2, _ = infr2.run(X=t(X), Y=t(Y))
    assert abs(float(loss) - float(loss2)) < 1e-10


def test_checkpoint_in_the_reference_format_loads_cpu(golden_dir):
    """A zip laid out as the reference's Inference.save writes it (tests/golden/reference_format_gp.zip, assembled by hand from the
    reference's source by tests/golden/make_reference_checkpoint.py: networkx node-link graphs.json incl. the module's internal graphs
    with their factors, every uuid foreign to this process) loads: the reference's reconciliation (names, then predecessor edges, modules
    recursively; factor_graph.py:479-588) pairs every parameter -- the named model variables, the UNNAMED kernel parameters inside the
    module (through F <- GaussianProcess <- 'rbf_lengthscale' / 'rbf_variance') and the posterior's L / LinvY / X."""
    from mxfusion_amd.inference import Inference, MAP
    exp = np.load(os.path.join(golden_dir, 'reference_format_gp_expected.npz'))
    rng = np.random.RandomState(5)
    m = _model('cpu', rng.rand(1), rng.rand(3), rng.rand(1))
    infr = Inference(MAP(model=m, observed=[m.X, m.Y]), dtype=DT, context=torch.device('cpu'))
    infr.initialize(X=(10, 3), Y=(10, 1))
    infr.load(os.path.join(golden_dir, 'reference_format_gp.zip'))
    mp = infr._uuid_map
    gp = m.Y.factor
    assert mp['ref_noise_var'] == m.noise_var.uuid and mp['ref_X'] == m.X.uuid and mp['ref_Y'] == m.Y.uuid and mp['ref_N'] == m.N.uuid
    assert mp['ref_lengthscale'] == gp.kernel.lengthscale.uuid and mp['ref_variance'] == gp.kernel.variance.uuid
    post = gp._extra_graphs[0]
    assert mp['ref_post_L'] == post.L.uuid and mp['ref_post_LinvY'] == post.LinvY.uuid and mp['ref_post_X'] == post.X.uuid
    sp = lambda r: np.log1p(np.exp(r.numpy()))
    assert np.allclose(sp(infr.params.raw(m.noise_var)).ravel(), exp['noise_var'])
    assert np.allclose(sp(infr.params.raw(gp.kernel.lengthscale)).ravel(), exp['lengthscale'])
    assert np.allclose(sp(infr.params.raw(gp.kernel.variance)).ravel(), exp['variance'])
    assert np.allclose(infr.params.raw(post.L).numpy().reshape(10, 10), exp['L'])
    assert np.allclose(infr.params.raw(post.X).numpy().reshape(10, 3), exp['X'])
    assert infr.params.constants[m.N.uuid] == 10


def test_svgp_with_meanfield_posterior_graphs_reconcile_cpu():
    """graphs.json of the bench's model class (latent inputs, SVGPRegression with a combination kernel, mean-field q(X)) written in the
    reference's layout and reconciled against a second, independently built copy of the same script: every parameter-carrying variable --
    named model variables, the kernel's parameters inside the module, q(u) in the module's posterior graph, the unnamed mean / variance of
    the mean-field factor (reached over the 'random_variable' <- Normal <- 'mean' / 'variance' edges) -- is paired with its counterpart."""
    from mxfusion_amd.util import graph_json as gj

    def build():
        from mxfusion_amd import Model, Variable
        from mxfusion_amd.components.variables import PositiveTransformation
        from mxfusion_amd.components.distributions import Normal
        from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
        from mxfusion_amd.modules.gp_modules import SVGPRegression
        from mxfusion_amd.inference import create_Gaussian_meanfield
        m = Model()
        m.N = Variable()
        m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, 3))
        m.Z = Variable(shape=(5, 3))
        m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
        m.Y = SVGPRegression.define_variable(X=m.X, kernel=Matern52(3, ARD=True) + RBF(3, ARD=True), noise_var=m.noise_var, inducing_inputs=m.Z,
                                             shape=(m.N, 1))
        return m, create_Gaussian_meanfield(model=m, observed=[m.Y])

    def params(m, q):
        gp, qX = m.Y.factor, q[m.X].factor
        post = gp._extra_graphs[0]
        d = {'Z': m.Z, 'noise': m.noise_var, 'qm': post.qU_mean, 'qW': post.qU_cov_W, 'qd': post.qU_cov_diag, 'xm': qX.mean, 'xv': qX.variance}
        d.update(gp.kernel.parameters)
        return d
    (m1, q1), (m2, q2) = build(), build()
    js = json.loads(json.dumps([gj.graph_as_json(m1), gj.graph_as_json(q1)]))
    mod = [n['id'] for n in js[0]['nodes'] if n['id']['type'] == 'SVGPRegression'][0]
    inner = {l['name'] for l in mod['graphs'][0]['links']}
    assert {'add_matern52_lengthscale', 'add_rbf_variance', 'random_variable'} <= inner      # kernel parameters feed the internal GP factor, unnamed
    saved = gj.load_graphs(js)
    cmap = gj.reconcile_graphs([m2, q2], saved[0], saved[1:])
    p1, p2 = params(m1, q1), params(m2, q2)
    assert len(p1) == 11
    for n in p1:
        assert cmap.get(p1[n].uuid) == p2[n].uuid, n
