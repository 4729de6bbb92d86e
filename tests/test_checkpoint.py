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
    # a structurally different model is refused
    from mxfusion_amd import Model, Variable
    m3 = Model()
    m3.a = Variable(shape=(2,))
    with open(z, 'rb') as f:
        raw = f.read()
    buf = io.BytesIO(raw)
    with zipfile.ZipFile(buf) as zf:
        graphs = json.loads(zf.read('graphs.json').decode())
    graphs[0]['variables'] = graphs[0]['variables'][:-1]
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
    with pytest.raises(SerializationError):
        infr2.load(rewrite(drop_one))
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
