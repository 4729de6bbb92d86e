"""mxf_svgp_logpdf_sampled: the SVGP bound with SAMPLED operands -- hyper-parameters, noise, inducing inputs, q(u) -- in ONE C-ABI call
(VERDICT r02 item 7).  The reference gives every runtime array a sample axis and broadcasts them all to S
(components/variables/runtime_variable.py:96-118); its test pattern testing/modules/svgpregression_test.py:142-167 feeds the module a
noise variable with its own leading axis.  Values and per-operand gradients against the oracle (float64 1e-9; float32 at the composites'
usual tolerance), for all operands sampled and for a mix of sampled and shared ones, through the C ABI and through the module API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

NAMES = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')


def _problem(S, sampled, seed=0, B=300, M=24, Q=3, P=2):
    rng = np.random.RandomState(seed)
    n = lambda name: S if name in sampled else 1
    a = {'X': rng.uniform(-2, 2, (n('X'), B, Q)), 'Y': rng.randn(n('Y'), B, P), 'Z': rng.uniform(-2, 2, (n('Z'), M, Q)),
         'noise': rng.rand(n('noise'), 1) * 0.3 + 0.05, 'qm': rng.randn(n('qm'), M, P) * 0.3, 'qW': rng.randn(n('qW'), M, M) * 0.1,
         'qd': rng.rand(n('qd'), M) + 0.5, 'ls': rng.rand(n('ls'), Q) + 0.8, 'var': rng.rand(n('var'), 1) + 0.5}
    return a


def _oracle(a, S, scaling):
    lv = {k: O.T(v).clone().requires_grad_(True) for k, v in a.items()}
    logL = O.svgp_log_pdf(O.RBF(a['X'].shape[-1], ARD=True), lv['X'], lv['Y'], lv['Z'], lv['noise'], lv['qm'], lv['qW'], lv['qd'],
                          {'rbf_lengthscale': lv['ls'], 'rbf_variance': lv['var']}, jitter=1e-6, log_pdf_scaling=scaling)
    assert logL.shape == (S,)
    logL.mean().backward()
    return logL.detach().numpy(), {k: v.grad.numpy() for k, v in lv.items()}


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 2e-4)])
@pytest.mark.parametrize('sampled', [NAMES, ('noise', 'ls', 'var'), ('X', 'Z', 'qm'), ('qW', 'qd', 'Y')])
def test_sampled_operands_in_one_call_vs_oracle(dtype, tol, sampled):
    from mxfusion_amd import ops
    S = 3
    a = _problem(S, sampled)
    if dtype == torch.float32:
        a['ls'] = a['ls'] * 0.5                       # the float32 composites are compared with the oracle on a well-conditioned Kuu
    ref, g = _oracle(a, S, 1.7)
    d = {k: torch.as_tensor(v, dtype=dtype).cuda() for k, v in a.items()}
    r = ops.svgp_logpdf_sampled('rbf', d['X'], d['Y'], d['Z'], d['noise'], d['qm'], d['qW'], d['qd'], d['ls'], d['var'], True, jitter=1e-6, scaling=1.7,
                                gscale=1.0 / S, want_grad=True)
    torch.cuda.synchronize()
    assert r['info'].shape == (S,) and int(r['info'].abs().sum()) == 0
    assert np.allclose(r['logL'].double().cpu().numpy(), ref, rtol=max(tol, 1e-5) if dtype == torch.float32 else tol, atol=0)
    for key, name in zip(('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar'), NAMES):
        got = r[key].double().cpu().numpy().reshape((S,) + a[name].shape[1:])
        if a[name].shape[0] == 1:
            got = got.sum(0, keepdims=True)          # a shared operand's gradient is the sum of its per-sample slices
        ref_g = g[name]
        assert np.linalg.norm(got - ref_g) <= tol * max(np.linalg.norm(ref_g), 1e-30), (name, np.linalg.norm(got - ref_g) / np.linalg.norm(ref_g))


def test_module_with_sampled_noise_and_kernel_parameters_is_one_call(monkeypatch):
    """Through SVGPRegressionLogPdf.compute with runtime arrays that carry a sample axis on the noise variance and the kernel parameters
    (X, Y, Z, q(u) shared): one mxf_svgp_logpdf_sampled call, value and gradients of every operand equal to the oracle's."""
    from mxfusion_amd import ops
    from mxfusion_amd.modules.gp_modules import _fused
    calls = []
    real = ops.svgp_logpdf_sampled
    monkeypatch.setattr(ops, 'svgp_logpdf_sampled', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    S = 4
    a = _problem(S, ('noise', 'ls', 'var'), seed=3, P=1)
    ref, g = _oracle(a, S, 1.0)
    t = {k: torch.as_tensor(v, dtype=torch.float64).cuda().requires_grad_(True) for k, v in a.items()}
    logL, info = _fused.SVGPSampledLogPdfFn.apply(None, 'rbf', True, 1e-6, 1.0, t['X'], t['Y'], t['Z'], t['noise'], t['qm'], t['qW'], t['qd'], t['ls'], t['var'])
    logL.mean().backward()
    torch.cuda.synchronize()
    assert len(calls) == 1
    assert np.allclose(logL.detach().cpu().numpy(), ref, rtol=1e-9, atol=0)
    for name in NAMES:
        got = t[name].grad.cpu().numpy()
        assert got.shape == a[name].shape
        assert np.linalg.norm(got - g[name]) <= 1e-9 * np.linalg.norm(g[name]), name
