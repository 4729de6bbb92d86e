"""The float32 guard of the SVGP training call: above cond_1(Kuu + jitter I) ~ 3e3 the float32 streaming form cannot
hold north_star's 1e-5 on the ELBO (its error grows like cond 2^-24: 2e-3 at 5e4), so the module switches its streaming stage to float64 BY
ITSELF -- first call checked synchronously, later calls through the condition number every finished call publishes into pinned host memory
(mxf_svgp_cond_nowait, no synchronisation).  Checked against the ORACLE at length-scale 3 (cond ~ 5e4), B = 8192, M = 1024."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _inputs(ell, seed=0):
    rng = np.random.default_rng(seed)
    B, Q, M = 8192, 8, 1024
    X = rng.uniform(-3., 3., (1, B, Q))
    Y = np.sin(X[0] @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((B, 1))
    Z = rng.uniform(-3., 3., (M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, 1)), 0.4 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    return dict(X=X, Y=Y[None], Z=Z[None], noise=np.array([[0.02]]), qm=qm[None], qW=qW[None], qd=qd[None], ls=np.full((1, Q), ell), var=np.array([[1.0]]))


def _oracle(a):
    T = O.T
    return float(O.svgp_log_pdf(O.RBF(8, ARD=True), T(a['X']), T(a['Y']), T(a['Z']), T(a['noise']), T(a['qm']), T(a['qW']), T(a['qd']),
                                {'rbf_lengthscale': T(a['ls']), 'rbf_variance': T(a['var'])}, jitter=1e-6)[0])


def _module_call(a):
    """The module's bridge (SVGPRegressionLogPdf.compute -> SVGPLogPdfFn) in float32 with the reverse mode requested, as a training step calls it."""
    from mxfusion_amd.modules.gp_modules._fused import SVGPLogPdfFn
    t = {k: torch.as_tensor(v, dtype=torch.float32).cuda().requires_grad_(k not in ('Y',)) for k, v in a.items()}
    logL, info = SVGPLogPdfFn.apply('rbf', True, 1e-6, 1.0, t['X'], t['Y'], t['Z'], t['noise'], t['qm'], t['qW'], t['qd'], t['ls'], t['var'])
    logL.sum().backward()
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0 and logL.dtype == torch.float32 and t['Z'].grad.dtype == torch.float32
    return float(logL[0])


def test_ill_conditioned_start_is_never_evaluated_in_float32():
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    from mxfusion_amd import ops
    a = _inputs(3.0)
    ref = _oracle(a)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = _module_call(a)
    assert Float32Guard.active and any('float64' in str(x.message) for x in w)
    assert ops.svgp_last_cond() > Float32Guard.LIMIT
    assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)                 # north_star's bar, at cond ~ 5e4, through the float32 API
    # the plain float32 call (guard bypassed) is what the guard protects from: it misses the bar there
    Float32Guard.reset()
    Float32Guard.enabled = False
    try:
        raw = _module_call(a)
    finally:
        Float32Guard.enabled = True
    assert abs(raw - ref) > 1e-5 * abs(ref)


def test_guard_trips_without_synchronising_when_training_leaves_the_float32_range():
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    from mxfusion_amd import ops
    good, bad = _inputs(1.0), _inputs(3.0)
    ref_good, ref_bad = _oracle(good), _oracle(bad)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        g = _module_call(good)                                           # first call: checked synchronously, well conditioned
        assert not Float32Guard.active and abs(g - ref_good) <= 1e-5 * abs(ref_good)
        assert 0 < ops.svgp_cond_nowait() < Float32Guard.LIMIT
        _module_call(bad)                                                # the parameters have drifted: this call still runs in float32 ...
        assert ops.svgp_cond_nowait() > Float32Guard.LIMIT               # ... and publishes its condition number
        b = _module_call(bad)                                            # the next one sees it (no host sync needed) and runs in float64
    assert Float32Guard.active
    assert abs(b - ref_bad) <= 1e-5 * abs(ref_bad), (b, ref_bad)
