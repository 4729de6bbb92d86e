"""GPU parity of the fused composites (C ABI) vs the CPU oracle: exact-GP log marginal and SVGP bound,
values and reverse mode, on the golden fixtures (reference tests' inputs) and on larger seeded problems."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

KINDS = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


_REPORT = {}
# float32 gradients against the float64 oracle, normwise (|error| <= tol * (|ref| + max|ref|)): the worst case over this file is 6e-4
# (MXF_TEST_REPORT=1 prints the worst error / bound ratio per quantity); it was 5e-3 in round 1
F32_GTOL = 1.5e-3


def _close(got, ref, rtol, name='', floor=None):
    """Element-wise: |got - ref| <= rtol * (|ref| + floor * max|ref|).  floor = 1e-3 for the float64 comparisons (rtol < 1e-6): an entry
    a thousand times smaller than the largest one is still checked to rtol relative to ITSELF plus a small absolute allowance -- the
    earlier normwise form (atol = rtol * max|ref|) left small-magnitude gradient entries effectively unchecked.  The float32 streaming
    path keeps floor = 1 (normwise): its f32 accumulations carry errors relative to the LARGEST partial sum, not to the entry."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if not ref.size:
        return
    if floor is None:
        floor = 1e-3 if rtol < 1e-6 else 1.0
    scale = max(1.0 if floor >= 1.0 else 0.0, float(np.abs(ref).max()))
    bound = rtol * (np.abs(ref) * (0.0 if floor >= 1.0 else 1.0) + floor * scale) if floor < 1.0 else rtol * (np.abs(ref) + scale)
    ratio = float((np.abs(got - ref) / np.maximum(bound, 1e-300)).max())
    if os.environ.get('MXF_TEST_REPORT'):
        key = (name, rtol)
        _REPORT[key] = max(_REPORT.get(key, 0.0), ratio)
    assert ratio <= 1.0, (name, float(np.abs(got - ref).max()), scale, ratio)


if os.environ.get('MXF_TEST_REPORT'):
    import atexit

    @atexit.register
    def _print_report():
        for (name, rtol), r in sorted(_REPORT.items(), key=lambda kv: -kv[1]):
            print('CLOSE-REPORT %-18s rtol %.1e  worst error / bound = %.2e' % (name, rtol, r))


def test_gp_logpdf_golden(golden_dir):
    """KAT-GP: testing/modules/gpregression_test.py:40-48 inputs, value asserted by the reference vs GPy (:96)."""
    from mxfusion_amd import ops
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    r = ops.gp_logpdf('rbf', _dev(g['X'][None]), _dev(g['Y'][None]), _dev(g['noise'][None]), _dev(g['ls'][None]),
                      _dev(g['var'][None]), True, want_grad=True)
    _close(r['logL'], g['logL'], 1e-12, 'logL')
    assert abs(float(r['logL'][0]) - (-18.814420362103)) < 1e-9
    _close(r['L'][0], g['L'], 1e-11, 'L')
    _close(r['LinvY'][0], g['LinvY'], 1e-11, 'LinvY')
    for n, k in (('dX', 'd_X'), ('dY', 'd_Y'), ('dnoise', 'd_noise'), ('dls', 'd_ls'), ('dvar', 'd_var')):
        _close(r[n][0].reshape(g[k].shape), g[k], 1e-9, n)


@pytest.mark.parametrize('kind', ['rbf', 'matern52', 'matern32'])
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])     # float32: worst observed 6e-8 (value), 6e-5 (gradients)
@pytest.mark.parametrize('N,Q,P,S', [(200, 4, 2, 1), (700, 8, 1, 2), (2240, 5, 2, 1)])     # N >= 2048: L^-1 Y through the explicit inverse
def test_gp_logpdf_vs_oracle(kind, dtype, tol, N, Q, P, S):
    if N > 2000 and kind not in ('rbf', 'matern32'):
        pytest.skip('large case: two kernels are enough')
    from mxfusion_amd import ops
    rng = np.random.RandomState(N)
    X = rng.uniform(-3, 3, (S, N, Q))
    Y = rng.randn(1, N, P)
    ls = rng.rand(1, Q) + 1.0
    var = rng.rand(1, 1) + 0.5
    noise = np.array([[0.3]])
    k = KINDS[kind](Q, ARD=True)
    leaves = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=X, Y=Y, ls=ls, var=var, noise=noise).items()}
    logL = O.gp_log_pdf(k, leaves['X'], leaves['Y'], leaves['noise'], {k.name + '_lengthscale': leaves['ls'], k.name + '_variance': leaves['var']}, jitter=1e-6)
    r = ops.gp_logpdf(kind, _dev(X, dtype), _dev(Y, dtype), _dev(noise, dtype), _dev(ls, dtype), _dev(var, dtype), True, jitter=1e-6, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')
    for s in range(S):
        grads = torch.autograd.grad(logL[s], [leaves[n] for n in ('X', 'Y', 'noise', 'ls', 'var')], retain_graph=True)
        _close(r['dX'][s], grads[0][s], tol * 30, 'dX')
        _close(r['dY'][s], grads[1][0], tol * 30, 'dY')
        _close(r['dnoise'][s], grads[2][0], tol * 30, 'dnoise')
        _close(r['dls'][s], grads[3][0], tol * 30, 'dls')
        _close(r['dvar'][s], grads[4][0], tol * 30, 'dvar')


def test_svgp_logpdf_golden(golden_dir):
    """KAT-SVGP: testing/modules/svgpregression_test.py:41-56,68 inputs; reference asserts the value vs GPy (:115)."""
    from mxfusion_amd import ops
    g = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    a = [_dev(g[n]) for n in ('Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')]
    r = ops.svgp_logpdf('rbf', _dev(g['X'][None]), _dev(g['Y'][None]), *a, True, jitter=1e-8)
    assert abs(float(r['logL'][0]) - (-32.725635407458)) < 1e-8
    _close(r['logL'], g['logL'], 1e-11, 'logL')
    r = ops.svgp_logpdf('rbf', _dev(g['X'][None]), _dev(g['Y'][None]), *a, True, jitter=1e-8, scaling=3.5, want_grad=True)
    _close(r['logL'], g['logL_scaled'], 1e-11, 'logL_scaled')
    for n, k in (('dX', 'd_X'), ('dY', 'd_Y'), ('dZ', 'd_Z'), ('dnoise', 'd_noise'), ('dmu', 'd_qm'), ('dW', 'd_qW'),
                 ('dSdiag', 'd_qd'), ('dls', 'd_ls'), ('dvar', 'd_var')):
        _close(r[n].reshape(g[k].shape), g[k], 1e-8, n)


@pytest.mark.parametrize('kind', ['rbf', 'matern52', 'matern12'])
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('B,M,Q,P,S', [(300, 20, 3, 1, 1), (1000, 130, 8, 2, 3),
                                       # float32, one output column, B % 16 == 0: the matrix-pipe reverse pass with ragged row bands (M % 128 != 0),
                                       # ragged column tiles (S B % 64 != 0) and padded coordinates (Q < 8)
                                       (208, 20, 3, 1, 2), (1040, 130, 8, 1, 3), (48, 200, 5, 1, 1)])
def test_svgp_logpdf_vs_oracle(kind, dtype, tol, B, M, Q, P, S):
    """sampled inputs X (S,B,Q) (the latent-input model of svgpregression_test.py:357-385), shared Y."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(B + M)
    X = rng.uniform(-2, 2, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    qm = rng.randn(M, P) * 0.3
    qW = rng.randn(M, M) * 0.1
    qd = rng.rand(M) + 0.5
    # float32 streams Kuf through the explicit-inverse form, whose rounding error scales with cond(Kuu) * 6e-8
    # (DESIGN.md "precision"): exercise it on a well-conditioned Kuu (short lengthscale); float64 on the hard one
    ls = rng.rand(Q) * 0.5 + (1.0 if dtype == torch.float64 else 0.25)
    var = np.array([1.3])
    noise = np.array([0.05])
    k = KINDS[kind](Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'][None], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6, log_pdf_scaling=2.0)
    obj = logL.mean()
    grads = torch.autograd.grad(obj, [lv[n] for n in names])
    r = ops.svgp_logpdf(kind, _dev(X, dtype), _dev(Y[None], dtype), _dev(Z, dtype), _dev(noise, dtype), _dev(qm, dtype), _dev(qW, dtype),
                        _dev(qd, dtype), _dev(ls, dtype), _dev(var, dtype), True, jitter=1e-6, scaling=2.0, gscale=1.0 / S, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')          # north_star: 1e-5 relative on the ELBO
    gtol = tol * 50 if dtype == torch.float64 else F32_GTOL
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        _close(r[key].reshape(grads[names.index(n)].shape), grads[names.index(n)], gtol, key)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('nshape', ['BP', 'B1', '1P'])
@pytest.mark.parametrize('B,M,Q,P,S', [(10, 3, 3, 2, 1), (700, 70, 5, 3, 2)])
def test_svgp_logpdf_heteroscedastic_vs_oracle(dtype, tol, nshape, B, M, Q, P, S):
    """noise_var of shape (N, D) / (N, 1) / (D,): svgp_regression.py:61-67, exercised by the reference in
    testing/modules/svgpregression_test.py:142-167 (test_log_pdf_w_samples_of_noise_var)."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(B + M + len(nshape))
    X = rng.uniform(-2, 2, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    ls = rng.rand(Q) * 0.5 + (1.0 if dtype == torch.float64 else 0.25)
    var = np.array([1.3])
    noise = rng.rand(*{'BP': (B, P), 'B1': (B, 1), '1P': (1, P)}[nshape]) * 0.2 + 0.03
    k = O.RBF(Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    o_noise = lv['noise'][None] if nshape != '1P' else lv['noise']          # (1,N,D') heteroscedastic; (1,D) per-output
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'][None], lv['Z'][None], o_noise, lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6, log_pdf_scaling=1.5)
    grads = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    r = ops.svgp_logpdf('rbf', _dev(X, dtype), _dev(Y[None], dtype), _dev(Z, dtype), _dev(noise, dtype), _dev(qm, dtype), _dev(qW, dtype),
                        _dev(qd, dtype), _dev(ls, dtype), _dev(var, dtype), True, jitter=1e-6, scaling=1.5, gscale=1.0 / S, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')
    gtol = tol * 50 if dtype == torch.float64 else F32_GTOL
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        _close(r[key].reshape(grads[names.index(n)].shape), grads[names.index(n)], gtol, key)


class _MatKernel(object):
    """oracle-side stand-in kernel whose K / Kdiag return fixed leaf tensors, so that autograd yields d/dKuu, d/dKuf, d/dKdiag."""
    name = 'mat'

    def __init__(self, Kuu, Kuf, Kdiag):
        self.Kuu, self.Kuf, self.Kdiag_ = Kuu, Kuf, Kdiag

    def K(self, X, X2=None, **kw):
        return self.Kuu[None] if X2 is None else self.Kuf[None]

    def Kdiag(self, X, **kw):
        return self.Kdiag_[None]


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('nshape', ['11', 'BP'])
def test_svgp_logpdf_mat_vs_oracle(dtype, tol, nshape):
    """mxf_svgp_logpdf_mat: the bound and its reverse mode w.r.t. materialised Kuu / Kuf / Kdiag (any PSD Kuu, any Kuf)."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(5)
    B, M, P = 400, 40, 2
    A = rng.randn(M, M)
    Kuu = A @ A.T / M + np.eye(M) * (0.5 if dtype == torch.float64 else 2.0)
    Kuf = rng.randn(M, B) * 0.3
    Kdiag = rng.rand(B) + 1.0
    Y = rng.randn(B, P)
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    noise = rng.rand(*{'11': (1,), 'BP': (B, P)}[nshape]) * 0.2 + 0.05
    names = ('Kuu', 'Kuf', 'Kdiag', 'Y', 'noise', 'qm', 'qW', 'qd')
    vals = dict(Kuu=Kuu, Kuf=Kuf, Kdiag=Kdiag, Y=Y, noise=noise, qm=qm, qW=qW, qd=qd)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    k = _MatKernel(lv['Kuu'], lv['Kuf'], lv['Kdiag'])
    logL = O.svgp_log_pdf(k, torch.zeros(1, B, 1, dtype=torch.float64), lv['Y'][None], torch.zeros(1, M, 1, dtype=torch.float64),
                          lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None], {'mat_unused': torch.zeros(1, 1, dtype=torch.float64)},
                          jitter=1e-6, log_pdf_scaling=3.0)
    grads = torch.autograd.grad(logL.sum(), [lv[n] for n in names])
    r = ops.svgp_logpdf_mat(*[_dev(vals[n], dtype) for n in names], jitter=1e-6, scaling=3.0, gscale=1.0, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')
    gtol = tol * 50 if dtype == torch.float64 else F32_GTOL
    for n, key in zip(names, ('dKuu', 'dKuf', 'dKdiag', 'dY', 'dnoise', 'dmu', 'dW', 'dSdiag')):
        got, ref = r[key], grads[names.index(n)]
        if key == 'dKuu':      # autograd of the oracle gives the gradient w.r.t. an unconstrained (non-symmetric) Kuu: compare symmetrised
            got, ref = 0.5 * (got + got.T), 0.5 * (ref + ref.T)
        _close(got.reshape(ref.shape), ref, gtol, key)


@pytest.mark.parametrize('kind', ['rbf', 'matern32'])
# (512, 128, .., 1, 2) / (1024, 256, 8, 1, 1): the matrix-pipe reverse pass with its f16 accumulation scaled from the max |T| word of the 128- / 256-row
# wide GEMM; (1040, 144, 5, 1, 1): the same pass on ragged tiles (M % 128 != 0, S B % 64 != 0) behind the narrow split GEMM, which reports no
# max |T|: the sigma^2 M max|H0| bound; P = 2 / Q = 16: the generic reverse pass
@pytest.mark.parametrize('B,M,Q,P,S', [(512, 128, 8, 1, 2), (2048, 256, 3, 2, 1), (1040, 144, 16, 1, 3), (1040, 144, 5, 1, 1), (1024, 256, 8, 1, 1)])
def test_svgp_logpdf_f32_split_path_vs_oracle(kind, B, M, Q, P, S):
    """The float32 training step as the bench runs it: Grams written as three-term bf16 planes, both big GEMMs on the bf16 matrix
    pipe (gemm_split.hip), w^T Kuf from the planes.  Taken when S*B and M are multiples of 16 and M >= 128 (every other f32 test in this
    file has ragged sizes and runs the plain f32-MFMA kernels).  ELBO to 1e-5 (north_star), gradients to the f32 tolerance."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(B + M + Q)
    X = rng.uniform(-2, 2, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.05, rng.rand(M) + 0.5
    ls = (rng.rand(Q) * 0.3 + 0.3) * np.sqrt(Q / 3.0)          # well-conditioned Kuu
    var, noise = np.array([1.3]), np.array([0.05])
    k = {'rbf': O.RBF, 'matern32': O.Matern32}[kind](Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'][None], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6, log_pdf_scaling=1.0)
    grads = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    dt = torch.float32
    r = ops.svgp_logpdf(kind, _dev(X, dt), _dev(Y[None], dt), _dev(Z, dt), _dev(noise, dt), _dev(qm, dt), _dev(qW, dt), _dev(qd, dt),
                        _dev(ls, dt), _dev(var, dt), True, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, 1e-5, 'logL')
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        _close(r[key].reshape(grads[names.index(n)].shape), grads[names.index(n)], F32_GTOL, key)
    # and the forward-only call (plain f32 kernels) agrees with the training call's value
    r0 = ops.svgp_logpdf(kind, _dev(X, dt), _dev(Y[None], dt), _dev(Z, dt), _dev(noise, dt), _dev(qm, dt), _dev(qW, dt), _dev(qd, dt),
                         _dev(ls, dt), _dev(var, dt), True, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=False)
    _close(r0['logL'], r['logL'].double().cpu(), 1e-4, 'fwd-only vs training value')     # the plain f32-MFMA kernels are the less accurate of the two


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('Q', [1, 2, 4, 7, 16])
@pytest.mark.parametrize('P', [1, 3, 8])
def test_svgp_logpdf_input_and_output_widths(dtype, tol, Q, P):
    """Every (Q tile, P tile) instantiation of the fused reverse pass and of the Gram kernels: Q in {1..16}, P in {1..8} (the merged
    LDS-atomic lane mapping differs per combination); B = 272, M = 128 takes the split path in float32."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(100 * Q + P)
    S, B, M = 2, 272, 128
    L = max(1.5, 0.3 * 128 ** (1.0 / Q))             # keep the 128 inducing points ~one length-scale apart (f32 needs a well-conditioned Kuu)
    X = rng.uniform(-L, L, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-L, L, (M, Q))
    if Q == 1:                                          # random points on a line come arbitrarily close: use a jittered lattice
        Z = (np.linspace(-L, L, M) + rng.uniform(-0.05, 0.05, M))[:, None]
    elif Q == 2:
        gx, gy = np.meshgrid(np.linspace(-L, L, 12), np.linspace(-L, L, 11))
        Z = np.stack([gx.ravel(), gy.ravel()], 1)[:M] + rng.uniform(-0.05, 0.05, (M, 2))
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.05, rng.rand(M) + 0.5
    ls = (rng.rand(Q) * 0.2 + 0.25) * max(1.0, np.sqrt(Q / 3.0))
    var, noise = np.array([1.1]), np.array([0.07])
    k = O.RBF(Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'][None], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-5)
    grads = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    r = ops.svgp_logpdf('rbf', _dev(X, dtype), _dev(Y[None], dtype), _dev(Z, dtype), _dev(noise, dtype), _dev(qm, dtype), _dev(qW, dtype),
                        _dev(qd, dtype), _dev(ls, dtype), _dev(var, dtype), True, jitter=1e-5, gscale=1.0 / S, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')
    gtol = tol * 1000 if dtype == torch.float64 else F32_GTOL
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        _close(r[key].reshape(grads[names.index(n)].shape), grads[names.index(n)], gtol, key)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-5)])
@pytest.mark.parametrize('B,M,Q,P,S', [(300, 20, 3, 2, 4), (1000, 130, 8, 1, 3)])
def test_svgp_logpdf_sampled_outputs_over_shared_inputs(dtype, tol, B, M, Q, P, S):
    """Y sampled (S,B,P), X shared (1,B,Q) -- the first layer of a deep GP (its output H is a sample of q(H), its input is data): the S
    samples share Kuf, T and U, one call instead of a per-sample loop; through the stationary-kernel entry point and through
    mxf_svgp_logpdf_mat."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(B + S)
    X = rng.uniform(-2, 2, (1, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P))[None] + 0.3 * rng.randn(S, B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    qm, qW, qd = rng.randn(M, P) * 0.3, rng.randn(M, M) * 0.1, rng.rand(M) + 0.5
    ls = rng.rand(Q) * 0.5 + (1.0 if dtype == torch.float64 else 0.25)
    var, noise = np.array([1.3]), np.array([0.05])
    k = O.RBF(Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    kp = {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None], kp,
                          jitter=1e-6, log_pdf_scaling=2.0)
    assert logL.shape == (S,)
    grads = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    r = ops.svgp_logpdf('rbf', _dev(X, dtype), _dev(Y, dtype), _dev(Z, dtype), _dev(noise, dtype), _dev(qm, dtype), _dev(qW, dtype),
                        _dev(qd, dtype), _dev(ls, dtype), _dev(var, dtype), True, jitter=1e-6, scaling=2.0, gscale=1.0 / S, want_grad=True)
    assert int(r['info'].abs().sum()) == 0
    _close(r['logL'], logL, tol, 'logL')
    gtol = tol * 50 if dtype == torch.float64 else F32_GTOL
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        _close(r[key].reshape(grads[names.index(n)].shape), grads[names.index(n)], gtol, key)
    # the same through the materialised-Gram entry point
    with torch.no_grad():
        Kuu = k.K(O.T(Z)[None], **{a: b.detach() for a, b in kp.items()})[0]
        Kuf = k.K(O.T(Z)[None], O.T(X), **{a: b.detach() for a, b in kp.items()})[0]
        Kd = k.Kdiag(O.T(X), **{a: b.detach() for a, b in kp.items()})[0]
    rm = ops.svgp_logpdf_mat(_dev(Kuu.numpy(), dtype), _dev(Kuf.numpy(), dtype), _dev(Kd.numpy(), dtype), _dev(Y, dtype), _dev(noise, dtype),
                             _dev(qm, dtype), _dev(qW, dtype), _dev(qd, dtype), jitter=1e-6, scaling=2.0, gscale=1.0 / S, want_grad=True)
    _close(rm['logL'], logL, tol, 'logL (mat)')
    for key in ('dY', 'dnoise', 'dmu', 'dW', 'dSdiag'):
        _close(rm[key].reshape(r[key].shape), r[key].double().cpu(), gtol, key + ' (mat)')


def test_svgp_split_path_training_call_is_graph_capturable():
    """The float32 training call on the split path (three HIP streams forked and joined by events) captured into a hipGraph and
    replayed gives the eager result.  Regression: a dependency between the two forked streams crashed hipStreamEndCapture."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(5)
    S, B, M, Q, P = 2, 512, 128, 8, 1
    X = rng.uniform(-2, 2, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P)
    Z = rng.uniform(-2, 2, (M, Q))
    dt = torch.float32
    args = [_dev(X, dt), _dev(Y[None], dt), _dev(Z, dt), _dev(np.array([0.05]), dt), _dev(rng.randn(M, P) * 0.3, dt), _dev(rng.randn(M, M) * 0.05, dt),
            _dev(rng.rand(M) + 0.5, dt), _dev((rng.rand(Q) * 0.3 + 0.3) * np.sqrt(Q / 3.0), dt), _dev(np.array([1.3]), dt), True]
    call = lambda: ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
    ref = call()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up off the default stream (scratch growth, lazy initialisation)
        call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = call()
    for _ in range(2):
        args[0].add_(0.0)                  # same inputs: the replay must reproduce the eager values
        g.replay()
    torch.cuda.synchronize()
    assert int(out['info'].abs().sum()) == 0
    for key in ('logL', 'dX', 'dZ', 'dW', 'dls', 'dmu', 'dnoise', 'dSdiag', 'dvar'):
        a, b = ref[key].double().cpu().numpy(), out[key].double().cpu().numpy()
        assert np.allclose(a, b, rtol=2e-4, atol=2e-5 * max(1e-30, np.abs(a).max())), key      # f32 atomics: summation order differs run to run


@pytest.mark.parametrize('var,noise,yscale', [(1e-3, 1e-4, 0.03), (60.0, 3.0, 8.0), (1.0, 0.05, 1.0)])
def test_svgp_split_path_is_insensitive_to_the_problem_scale(var, noise, yscale):
    """The f16x2 operand format of the float32 training step carries power-of-two scales (Gram planes hold k / variance * 2^14, H0's scale
    comes from its max-abs word): kernel variances / noise levels / output scales far from 1 must give the same agreement with the
    float64 step as the unit-scale problem."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(11)
    S, B, M, Q, P = 2, 1024, 128, 4, 1
    X = rng.uniform(-2, 2, (S, B, Q))
    Y = yscale * (np.sin(X[0] @ rng.randn(Q, P)) + 0.1 * rng.randn(B, P))
    Z = rng.uniform(-2, 2, (M, Q))
    qm, qW, qd = yscale * rng.randn(M, P) * 0.3, np.sqrt(var) * rng.randn(M, M) * 0.05, var * (rng.rand(M) + 0.5)
    ls = (rng.rand(Q) * 0.3 + 0.5)
    out = {}
    for dt in (torch.float64, torch.float32):
        out[dt] = ops.svgp_logpdf('rbf', _dev(X, dt), _dev(Y[None], dt), _dev(Z, dt), _dev(np.array([noise]), dt), _dev(qm, dt), _dev(qW, dt),
                                  _dev(qd, dt), _dev(ls, dt), _dev(np.array([var]), dt), True, jitter=1e-6 * var, scaling=1.0, gscale=1.0 / S,
                                  want_grad=True)
        assert int(out[dt]['info'].abs().sum()) == 0
    ref, got = out[torch.float64], out[torch.float32]
    assert abs(float(got['logL'].double().mean() - ref['logL'].mean())) <= 1e-5 * abs(float(ref['logL'].mean()))
    for key in ('dX', 'dZ', 'dW', 'dSdiag', 'dmu', 'dls', 'dvar', 'dnoise', 'dY'):
        a, b = ref[key].double().cpu().numpy(), got[key].double().cpu().numpy()
        assert np.linalg.norm(a - b) <= 5e-4 * np.linalg.norm(a), (key, np.linalg.norm(a - b) / np.linalg.norm(a))


def test_fused_bridges_reject_non_uniform_sample_weights():
    """modules/gp_modules/_fused.py: the fused composites return the gradients of gscale * sum_s logL[s] with ONE gscale; an upstream
    gradient that weights the samples differently cannot be represented and must not come back as a plausible-looking number."""
    from mxfusion_amd.modules.gp_modules._fused import SVGPLogPdfFn
    rng = np.random.RandomState(0)
    S, B, M, Q = 3, 40, 6, 2
    X = _dev(rng.rand(S, B, Q)).requires_grad_(True)
    Y, Z = _dev(rng.rand(1, B, 1)), _dev(rng.rand(1, M, Q)).requires_grad_(True)
    args = (_dev([[0.1]]), _dev(rng.randn(1, M, 1) * 0.1), _dev(rng.randn(1, M, M) * 0.05), _dev(rng.rand(1, M) + 0.5), _dev(np.ones((1, Q))),
            _dev([[1.0]]))
    logL, _ = SVGPLogPdfFn.apply(None, 'rbf', True, 1e-6, 1.0, X, Y, Z, *args)
    gX, gZ = torch.autograd.grad(logL.mean(), (X, Z), retain_graph=True)          # the reference's reduction: fine
    assert torch.isfinite(gX).all() and torch.isfinite(gZ).all()
    w = _dev([1.0, 2.0, 3.0])
    gX2, gZ2 = torch.autograd.grad((logL * w).sum(), (X, Z))                         # per-sample weights: poisoned, not silently wrong
    assert torch.isnan(gX2).all() and torch.isnan(gZ2).all()


@pytest.mark.parametrize('B,M,Q,S,kind,shared_y', [(512, 128, 5, 2, 'rbf', True), (768, 256, 8, 1, 'matern32', True), (512, 128, 3, 3, 'rbf', False),
                                                    (65536, 1024, 8, 1, 'rbf', True)])
def test_svgp_logpdf_heteroscedastic_streams_on_the_fused_path(B, M, Q, S, kind, shared_y):
    """Per-row noise (N, 1) with one output column in float32 (svgp_regression.py:61-67) runs the STREAMING split path since r04 (VERDICT r03
    item 8): planes of Kuf diag(sqrt(nmin beta)) and of diag(nmin beta) Kfu, the homoscedastic products and fused reverse pass with
    noise := nmin, one extra pass for sum beta e^2 and the per-row noise gradient.  Values and every gradient (the (N, 1) noise gradient
    included) against the oracle's autograd, up to the bench shape B = 65 536, M = 1 024; the in-library stage timer confirms that the
    planes / split-GEMM stages ran (the generic path has none)."""
    from mxfusion_amd import _lib, ops
    rng = np.random.RandomState(B % 1000 + M + Q)
    X = rng.uniform(-3, 3, (S, B, Q))
    Y = np.sin(X[0] @ rng.randn(Q, 1)) + 0.1 * rng.randn(B, 1)
    Ys = Y[None] if shared_y else np.stack([Y + 0.01 * rng.randn(B, 1) for _ in range(S)])
    Z = rng.uniform(-3, 3, (M, Q))
    qm, qW, qd = rng.randn(M, 1) * 0.3, rng.randn(M, M) * 0.4 / np.sqrt(M), rng.rand(M) * 0.4 + 0.1
    ls, var = rng.rand(Q) * 0.3 + (1.0 if Q >= 5 else 0.4), np.array([1.3])      # (well-conditioned Kuu: this is the explicit-inverse float32 form)
    noise = rng.rand(B, 1) * 0.2 + 0.02
    k = {'rbf': O.RBF, 'matern32': O.Matern32}[kind](Q, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')
    vals = dict(X=X, Y=Ys, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var)
    lv = {n: O.T(vals[n]).clone().requires_grad_(True) for n in names}
    logL = O.svgp_log_pdf(k, lv['X'], lv['Y'], lv['Z'][None], lv['noise'][None], lv['qm'][None], lv['qW'][None], lv['qd'][None],
                          {k.name + '_lengthscale': lv['ls'][None], k.name + '_variance': lv['var'][None]}, jitter=1e-6, log_pdf_scaling=1.5)
    grads = torch.autograd.grad(logL.mean(), [lv[n] for n in names])
    dev = torch.cuda.current_device()
    dt = torch.float32
    _lib.svgp_timing(dev, True)
    try:
        r = ops.svgp_logpdf(kind, _dev(X, dt), _dev(Ys, dt), _dev(Z, dt), _dev(noise, dt), _dev(qm, dt), _dev(qW, dt), _dev(qd, dt), _dev(ls, dt),
                            _dev(var, dt), True, jitter=1e-6, scaling=1.5, gscale=1.0 / S, want_grad=True)
        stages = _lib.svgp_timing_read(dev)
    finally:
        _lib.svgp_timing(dev, False)
    assert int(r['info'].abs().sum()) == 0
    if kind == 'rbf':
        assert 'planes_a' in stages and 't_gemm' in stages and 'reverse_pass' in stages, stages      # the streaming split path ran
    else:       # (r04: the Matern kinds keep the difference-form reverse pass -- gram_bwd.hip, mxf_svgp_bwd_is_mfma -- and with it the generic per-row-noise path)
        assert 'planes_a' not in stages, stages
    ref = logL.detach().numpy()
    assert np.abs(r['logL'].double().cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), (r['logL'], ref)
    for n, key in zip(names, ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar')):
        g = grads[names.index(n)].numpy()
        got = r[key].double().cpu().numpy().reshape(g.shape)
        err = np.linalg.norm(got - g) / max(np.linalg.norm(g), 1e-300)
        assert err <= 2e-3, (key, err)
