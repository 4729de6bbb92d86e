"""Edge cases through the C ABI: empty and single-element inputs, duplicated points (zero distances: the Matern clip), the widest
supported tiles (Q = 16, P = 8), inputs wider than the tiled kernels support (loud failure or generic fallback), and non-finite
propagation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _t(a, dt=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda()


def test_empty_inputs_are_no_ops():
    from mxfusion_amd import ops
    ls, var = _t(np.ones((1, 3))), _t(np.ones((1, 1)))
    K = ops.gram('rbf', _t(np.zeros((1, 0, 3))), _t(np.zeros((1, 5, 3))), ls, var, True)
    assert K.shape == (1, 0, 5)
    K = ops.gram('rbf', _t(np.zeros((1, 4, 3))), _t(np.zeros((1, 0, 3))), ls, var, True)
    assert K.shape == (1, 4, 0)
    C = ops.gemm(_t(np.zeros((1, 3, 0))), _t(np.zeros((1, 0, 4))))          # K = 0: C = 0
    assert C.shape == (1, 3, 4) and float(C.abs().max()) == 0.0
    s = ops.coldot(_t(np.zeros((1, 0, 6))), _t(np.zeros((1, 0, 6))))
    assert s.shape == (1, 6) and float(s.abs().max()) == 0.0


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 1e-6)])
def test_single_point_and_single_inducing_point(dtype, tol):
    from mxfusion_amd import ops
    rng = np.random.RandomState(0)
    X, Y, noise, ls, var = rng.rand(1, 1, 2), rng.rand(1, 1, 1), np.array([[0.3]]), rng.rand(1, 2) + 0.5, np.array([[0.9]])
    r = ops.gp_logpdf('rbf', _t(X, dtype), _t(Y, dtype), _t(noise, dtype), _t(ls, dtype), _t(var, dtype), True, want_grad=True)
    k = O.RBF(2, ARD=True)
    ref = O.gp_log_pdf(k, O.T(X), O.T(Y), O.T(noise), {'rbf_lengthscale': O.T(ls), 'rbf_variance': O.T(var)})
    assert abs(float(r['logL'][0]) - float(ref[0])) < tol * max(1.0, abs(float(ref[0])))
    # SVGP with B = 1, M = 1, P = 1
    Z, qm, qW, qd = rng.rand(1, 2), rng.randn(1, 1), rng.randn(1, 1) * 0.1, rng.rand(1) + 0.5
    r = ops.svgp_logpdf('rbf', _t(X, dtype), _t(Y, dtype), _t(Z, dtype), _t(noise[0], dtype), _t(qm, dtype), _t(qW, dtype), _t(qd, dtype),
                        _t(ls[0], dtype), _t(var[0], dtype), True, jitter=1e-8, want_grad=True)
    ref = O.svgp_log_pdf(k, O.T(X), O.T(Y), O.T(Z)[None], O.T(noise), O.T(qm)[None], O.T(qW)[None], O.T(qd)[None],
                         {'rbf_lengthscale': O.T(ls), 'rbf_variance': O.T(var)}, jitter=1e-8)
    assert abs(float(r['logL'][0]) - float(ref[0])) < max(tol, 1e-6 if dtype == torch.float32 else tol) * max(1.0, abs(float(ref[0])))
    assert all(bool(torch.isfinite(r[k_]).all()) for k_ in ('dX', 'dY', 'dZ', 'dnoise', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar'))


@pytest.mark.parametrize('kind', ['rbf', 'matern12', 'matern32', 'matern52'])
def test_duplicated_points_zero_distance(kind):
    """collisions: identical rows in X (and X2) give r2 = 0 exactly; Matern clips r2 at 1e-14 (matern.py:84,116,148) and its reverse
    mode must stay finite (the reference's sqrt has an infinite slope at 0 without the clip)."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(1)
    base = rng.rand(5, 3)
    X = np.concatenate([base, base[:3], base[:1]])[None]            # 9 rows, several exact duplicates
    ls, var = rng.rand(1, 3) + 0.5, np.array([[1.7]])
    k = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}[kind](3, ARD=True)
    K = ops.gram(kind, _t(X), None, _t(ls), _t(var), True)
    Ko = k.K(O.T(X), **{k.name + '_lengthscale': O.T(ls), k.name + '_variance': O.T(var)})
    assert np.allclose(K.cpu().numpy(), Ko.numpy(), rtol=1e-9, atol=1e-9)
    dK = rng.randn(1, 9, 9)
    dX, _, dls, dvar = ops.gram_bwd(kind, _t(X), None, _t(ls), _t(var), True, _t(dK))
    assert bool(torch.isfinite(dX).all()) and bool(torch.isfinite(dls).all()) and bool(torch.isfinite(dvar).all())
    tX, tls, tvar = [O.T(a).clone().requires_grad_(True) for a in (X, ls, var)]
    (k.K(tX, **{k.name + '_lengthscale': tls, k.name + '_variance': tvar}) * O.T(dK)).sum().backward()
    assert np.allclose(dvar.cpu().numpy(), tvar.grad.numpy(), rtol=1e-9, atol=1e-9)
    assert np.allclose(dls.cpu().numpy(), tls.grad.numpy(), rtol=1e-7, atol=1e-7)


def test_wide_inputs_beyond_the_tiled_kernels():
    """Q > 16 inputs and P > 8 outputs (the register tiles of the fast kernels): the Gram and its reverse mode fall back to the generic
    kernels, the SVGP bound runs the materialised-dKuf path and splits Y into column blocks (the bound is additive over output columns) --
    same values and gradients as the oracle, no refusal (the reference handles any Q and D)."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(2)
    Q = 20
    X, X2, ls, var = rng.rand(1, 33, Q), rng.rand(1, 17, Q), rng.rand(1, Q) + 0.8, np.array([[1.1]])
    dK = rng.randn(1, 33, 17)
    K = ops.gram('rbf', _t(X), _t(X2), _t(ls), _t(var), True)
    k = O.RBF(Q, ARD=True)
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=X, X2=X2, ls=ls, var=var).items()}
    Ko = k.K(lv['X'], lv['X2'], rbf_lengthscale=lv['ls'], rbf_variance=lv['var'])
    assert np.allclose(K.cpu().numpy(), Ko.detach().numpy(), rtol=1e-11, atol=1e-12)
    (Ko * O.T(dK)).sum().backward()
    g = dict(zip(('dX', 'dX2', 'dls', 'dvar'), ops.gram_bwd('rbf', _t(X), _t(X2), _t(ls), _t(var), True, _t(dK))))
    for key, n in (('dX', 'X'), ('dX2', 'X2'), ('dls', 'ls'), ('dvar', 'var')):
        assert np.allclose(g[key].cpu().numpy().reshape(lv[n].grad.shape), lv[n].grad.numpy(), rtol=1e-9, atol=1e-11), key
    # square Gram (both roles flow into dX), non-ARD length-scale, Matern
    k2 = O.Matern32(Q, ARD=False)
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=X, ls=ls[:, :1], var=var).items()}
    dKs = rng.randn(1, 33, 33)
    (k2.K(lv['X'], matern32_lengthscale=lv['ls'], matern32_variance=lv['var']) * O.T(dKs)).sum().backward()
    g = dict(zip(('dX', 'dX2', 'dls', 'dvar'), ops.gram_bwd('matern32', _t(X), None, _t(ls[:, :1]), _t(var), False, _t(dKs))))
    for key, n in (('dX', 'X'), ('dls', 'ls'), ('dvar', 'var')):
        assert np.allclose(g[key].cpu().numpy().reshape(lv[n].grad.shape), lv[n].grad.numpy(), rtol=1e-9, atol=1e-10), key
    # SVGP bound with Q = 20 inputs and P = 11 outputs through the module path (column blocks of 8 + 3)
    from mxfusion_amd.modules.gp_modules._fused import SVGPLogPdfFn
    B, M, P = 40, 6, 11
    Xs, Y, Z = rng.rand(2, B, Q), rng.rand(1, B, P), rng.rand(1, M, Q)
    qm, qW, qd, noise = rng.randn(1, M, P) * 0.1, rng.randn(1, M, M) * 0.05, rng.rand(1, M) + 0.5, np.array([[0.1]])
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=Xs, Z=Z, qm=qm, qW=qW, qd=qd, ls=ls, noise=noise).items()}
    ref = O.svgp_log_pdf(k, lv['X'], O.T(Y), lv['Z'], lv['noise'], lv['qm'], lv['qW'], lv['qd'], {'rbf_lengthscale': lv['ls'], 'rbf_variance': O.T(var)},
                         jitter=1e-6)
    ref.mean().backward()
    dv = {n: _t(v).requires_grad_(True) for n, v in dict(X=Xs, Z=Z, qm=qm, qW=qW, qd=qd, ls=ls, noise=noise).items()}
    total = 0
    for p0 in range(0, P, 8):
        sl = slice(p0, min(p0 + 8, P))
        total = total + SVGPLogPdfFn.apply(None, 'rbf', True, 1e-6, 1.0, dv['X'], _t(Y[..., sl]), dv['Z'], dv['noise'], dv['qm'][..., sl], dv['qW'], dv['qd'],
                                           dv['ls'], _t(var))[0]
    assert np.allclose(total.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-9)
    total.mean().backward()
    for n in lv:
        assert np.allclose(dv[n].grad.cpu().numpy(), lv[n].grad.numpy(), rtol=1e-7, atol=1e-9 * float(lv[n].grad.abs().max())), n


def test_nan_input_is_reported_not_hidden():
    """a NaN in the data makes the factorisation report a failing pivot (LAPACK-style info > 0) instead of returning a finite number"""
    from mxfusion_amd import ops, _lib
    X = np.random.RandomState(3).rand(1, 6, 2)
    X[0, 2, 1] = np.nan
    r = ops.gp_logpdf('rbf', _t(X), _t(np.ones((1, 6, 1))), _t([[0.1]]), _t(np.ones((1, 2))), _t([[1.0]]), True, want_grad=False)
    assert int(r['info'][0]) > 0
    with pytest.raises(_lib.MXFError):
        ops.check_info(r['info'])


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('S,M,N', [(1, 1000, 64), (3, 70, 5), (2, 64, 8192), (1, 40, 7), (2, 300, 20000)])
def test_coldot_shapes(dtype, S, M, N):
    """mxf_coldot (F.sum(A*B, axis=-2)): the few-column workgroup-per-16-columns kernel and the thread-per-column one, one operand shared
    over the sample axis."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(S + M + N)
    A, B = rng.randn(S, M, N), rng.randn(1, M, N)
    got = ops.coldot(torch.as_tensor(A, dtype=dtype).cuda(), torch.as_tensor(B, dtype=dtype).cuda())
    ref = (A * B).sum(-2)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert got.shape == (S, N)
    assert np.allclose(got.double().cpu().numpy(), ref, rtol=tol, atol=tol * np.abs(ref).max())
