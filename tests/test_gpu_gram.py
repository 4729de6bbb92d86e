"""GPU parity: HIP Gram kernel (through the C ABI) vs the CPU oracle and the golden fixtures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

KINDS = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}


def _dev(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def _oracle_K(kind, X, X2, ls, var, ard):
    k = KINDS[kind](X.shape[-1], ARD=ard)
    p = {k.name + '_lengthscale': O.T(ls), k.name + '_variance': O.T(var)}
    return k.K(O.T(X), None if X2 is None else O.T(X2), **p).numpy()


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 2e-6)])
def test_gram_golden(golden_dir, dtype, tol):
    from mxfusion_amd import ops
    g = np.load(os.path.join(golden_dir, 'kat_kernels.npz'))
    for kind in KINDS:
        K = ops.gram(kind, _dev(g['X'][None], dtype), _dev(g['X2'][None], dtype), _dev(g['ls'][None], dtype),
                     _dev(g['var'][None], dtype), True)
        assert np.allclose(K[0].cpu().numpy(), g['K_' + kind], rtol=tol, atol=tol), kind
        K = ops.gram(kind, _dev(g['X'][None], dtype), None, _dev(g['ls'][None], dtype), _dev(g['var'][None], dtype), True)
        assert np.allclose(K[0].cpu().numpy(), g['Kxx_' + kind], rtol=tol, atol=tol), kind
        # sampled X, X2, lengthscale and variance (S=3), non-ARD
        K = ops.gram(kind, _dev(g['Xs'], dtype), _dev(g['X2s'], dtype), _dev(g['lss'], dtype), _dev(g['vars'], dtype), False)
        assert np.allclose(K.cpu().numpy(), g['Ks_' + kind], rtol=tol, atol=tol), kind


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-11), (torch.float32, 1e-5)])
@pytest.mark.parametrize('kind', list(KINDS))
@pytest.mark.parametrize('N,N2,Q,S,SX2', [(1, 1, 1, 1, 1), (7, 5, 3, 1, 1), (130, 1030, 8, 2, 1), (257, 513, 5, 1, 3),
                                           (64, 2051, 16, 1, 1), (33, 70, 20, 2, 2), (300, None, 8, 2, None)])
def test_gram_shapes_vs_oracle(dtype, tol, kind, N, N2, Q, S, SX2):
    """ragged sizes (non-multiples of the tile, unaligned row strides), broadcast combos of the S axis,
    square (X2=None) Grams, Q > 16 fallback."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(hash((N, N2 or 0, Q)) % 2**31)
    X = rng.uniform(-3, 3, (S, N, Q))
    X2 = None if N2 is None else rng.uniform(-3, 3, (SX2, N2, Q))
    ard = Q % 2 == 1 or Q == 8
    ls = rng.rand(1, Q if ard else 1) * 2 + 0.7
    var = rng.rand(1, 1) + 0.5
    K = ops.gram(kind, _dev(X, dtype), None if X2 is None else _dev(X2, dtype), _dev(ls, dtype), _dev(var, dtype), ard)
    ref = _oracle_K(kind, X, X2, ls, var, ard)
    assert K.shape == ref.shape
    assert np.allclose(K.cpu().numpy(), ref, rtol=tol, atol=tol * float(var.max()))
    if X2 is None:   # exact symmetry and exact diagonal (stationary.py:123-124: Kdiag == variance)
        Kc = K.cpu().numpy()
        if kind == 'rbf' or dtype == torch.float64:
            assert np.array_equal(Kc, Kc.transpose(0, 2, 1))
        else:   # f32 Matern epilogues may contract FMAs differently in unrolled copies: symmetric to rounding only
            assert np.allclose(Kc, Kc.transpose(0, 2, 1), rtol=1e-6, atol=1e-7)
        if kind == 'rbf':   # Matern clips r2 at 1e-14 (matern.py:85) so its diagonal is var*exp(-c*1e-7), as in the oracle
            assert np.allclose(np.diagonal(Kc, axis1=1, axis2=2), var[0, 0], rtol=1e-7 if dtype == torch.float32 else 1e-15)


def test_gram_diag_add_modes_and_static_kernels():
    from mxfusion_amd import ops
    rng = np.random.RandomState(1)
    dt = torch.float64
    X = rng.rand(2, 50, 4)
    ls = rng.rand(1, 1) + 0.5
    var = rng.rand(1, 1) + 0.5
    noise = rng.rand(2, 1) + 0.1
    K = ops.gram('rbf', _dev(X, dt), None, _dev(ls, dt), _dev(var, dt), False, diag_add=_dev(noise, dt), jitter=1e-3)
    ref = _oracle_K('rbf', X, None, ls, var, False) + np.eye(50)[None] * (noise[:, :, None] + 1e-3)
    assert np.allclose(K.cpu().numpy(), ref, atol=1e-13)
    # AddKernel / MultiplyKernel accumulation modes (add_kernel.py:44-68, multiply_kernel.py:44-67)
    X2 = rng.rand(2, 31, 4)
    K = ops.gram('matern52', _dev(X, dt), _dev(X2, dt), _dev(ls, dt), _dev(var, dt), False)
    ops.gram('rbf', _dev(X, dt), _dev(X2, dt), _dev(ls * 1.7, dt), _dev(var * .5, dt), False, out=K, mode=ops.ACC_ADD)
    ref = _oracle_K('matern52', X, X2, ls, var, False) + _oracle_K('rbf', X, X2, ls * 1.7, var * .5, False)
    assert np.allclose(K.cpu().numpy(), ref, atol=1e-13)
    ops.gram('matern32', _dev(X, dt), _dev(X2, dt), _dev(ls, dt), _dev(var, dt), False, out=K, mode=ops.ACC_MUL)
    assert np.allclose(K.cpu().numpy(), ref * _oracle_K('matern32', X, X2, ls, var, False), atol=1e-13)
    # Linear / Bias / White (linear.py:59-89, static.py:56-74,125-150)
    lv = rng.rand(1, 4) + 0.1
    lin = O.Linear(4, ARD=True)
    K = ops.gram('linear', _dev(X, dt), _dev(X2, dt), _dev(lv, dt), None, True)
    assert np.allclose(K.cpu().numpy(), lin.K(O.T(X), O.T(X2), linear_variances=O.T(lv)).numpy(), atol=1e-13)
    K = ops.gram('bias', _dev(X, dt), _dev(X2, dt), None, _dev(var, dt), False)
    assert np.allclose(K.cpu().numpy(), var[0, 0])
    K = ops.gram('white', _dev(X, dt), None, None, _dev(var, dt), False)
    assert np.allclose(K.cpu().numpy(), np.eye(50)[None] * var[0, 0])
    K = ops.gram('white', _dev(X, dt), _dev(X2, dt), None, _dev(var, dt), False)
    assert np.all(K.cpu().numpy() == 0)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 3e-6)])
@pytest.mark.parametrize('N,N2,Q', [(50, 31, 4), (129, None, 8), (33, 300, 12), (1, 7, 1)])
def test_gram2_two_kernel_epilogue_vs_oracle(dtype, tol, N, N2, Q):
    """mxf_gram2: k1 + k2 and k1 * k2 of two stationary kernels in ONE pass (add_kernel.py:44-68, multiply_kernel.py:44-67) == the oracle's
    sum / product of the two Grams, every pair of kinds, ARD mixed with isotropic, sampled inputs and parameters, ragged shapes, the
    diagonal term of a square Gram."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(N + Q)
    S = 2
    X = rng.uniform(-2, 2, (S, N, Q))
    X2 = None if N2 is None else rng.uniform(-2, 2, (1, N2, Q))
    for k1 in KINDS:
        for k2 in KINDS:
            ls1, var1 = rng.rand(S, Q) + 0.5, rng.rand(1, 1) + 0.5
            ls2, var2 = rng.rand(1, 1) + 0.5, rng.rand(S, 1) + 0.5
            r1, r2 = _oracle_K(k1, X, X2, ls1, var1, True), _oracle_K(k2, X, X2, ls2, var2, False)
            for op, ref in ((ops.ACC_ADD, r1 + r2), (ops.ACC_MUL, r1 * r2)):
                dadd = rng.rand(S, 1) if N2 is None else None
                K = ops.gram2(k1, k2, op, _dev(X, dtype), None if X2 is None else _dev(X2, dtype), _dev(ls1, dtype), _dev(var1, dtype), True,
                              _dev(ls2, dtype), _dev(var2, dtype), False, diag_add=None if dadd is None else _dev(dadd, dtype), jitter=1e-3 if N2 is None else 0.0)
                if N2 is None:
                    ref = ref + np.eye(N)[None] * (dadd[:, :, None] + 1e-3)
                err = np.abs(K.double().cpu().numpy() - ref)
                if N2 is None:
                    # the diagonal of a Matern Gram is v f(sqrt(clip(r^2, 1e-14))): the reference's expansion form |x|^2 + |z|^2 - 2 x.z leaves r^2 ~ 1e-14
                    # of rounding there (sqrt -> 1e-7), the difference form an exact zero -- both clipped, equal to ~1e-7 only
                    d = np.arange(N)
                    assert err[:, d, d].max() <= 1e-6 * np.abs(ref).max(), (k1, k2, op)
                    err[:, d, d] = 0
                assert err.max() <= tol * max(1.0, np.abs(ref).max()), (k1, k2, op)


@pytest.mark.parametrize('comb', ['add', 'mul'])
def test_combination_kernel_classes_take_the_fused_pair_path_and_differentiate(comb):
    """AddKernel / MultiplyKernel of two stationary kernels (the deep-GP config's Matern52 + RBF) through the kernel classes: value and the
    gradients w.r.t. inputs and all four parameters == autograd through the oracle; the forward pass is ONE mxf_gram2 launch."""
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.components.distributions.gp.kernels import kernel as kmod
    rng = np.random.RandomState(3)
    dt = torch.float64
    S, N, N2, Q = 2, 40, 23, 3
    X, X2 = rng.uniform(-2, 2, (S, N, Q)), rng.uniform(-2, 2, (1, N2, Q))
    ls1, v1, ls2, v2 = rng.rand(1, Q) + 0.5, rng.rand(1, 1) + 0.5, rng.rand(1, 1) + 0.5, rng.rand(1, 1) + 0.5
    k = (Matern52(Q, ARD=True, dtype='float64') + RBF(Q, dtype='float64')) if comb == 'add' else (Matern52(Q, ARD=True, dtype='float64') * RBF(Q, dtype='float64'))
    ok = (O.AddKernel if comb == 'add' else O.MultiplyKernel)([O.Matern52(Q, ARD=True), O.RBF(Q)])
    names = ('X', 'X2', 'ls1', 'v1', 'ls2', 'v2')
    dv = {n: _dev(a, dt).requires_grad_(True) for n, a in zip(names, (X, X2, ls1, v1, ls2, v2))}
    ov = {n: O.T(a).clone().requires_grad_(True) for n, a in zip(names, (X, X2, ls1, v1, ls2, v2))}
    pre = comb + '_'
    calls = []
    orig = kmod._Gram2Fn.apply
    kmod._Gram2Fn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        K = k.K(None, dv['X'], dv['X2'], **{pre + 'matern52_lengthscale': dv['ls1'], pre + 'matern52_variance': dv['v1'], pre + 'rbf_lengthscale': dv['ls2'],
                                            pre + 'rbf_variance': dv['v2']})
    finally:
        kmod._Gram2Fn.apply = orig
    assert len(calls) == 1
    Kr = ok.K(ov['X'], ov['X2'], **{pre + 'matern52_lengthscale': ov['ls1'], pre + 'matern52_variance': ov['v1'], pre + 'rbf_lengthscale': ov['ls2'],
                                    pre + 'rbf_variance': ov['v2']})
    assert np.allclose(K.detach().cpu().numpy(), Kr.detach().numpy(), atol=1e-12)
    w = rng.randn(S, N, N2)
    (K * _dev(w, dt)).sum().backward()
    (Kr * O.T(w)).sum().backward()
    for n in names:
        assert np.allclose(dv[n].grad.cpu().numpy(), ov[n].grad.numpy(), rtol=1e-8, atol=1e-10), n


@pytest.mark.parametrize('kind', ['rbf', 'matern12', 'matern52'])
@pytest.mark.parametrize('sampled', ['none', 'X', 'X2'])
def test_float32_gram_does_not_depend_on_where_the_inputs_sit(kind, sampled):
    """Inputs at an offset of 10 000 units (raw time stamps, sensor readings): x / l rounds proportionally to |x| / l, which cost 3e-4 on K
    and on its reverse mode in float32 (1e-7 for centred inputs).  Both operands are centred on a common point before they are scaled
    (first row of X; of X2 when X is sampled and X2 shared): float32 K, K(X, X2) and the reverse mode against the oracle at 1e-6."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(3)
    S, N, N2, Q, off = 3, 70, 45, 5, 1.0e4
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    X = r32(off + rng.uniform(-2, 2, (S if sampled == 'X' else 1, N, Q)))
    X2 = r32(off + rng.uniform(-2, 2, (S if sampled == 'X2' else 1, N2, Q)))
    ls, var = r32(rng.rand(1, Q) + 0.8), r32([[1.2]])
    G = rng.randn(max(X.shape[0], X2.shape[0]), N, N2)
    ok = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern52': O.Matern52}[kind](Q, ARD=True)
    # the oracle on the CENTRED inputs (the same distances, exactly: X - off is exact in float64): the reference's own expansion-form distances
    # lose 6e-8 of r2 at this offset even in float64 (Matern12's diagonal comes out 1.1997 instead of 1.2)
    lv = {n: O.T(v).clone().requires_grad_(True) for n, v in (('X', X - off), ('X2', X2 - off), ('ls', ls), ('var', var))}
    kp = {ok.name + '_lengthscale': lv['ls'], ok.name + '_variance': lv['var']}
    K = ok.K(lv['X'], lv['X2'], **kp)
    gref = torch.autograd.grad((K * O.T(G)).sum(), [lv['X'], lv['X2'], lv['ls']])
    d = lambda a: torch.as_tensor(a, dtype=torch.float32).cuda()
    nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / np.linalg.norm(b.ravel()))
    Kd = ops.gram(kind, d(X), d(X2), d(ls), d(var), True)
    assert nrm(Kd.double().cpu().numpy(), K.detach().numpy()) <= 1e-6
    Ks = ops.gram(kind, d(X), None, d(ls), d(var), True)
    assert nrm(Ks.double().cpu().numpy(), ok.K(O.T(X - off), None, **{k: v.detach() for k, v in kp.items()}).numpy()) <= 1e-6
    g = ops.gram_bwd(kind, d(X), d(X2), d(ls), d(var), True, d(G))
    for got, ref, name in zip((g[0], g[1], g[2]), gref, ('dX', 'dX2', 'dls')):
        got = got.double().cpu().numpy()
        if got.shape != tuple(ref.shape):
            got = got.sum(0, keepdims=True)
        assert nrm(got, ref.numpy()) <= 2e-5, (name, nrm(got, ref.numpy()))
