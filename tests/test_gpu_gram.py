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
