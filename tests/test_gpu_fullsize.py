"""Size-independent properties at BASELINE.json's FULL sizes: the N=65 536 RBF Gram (configs' metric kernel), the exact-GP factorisation at
N=8 192 (configs[1]) and the SVGP training call at N=65 536, M=1 024 (configs[2]) incl. all 32 samples (2.1 M columns per call, which the
oracle's materialising form cannot hold).  The comparison with the ORACLE itself at these shapes (S <= 2) is tests/test_gpu_fullsize_oracle.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _synth(N, Q, M, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3., 3., (N, Q))
    Y = np.sin(X @ rng.standard_normal(Q))[:, None] + 0.05 * rng.standard_normal((N, 1))
    Z = X[rng.permutation(N)[:M]].copy()
    return rng, X, Y, Z


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.float64, 1e-13)])
def test_full_size_gram_symmetry_range_and_oracle_sub_blocks(dtype, tol):
    """N = 65 536, Q = 8: unit diagonal * variance, symmetry, range (0, variance], and 256 x 256 sub-blocks (rows / columns drawn from
    the whole matrix) equal to the oracle's Gram of the same points."""
    from mxfusion_amd import ops
    N, Q = 65536, 8
    rng, X, _, _ = _synth(N, Q, 8)
    ls, var = rng.random(Q) + 0.7, np.array([1.7])
    Xd = torch.as_tensor(X, dtype=dtype).cuda()[None]
    K = ops.gram('rbf', Xd, None, torch.as_tensor(ls, dtype=dtype).cuda()[None], torch.as_tensor(var, dtype=dtype).cuda()[None], True)[0]
    d = torch.diagonal(K)
    assert float((d - var[0]).abs().max()) <= tol * var[0]
    assert float(K.max()) <= var[0] * (1 + tol) and float(K.min()) >= 0.0
    k = O.RBF(Q, ARD=True)
    kp = {'rbf_lengthscale': O.T(ls)[None], 'rbf_variance': O.T(var)[None]}
    for t in range(3):
        ri, ci = np.sort(rng.choice(N, 256, replace=False)), np.sort(rng.choice(N, 256, replace=False))
        ref = k.K(O.T(X[ri])[None], O.T(X[ci])[None], **kp)[0].numpy()
        rt, ct = torch.as_tensor(ri).cuda(), torch.as_tensor(ci).cuda()
        got = K[rt][:, ct].double().cpu().numpy()
        assert np.allclose(got, ref, rtol=0, atol=tol * var[0] * 4)
        assert torch.equal(K[rt][:, ct], K[ct][:, rt].T)                      # symmetric to the bit
    del K


def test_full_size_exact_gp_factorisation_reconstructs():
    """configs[1]: N = 8 192, float64.  L L^T = K + noise I and L L^-1 = I to rounding; log-det from the factor == 2 sum log diag."""
    from mxfusion_amd import ops
    N, Q = 8192, 8
    rng, X, Y, _ = _synth(N, Q, 1, seed=1)
    dt = torch.float64
    Xd = torch.as_tensor(X, dtype=dt).cuda()[None]
    one = lambda v: torch.as_tensor(np.atleast_1d(v), dtype=dt).cuda()[None]
    K = ops.gram('rbf', Xd, None, one(np.ones(Q)), one(1.0), True, diag_add=one(0.01), jitter=0.0)
    L, info = ops.potrf_(K.clone())
    assert int(info.abs().sum()) == 0
    R = ops.gemm(L, L, transB=True)
    assert float((R - K).abs().max()) <= 1e-12 * float(K.abs().max())
    Linv = ops.trtri(L)
    E = ops.gemm(L, Linv)
    E.diagonal(dim1=-2, dim2=-1).sub_(1.0)
    assert float(E.abs().max()) <= 1e-9                                      # cond(L) ~ 1e2 at noise 0.01
    assert abs(float(ops.sumlogdiag(L)[0]) - float(torch.log(torch.diagonal(L[0])).sum())) <= 1e-9 * N
    r = ops.gp_logpdf('rbf', Xd, torch.as_tensor(Y, dtype=dt).cuda()[None], one(0.01), one(np.ones(Q)), one(1.0), True, jitter=0.0, want_grad=True)
    a = ops.trsm_(L, torch.as_tensor(Y, dtype=dt).cuda()[None].clone())
    ref = -float(ops.sumlogdiag(L)[0]) - 0.5 * float((a ** 2).sum()) - 0.5 * N * np.log(2 * np.pi)
    assert abs(float(r['logL'][0]) - ref) <= 1e-10 * abs(ref)
    # reverse mode at full size: dlogL/dK = 1/2 (alpha alpha^T - K^-1) formed with torch from the same L^-1, pushed through mxf_gram_bwd --
    # every lower tile of the library's triangular K^-1 product (balanced tile mapping, gemm.hip) has to be there
    alpha = Linv[0].T @ a[0]
    dK = 0.5 * (alpha @ alpha.T - Linv[0].T @ Linv[0])
    gx, _, gls, gvar = ops.gram_bwd('rbf', Xd, None, one(np.ones(Q)), one(1.0), True, dK[None])
    for k, g in (('dX', gx), ('dls', gls), ('dvar', gvar)):
        assert float((r[k] - g).abs().max()) <= 1e-9 * float(g.abs().max()), k
    assert abs(float(r['dnoise'].sum()) - float(torch.trace(dK))) <= 1e-10 * abs(float(torch.trace(dK)))


def test_full_size_exact_gp_map_step_vs_oracle():
    """configs[1] at its FULL size against the ORACLE (missing 4): the oracle's gp_log_pdf (gp_regression.py:55-70 op for op: Gram,
    potrf, trsm, sumlogdiag on the host, float64) and its autograd gradients at N = 8 192, Q = 8, P = 1 -- log marginal likelihood and
    dX, dY, dnoise, dlengthscale, dvariance of the HIP path to 1e-9."""
    from oracle import gp_oracle as O
    from mxfusion_amd import ops
    N, Q = 8192, 8
    rng, X, Y, _ = _synth(N, Q, 1, seed=1)
    ls, var, noise = rng.uniform(0.8, 1.3, Q), np.array([1.2]), np.array([0.013])
    torch.set_num_threads(min(32, torch.get_num_threads() or 1))
    lv = {k: O.T(v).clone().requires_grad_(True) for k, v in (('X', X[None]), ('Y', Y[None]), ('noise', noise[None]), ('ls', ls[None]), ('var', var[None]))}
    ref = O.gp_log_pdf(O.RBF(Q, ARD=True), lv['X'], lv['Y'], lv['noise'], {'rbf_lengthscale': lv['ls'], 'rbf_variance': lv['var']}, jitter=0.)
    gref = torch.autograd.grad(ref.sum(), list(lv.values()))
    d = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()
    r = ops.gp_logpdf('rbf', d(X)[None], d(Y)[None], d(noise)[None], d(ls)[None], d(var)[None], True, jitter=0.0, want_grad=True)
    torch.cuda.synchronize()
    assert int(r['info'].abs().sum()) == 0
    assert abs(float(r['logL'][0]) - float(ref.detach())) <= 1e-9 * abs(float(ref.detach()))
    for key, g in zip(('dX', 'dY', 'dnoise', 'dls', 'dvar'), gref):
        a, b = r[key].double().cpu().numpy().ravel(), g.numpy().ravel()
        assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), (key, np.abs(a - b).max(), np.abs(b).max())


def test_full_size_svgp_training_call_properties():
    """configs[2] shapes (N = 65 536, Q = 8, M = 1 024; 2 samples): (i) float32 training path (split GEMMs on the f16 pipe) vs float64:
    ELBO to 1e-5 (north_star), gradients to the f32 tolerance; (ii) float64 central differences of the ELBO in the noise and the kernel
    variance reproduce dnoise / dvar; (iii) row additivity of the bound: L(A u B) + L(C) == L(A) + L(B u C) (the data term is a sum over
    rows, the KL term appears once per call) -- the identity the row-sharded multi-GPU layout relies on."""
    from mxfusion_amd import ops
    N, Q, M, S, P = 65536, 8, 1024, 2, 1
    rng, X0, Y, Z = _synth(N, Q, M, seed=3)
    X = X0[None] + 0.1 * rng.standard_normal((S, N, Q))
    qm, qW, qd = rng.standard_normal((M, P)) * 0.3, rng.standard_normal((M, M)) * 0.02, rng.random(M) + 0.5
    ls, var, noise = np.ones(Q) + 0.2 * rng.random(Q), np.array([1.2]), np.array([0.02])

    def call(dt, noise_=noise, var_=var, rows=slice(None), grad=True):
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()
        return ops.svgp_logpdf('rbf', d(X[:, rows]), d(Y[rows][None]), d(Z), d(noise_), d(qm), d(qW), d(qd), d(ls), d(var_), True, jitter=1e-6,
                               scaling=1.0, gscale=1.0 / S, want_grad=grad)
    r64, r32 = call(torch.float64), call(torch.float32)
    assert int(r64['info'].abs().sum()) == 0 and int(r32['info'].abs().sum()) == 0
    l64 = r64['logL'].cpu().numpy()
    assert np.allclose(r32['logL'].double().cpu().numpy(), l64, rtol=1e-5, atol=0)
    for key in ('dX', 'dZ', 'dW', 'dSdiag', 'dmu', 'dls', 'dvar', 'dnoise'):
        a, b = r64[key].cpu().numpy(), r32[key].double().cpu().numpy()
        assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(a), key
    # (ii) central differences (float64): d mean_s(logL) / d theta
    for name, base, key in (('noise', noise, 'dnoise'), ('var', var, 'dvar')):
        h = 1e-6 * base
        kw = {name + '_': base + h}
        lp = call(torch.float64, grad=False, **kw)['logL'].mean()
        kw = {name + '_': base - h}
        lm = call(torch.float64, grad=False, **kw)['logL'].mean()
        fd = float((lp - lm) / (2 * h[0]))
        an = float(r64[key].sum())
        assert abs(fd - an) <= 2e-6 * abs(an), (name, fd, an)
    # (iii) row additivity
    A, B_, C = slice(0, 20000), slice(20000, 45000), slice(45000, N)
    AB, BC = slice(0, 45000), slice(20000, N)
    L = lambda rows: call(torch.float64, rows=rows, grad=False)['logL'].cpu().numpy()
    lhs, rhs = L(AB) + L(C), L(A) + L(BC)
    assert np.allclose(lhs, rhs, rtol=1e-11, atol=0)


def test_config3_at_all_32_samples():
    """configs[2] at its full sample count (N = 65 536, Q = 8, M = 1 024, S = 32: 2.1 M columns per call -- what bench.py times): the
    per-sample bounds of the float32 call agree with float64 to 1e-5 (north_star); the samples are independent, so the S = 32 call equals
    the two S = 16 calls on its halves sample by sample, and its gradient (weight 1/32 each) is the mean of theirs -- the identity the
    sample-sharded multi-GPU layout relies on."""
    from mxfusion_amd import ops
    N, Q, M, S, P = 65536, 8, 1024, 32, 1
    rng, X0, Y, Z = _synth(N, Q, M, seed=5)
    qm, qW, qd = rng.standard_normal((M, P)) * 0.3, rng.standard_normal((M, M)) * 0.02, rng.random(M) + 0.5
    ls, var, noise = np.ones(Q) + 0.2 * rng.random(Q), np.array([1.2]), np.array([0.02])
    gen = torch.Generator(device='cuda').manual_seed(11)
    X64 = torch.as_tensor(X0).cuda()[None] + 0.1 * torch.randn(S, N, Q, generator=gen, device='cuda', dtype=torch.float64)

    def call(dt, sl):
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()
        Xs = X64[sl].to(dt).contiguous()
        r = ops.svgp_logpdf('rbf', Xs, d(Y[None]), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, scaling=1.0,
                            gscale=1.0 / Xs.shape[0], want_grad=True)
        torch.cuda.synchronize()
        assert int(r['info'].abs().sum()) == 0
        return {k: v.double().cpu().numpy() for k, v in r.items() if k != 'dX'}
    full = call(torch.float32, slice(0, S))
    lo, hi = call(torch.float32, slice(0, S // 2)), call(torch.float32, slice(S // 2, S))
    assert np.allclose(full['logL'], np.concatenate([lo['logL'], hi['logL']]), rtol=2e-7, atol=0)
    for key in ('dZ', 'dW', 'dSdiag', 'dmu', 'dls', 'dvar', 'dnoise'):
        a, b = full[key], 0.5 * (lo[key] + hi[key])
        assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(a), (key, np.linalg.norm(a - b) / np.linalg.norm(a))
    f64 = call(torch.float64, slice(0, S))
    assert np.allclose(full['logL'], f64['logL'], rtol=1e-5, atol=0), np.abs(full['logL'] / f64['logL'] - 1).max()
    for key in ('dZ', 'dW', 'dSdiag', 'dmu', 'dls', 'dvar', 'dnoise'):
        assert np.linalg.norm(full[key] - f64[key]) <= 1e-4 * np.linalg.norm(f64[key]), key
