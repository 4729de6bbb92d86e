"""CPU tests: the oracle against (a) numbers recorded by the reference itself, (b) the committed
golden fixtures, (c) independent closed forms (SciPy MVN / explicit-inverse Hensman / Titsias)."""
import json
import os

import numpy as np
import pytest
import torch
from scipy.stats import multivariate_normal

from oracle import gp_oracle as O

T = O.T


def _kp(k, ls, var):
    return {k.name + '_lengthscale': T(ls)[None], k.name + '_variance': T(var)[None]}


@pytest.fixture(scope="module")
def rec(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'reference_recorded.json')))


def test_notebook_trajectory_recorded_by_reference(rec):
    """examples/notebooks/gp_regression.ipynb cells 4,10,12,14: the reference's own printed losses."""
    np.random.seed(0)
    X = np.random.uniform(-3., 3., (20, 1))
    Y = np.sin(X) + np.random.randn(20, 1) * 0.05
    losses, final = O.run_map_gp_notebook(X, Y)
    for it, ref in rec['gp_notebook_loss_trajectory'].items():
        assert abs(losses[int(it)] - ref) <= 3e-6 * abs(ref), (it, losses[int(it)], ref)
    for k, ref in rec['gp_notebook_learned'].items():
        assert abs(final[k] - ref) < 1e-6, (k, final[k], ref)   # 6 printed digits
    # initial loss quoted in SURVEY 8(c) KAT-notebook
    k = O.RBF(1)
    raw = {'lengthscale': O.inv_softplus(T([1.])), 'variance': O.inv_softplus(T([1.])),
           'noise_var': O.inv_softplus(T([0.01]))}
    assert abs(float(O.map_gp_loss(k, T(X), T(Y), raw)) - (-8.321443970764)) < 1e-9


def test_gp_loglik_vs_scipy_and_golden(golden_dir, rec):
    g = np.load(os.path.join(golden_dir, 'kat_gp.npz'))
    k = O.RBF(3, ARD=True)
    logL = O.gp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['noise'])[None], _kp(k, g['ls'], g['var']))
    assert np.allclose(logL.numpy(), g['logL'], rtol=1e-13)
    assert abs(float(logL[0]) - rec['survey_kats']['gp_loglik']) < 1e-9
    # independent: sum_p log N(y_p | 0, K + s2 I) with a naive double-loop kernel
    X, ls, var = g['X'], g['ls'], g['var']
    K = np.array([[var[0] * np.exp(-0.5 * np.sum(((a - b) / ls) ** 2)) for b in X] for a in X])
    cov = K + g['noise'][0] * np.eye(10)
    ind = sum(multivariate_normal.logpdf(g['Y'][:, p], mean=None, cov=cov) for p in range(2))
    assert abs(float(logL[0]) - ind) < 1e-10
    assert np.allclose(g['mu_nf_diag'][0, 0], rec['survey_kats']['gp_pred_mu0'], atol=1e-7)
    assert abs(g['var_nf_diag'][0, 0] - rec['survey_kats']['gp_pred_var0']) < 1e-9
    # prediction vs textbook formulas
    Xt = g['Xt']
    Ks = np.array([[var[0] * np.exp(-0.5 * np.sum(((a - b) / ls) ** 2)) for b in Xt] for a in X])
    mu = Ks.T @ np.linalg.solve(cov, g['Y'])
    v = var[0] - np.sum(Ks * np.linalg.solve(cov, Ks), 0)
    assert np.allclose(g['mu_nf_diag'][0], mu, atol=1e-10)
    assert np.allclose(g['var_nf_diag'][0], v, atol=1e-10)
    assert np.allclose(g['var_noisy_diag'][0], v + g['noise'][0], atol=1e-10)


def test_svgp_bound_vs_hensman_and_golden(golden_dir, rec):
    g = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    k = O.RBF(3, ARD=True)
    args = [T(g[n])[None] for n in ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd')]
    logL = O.svgp_log_pdf(k, *args, _kp(k, g['ls'], g['var']), jitter=1e-8)
    assert np.allclose(logL.numpy(), g['logL'], rtol=1e-13)
    assert abs(float(logL[0]) - rec['survey_kats']['svgp_elbo']) < 1e-9
    ss = O.svgp_log_pdf_suffstats(k, *args, _kp(k, g['ls'], g['var']), jitter=1e-8)
    assert abs(float(ss[0]) - float(logL[0])) < 1e-9
    ss = O.svgp_log_pdf_suffstats(k, *args, _kp(k, g['ls'], g['var']), jitter=1e-8, log_pdf_scaling=3.5)
    assert abs(float(ss[0]) - float(g['logL_scaled'][0])) < 1e-9
    # independent Hensman-2013 bound with explicit inverses
    X, Y, Z, ls, var, s2 = g['X'], g['Y'], g['Z'], g['ls'], g['var'][0], g['noise'][0]
    kf = lambda A, B: np.array([[var * np.exp(-0.5 * np.sum(((a - b) / ls) ** 2)) for b in B] for a in A])
    Kuu = kf(Z, Z) + 1e-8 * np.eye(3)
    Kuf = kf(Z, X)
    Su = g['qW'] @ g['qW'].T + np.diag(g['qd'])
    A = Kuf.T @ np.linalg.inv(Kuu)
    fm = A @ g['qm']
    fv = var - np.sum(A * Kuf.T, 1) + np.sum((A @ Su) * A, 1)
    ell = np.sum(-0.5 * np.log(2 * np.pi * s2) - 0.5 * ((Y[:, 0] - fm[:, 0]) ** 2 + fv) / s2)
    kl = 0.5 * (np.trace(np.linalg.solve(Kuu, Su)) + g['qm'][:, 0] @ np.linalg.solve(Kuu, g['qm'][:, 0]) - 3
                + np.linalg.slogdet(Kuu)[1] - np.linalg.slogdet(Su)[1])
    assert abs(float(logL[0]) - (ell - kl)) < 1e-8
    assert np.allclose(g['mu_nf_diag'][0, :, 0], rec['survey_kats']['svgp_pred_mu'], atol=1e-7)
    assert np.allclose(g['var_nf_diag'][0, :, 0], rec['survey_kats']['svgp_pred_var'], atol=1e-7)
    # SVGP prediction: k** - diag(A Kuu A^T) + diag(A S A^T)
    Kus = kf(Z, g['Xt'])
    At = Kus.T @ np.linalg.inv(kf(Z, Z))
    assert np.allclose(g['mu_nf_diag'][0, :, 0], (At @ g['qm'])[:, 0], atol=1e-8)
    assert np.allclose(g['var_nf_diag'][0, :, 0], var - np.sum(At * Kus.T, 1) + np.sum((At @ Su) * At, 1), atol=1e-8)
    assert g['var_nf_diag'].shape == (1, 5, 1) and g['var_nf_full'].shape == (1, 5, 5, 1)


def test_sgp_bound_vs_titsias_and_golden(golden_dir, rec):
    g = np.load(os.path.join(golden_dir, 'kat_sgp.npz'))
    k = O.RBF(3, ARD=True)
    logL = O.sgp_log_pdf(k, T(g['X'])[None], T(g['Y'])[None], T(g['Z'])[None], T(g['noise'])[None],
                         _kp(k, g['ls'], g['var']), jitter=1e-8)
    assert np.allclose(logL.numpy(), g['logL'], rtol=1e-13)
    assert abs(float(logL[0]) - rec['survey_kats']['sgp_bound']) < 1e-9
    X, Y, Z, ls, var, s2 = g['X'], g['Y'], g['Z'], g['ls'], g['var'][0], g['noise'][0]
    kf = lambda A, B: np.array([[var * np.exp(-0.5 * np.sum(((a - b) / ls) ** 2)) for b in B] for a in A])
    Kuu = kf(Z, Z) + 1e-8 * np.eye(3)
    Kuf = kf(Z, X)
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    ind = sum(multivariate_normal.logpdf(Y[:, p], cov=Qff + s2 * np.eye(10)) for p in range(2)) \
        - 2 / (2 * s2) * (10 * var - np.trace(Qff))
    assert abs(float(logL[0]) - ind) < 1e-8


def test_kernel_spots_and_naive(golden_dir, rec):
    g = np.load(os.path.join(golden_dir, 'kat_kernels.npz'))
    for name, ref in rec['survey_kats']['kernel_spot'].items():
        assert abs(g['K_' + name][0, 0] - ref) < 1e-11
    X, X2, ls, var = g['X'], g['X2'], g['ls'], g['var'][0]
    r = np.array([[np.sqrt(np.sum(((a - b) / ls) ** 2)) for b in X2] for a in X])
    assert np.allclose(g['K_rbf'], var * np.exp(-0.5 * r ** 2), atol=1e-13)
    assert np.allclose(g['K_matern52'], var * (1 + np.sqrt(5) * r + 5 / 3 * r ** 2) * np.exp(-np.sqrt(5) * r), atol=1e-12)
    assert np.allclose(g['K_matern32'], var * (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r), atol=1e-12)
    assert np.allclose(g['K_matern12'], var * np.exp(-r), atol=1e-12)
    # Kdiag is exactly the variance (stationary.py:123-124)
    assert np.allclose(g['Kdiag_rbf'], np.broadcast_to(g['vars'], (3, 6)))


def test_gradients_finite_difference(golden_dir):
    """The reference never tests GP gradients numerically; pin the oracle's autograd
    gradients with central finite differences."""
    g = np.load(os.path.join(golden_dir, 'kat_svgp.npz'))
    k = O.RBF(3, ARD=True)
    names = ('X', 'Y', 'Z', 'noise', 'qm', 'qW', 'qd', 'ls', 'var')

    def f(d):
        return float(O.svgp_log_pdf(k, T(d['X'])[None], T(d['Y'])[None], T(d['Z'])[None], T(d['noise'])[None],
                                    T(d['qm'])[None], T(d['qW'])[None], T(d['qd'])[None],
                                    _kp(k, d['ls'], d['var']), jitter=1e-8, log_pdf_scaling=3.5)[0])
    base = {n: g[n].copy() for n in names}
    rng = np.random.RandomState(0)
    for n in names:
        for _ in range(2):
            idx = tuple(rng.randint(0, s) for s in base[n].shape)
            h = 1e-6
            dp = {m: v.copy() for m, v in base.items()}
            dm = {m: v.copy() for m, v in base.items()}
            dp[n][idx] += h
            dm[n][idx] -= h
            fd = (f(dp) - f(dm)) / (2 * h)
            assert abs(fd - g['d_' + n][idx]) < 1e-5 * max(1., abs(fd)), (n, idx, fd, g['d_' + n][idx])


def test_svi_trajectory_regenerates(golden_dir):
    g = np.load(os.path.join(golden_dir, 'kat_svi.npz'))
    k = O.RBF(3, ARD=True)
    raw = {n[5:]: T(g[n]) for n in g.files if n.startswith('init_')}
    opt = O.MXNetAdam(0.1)
    for it in range(3):
        lv = {n: v.clone().requires_grad_(True) for n, v in raw.items()}
        loss = O.svi_latent_svgp_loss(k, T(g['Y']), lv['Z'], lv, T(g['eps'][it]), jitter=1e-8)
        loss.backward()
        assert abs(float(loss.detach()) - g['losses'][it]) < 1e-10
        raw = opt.step({n: v.detach() for n, v in lv.items()}, {n: v.grad for n, v in lv.items()})
    for n, v in raw.items():
        assert np.allclose(v.numpy(), g['final_' + n], atol=1e-12)
