"""
Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/gp_oracle.py).

Inputs are the reference tests' own seeded inputs (testing/modules/*_test.py,
testing/components/distributions/gp/kernel_test.py) and the notebook's data
(examples/notebooks/gp_regression.ipynb).  `reference_recorded.json` holds numbers that were
*printed by the reference itself* (notebook cell outputs) -- data, not code.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gp_oracle as O  # noqa: E402

T = O.T


def kp(kern, ls, var):
    return {kern.name + '_lengthscale': T(ls)[None], kern.name + '_variance': T(var)[None]}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez(os.path.join(HERE, name + '.npz'), **out)


def grads_of(fn, leaves):
    """leaves: dict name->numpy; fn(dict of tensors)->scalar.  Returns value, dict of grads."""
    ts = {k: T(v).clone().requires_grad_(True) for k, v in leaves.items()}
    val = fn(ts)
    val.backward()
    return val.detach(), {('d_' + k): (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in ts.items()}


def kat_gp():
    # testing/modules/gpregression_test.py:40-48, :171
    np.random.seed(0)
    D = 2
    X = np.random.rand(10, 3)
    Y = np.random.rand(10, D)
    noise = np.random.rand(1)
    ls = np.random.rand(3)
    var = np.random.rand(1)
    Xt = np.random.rand(20, 3)
    k = O.RBF(3, ARD=True)
    logL, (Xc, L, LinvY) = O.gp_log_pdf(k, T(X)[None], T(Y)[None], T(noise)[None], kp(k, ls, var),
                                        return_posterior=True)
    out = {}
    for nf in (True, False):
        for dg in (True, False):
            mu, v = O.gp_predict(k, T(Xt)[None], T(noise)[None], Xc[None], L[None], LinvY[None], kp(k, ls, var),
                                 noise_free=nf, diagonal_variance=dg)
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            out['mu_' + tag] = mu
            out['var_' + tag] = v

    def f(t):
        return O.gp_log_pdf(k, t['X'][None], t['Y'][None], t['noise'][None],
                            {'rbf_lengthscale': t['ls'][None], 'rbf_variance': t['var'][None]})[0]
    _, g = grads_of(f, dict(X=X, Y=Y, noise=noise, ls=ls, var=var))
    save('kat_gp', X=X, Y=Y, noise=noise, ls=ls, var=var, Xt=Xt, logL=logL, L=L, LinvY=LinvY, **out, **g)


def kat_svgp():
    # testing/modules/svgpregression_test.py:41-56, :68, :174
    np.random.seed(0)
    D = 1
    X = np.random.rand(10, 3)
    Y = np.random.rand(10, D)
    Z = np.random.rand(3, 3)
    qm = np.random.rand(3, D)
    qW = np.random.rand(3, 3)
    qd = np.random.rand(3,)
    noise = np.random.rand(1)
    ls = np.random.rand(3)
    var = np.random.rand(1)
    Xt = np.random.rand(5, 3)
    k = O.RBF(3, ARD=True)
    args = (T(X)[None], T(Y)[None], T(Z)[None], T(noise)[None], T(qm)[None], T(qW)[None], T(qd)[None])
    logL = O.svgp_log_pdf(k, *args, kp(k, ls, var), jitter=1e-8)
    logL_scaled = O.svgp_log_pdf(k, *args, kp(k, ls, var), jitter=1e-8, log_pdf_scaling=3.5)
    out = {}
    for nf in (True, False):
        for dg in (True, False):
            mu, v = O.svgp_predict(k, T(Xt)[None], T(Z)[None], T(noise)[None], T(qm)[None], T(qW)[None],
                                   T(qd)[None], kp(k, ls, var), noise_free=nf, diagonal_variance=dg)
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            out['mu_' + tag] = mu
            out['var_' + tag] = v

    def f(t):
        return O.svgp_log_pdf(k, t['X'][None], t['Y'][None], t['Z'][None], t['noise'][None], t['qm'][None],
                              t['qW'][None], t['qd'][None],
                              {'rbf_lengthscale': t['ls'][None], 'rbf_variance': t['var'][None]},
                              jitter=1e-8, log_pdf_scaling=3.5)[0]
    _, g = grads_of(f, dict(X=X, Y=Y, Z=Z, noise=noise, qm=qm, qW=qW, qd=qd, ls=ls, var=var))
    # heteroscedastic noise (svgpregression_test.py:143-168 exercises noise_var of shape (N,D))
    np.random.seed(1)
    noise_het = np.random.rand(10, D) + 0.1
    logL_het = O.svgp_log_pdf(k, T(X)[None], T(Y)[None], T(Z)[None], T(noise_het)[None], T(qm)[None], T(qW)[None],
                              T(qd)[None], kp(k, ls, var), jitter=1e-8)
    save('kat_svgp', X=X, Y=Y, Z=Z, qm=qm, qW=qW, qd=qd, noise=noise, ls=ls, var=var, Xt=Xt, logL=logL,
         logL_scaled=logL_scaled, noise_het=noise_het, logL_het=logL_het, **out, **g)


def kat_sgp():
    # testing/modules/sparsegpregression_test.py:38-47, :58
    np.random.seed(0)
    D = 2
    X = np.random.rand(10, 3)
    Y = np.random.rand(10, D)
    Z = np.random.rand(3, 3)
    noise = np.random.rand(1)
    ls = np.random.rand(3)
    var = np.random.rand(1)
    Xt = np.random.rand(20, 3)
    k = O.RBF(3, ARD=True)
    logL, (wv, L, LA) = O.sgp_log_pdf(k, T(X)[None], T(Y)[None], T(Z)[None], T(noise)[None], kp(k, ls, var),
                                      jitter=1e-8, return_posterior=True)
    out = {}
    for nf in (True, False):
        for dg in (True, False):
            mu, v = O.sgp_predict(k, T(Xt)[None], T(Z)[None], T(noise)[None], L[None], LA[None], wv[None],
                                  kp(k, ls, var), noise_free=nf, diagonal_variance=dg)
            tag = ('nf' if nf else 'noisy') + ('_diag' if dg else '_full')
            out['mu_' + tag] = mu
            out['var_' + tag] = v

    def f(t):
        return O.sgp_log_pdf(k, t['X'][None], t['Y'][None], t['Z'][None], t['noise'][None],
                             {'rbf_lengthscale': t['ls'][None], 'rbf_variance': t['var'][None]}, jitter=1e-8)[0]
    _, g = grads_of(f, dict(X=X, Y=Y, Z=Z, noise=noise, ls=ls, var=var))
    save('kat_sgp', X=X, Y=Y, Z=Z, noise=noise, ls=ls, var=var, Xt=Xt, logL=logL, wv=wv, L=L, LA=LA, **out, **g)


def kat_kernels():
    # SURVEY 8(c) "KAT-kernel spot" + a sample-axis broadcast matrix in the spirit of
    # testing/components/distributions/gp/kernel_test.py:101-107 (own seeds)
    np.random.seed(0)
    X = np.random.rand(5, 2)
    X2 = np.random.rand(4, 2)
    ls = np.random.rand(2) + 1e-4
    var = np.random.rand(1) + 1e-4
    out = dict(X=X, X2=X2, ls=ls, var=var)
    for name, cls in (('rbf', O.RBF), ('matern52', O.Matern52), ('matern32', O.Matern32), ('matern12', O.Matern12)):
        k = cls(2, ARD=True)
        out['K_' + name] = k.K(T(X)[None], T(X2)[None], **kp(k, ls, var))[0]
        out['Kxx_' + name] = k.K(T(X)[None], **kp(k, ls, var))[0]
    # batched / sampled inputs (S=3), sampled lengthscale & variance, non-ARD
    rng = np.random.RandomState(7)
    Xs = rng.rand(3, 6, 4)
    X2s = rng.rand(3, 5, 4)
    lss = rng.rand(3, 1) + 0.3
    vars_ = rng.rand(3, 1) + 0.2
    out.update(Xs=Xs, X2s=X2s, lss=lss, vars=vars_)
    for name, cls in (('rbf', O.RBF), ('matern52', O.Matern52), ('matern32', O.Matern32), ('matern12', O.Matern12)):
        k = cls(4, ARD=False)
        p = {k.name + '_lengthscale': T(lss), k.name + '_variance': T(vars_)}
        out['Ks_' + name] = k.K(T(Xs), T(X2s), **p)
        out['Kdiag_' + name] = k.Kdiag(T(Xs), **p)
    # linear / bias / white / add / mul
    lin = O.Linear(4, ARD=True)
    lv = rng.rand(4) + 0.1
    out['lin_variances'] = lv
    out['K_linear'] = lin.K(T(Xs), T(X2s), linear_variances=T(lv)[None])
    out['Kdiag_linear'] = lin.Kdiag(T(Xs), linear_variances=T(lv)[None])
    add = O.AddKernel([O.Matern52(4, ARD=False), O.RBF(4, ARD=False)])
    p = {'add_matern52_lengthscale': T(lss), 'add_matern52_variance': T(vars_),
         'add_rbf_lengthscale': T(lss * 1.7), 'add_rbf_variance': T(vars_ * 0.5)}
    out['K_add'] = add.K(T(Xs), T(X2s), **p)
    mul = O.MultiplyKernel([O.Matern32(4, ARD=False), O.RBF(4, ARD=False)])
    p = {'mul_matern32_lengthscale': T(lss), 'mul_matern32_variance': T(vars_),
         'mul_rbf_lengthscale': T(lss * 1.7), 'mul_rbf_variance': T(vars_ * 0.5)}
    out['K_mul'] = mul.K(T(Xs), T(X2s), **p)
    save('kat_kernels', **out)


def kat_svi():
    """A 3-iteration Adam trajectory of the SVI objective on the latent-input SVGP model of
    testing/modules/svgpregression_test.py:357-385 (S=4 injected-noise samples)."""
    np.random.seed(0)
    D = 1
    _X = np.random.rand(10, 3)
    Y = np.random.rand(10, D)
    Z = np.random.rand(3, 3)
    qm = np.random.rand(3, D)
    qW = np.random.rand(3, 3)
    qd = np.random.rand(3,)
    noise = np.random.rand(1)
    ls = np.random.rand(3)
    var = np.random.rand(1)
    rng = np.random.RandomState(3)
    S = 4
    eps = rng.randn(3, S, 10, 3)          # one noise block per iteration
    qXm = rng.randn(10, 3) * 0.5
    qXv = rng.rand(10, 3) * 0.5 + 0.1
    k = O.RBF(3, ARD=True)
    raw = {'qX_mean': T(qXm), 'qX_var': O.inv_softplus(T(qXv)), 'noise_var': O.inv_softplus(T(noise)),
           'lengthscale': O.inv_softplus(T(ls)), 'variance': O.inv_softplus(T(var)),
           'qU_mean': T(qm), 'qU_cov_W': T(qW), 'qU_cov_diag': O.inv_softplus(T(qd)), 'Z': T(Z)}
    init = {('init_' + k_): v.clone() for k_, v in raw.items()}
    opt = O.MXNetAdam(0.1)
    losses = []
    g0 = None
    for it in range(3):
        lv = {k_: v.clone().requires_grad_(True) for k_, v in raw.items()}
        loss = O.svi_latent_svgp_loss(k, T(Y), lv['Z'], lv, T(eps[it]), jitter=1e-8)
        loss.backward()
        losses.append(float(loss.detach()))
        if it == 0:
            g0 = {('g0_' + k_): v.grad.clone() for k_, v in lv.items()}
        raw = opt.step({k_: v.detach() for k_, v in lv.items()}, {k_: v.grad for k_, v in lv.items()}, batch_size=1)
    final = {('final_' + k_): v for k_, v in raw.items()}
    save('kat_svi', Y=Y, eps=eps, losses=np.array(losses), **init, **g0, **final)


def reference_recorded():
    rec = {
        "_source": "numbers printed by MXFusion itself in /root/reference/examples/notebooks/gp_regression.ipynb",
        "gp_notebook_loss_trajectory": {  # cell 12 output (Adam lr=0.05, float64)
            "10": -13.09287954321266, "20": -15.971970034359586, "30": -16.725359053995163,
            "40": -16.835084442759314, "50": -16.850332113428053, "60": -16.893812683762203,
            "70": -16.900137667771077, "80": -16.901158761459012, "90": -16.903085976668137,
            "100": -16.903135093930537},
        "gp_notebook_learned": {"variance": 0.616992, "lengthscale": 1.649073, "noise_var": 0.002251},  # cell 14
        "gp_notebook_gpy_optimum": {"objective": -16.903456670910902, "variance": 0.6148038604494702,
                                    "lengthscale": 1.6500299722611123, "noise_var": 0.002270049772204339},  # cell 16
        # examples/notebooks/svgp_regression.ipynb: cell 13 output (the learned hyper-parameters after 50 epochs at lr 0.1 + 50 at 0.01,
        # N=1000, M=20, minibatch 10) and the per-epoch mean losses printed by cell 11 (verbose=True); MXNet-RNG dependent (initial
        # qU_*, DataLoader shuffles), so a BAND pin, not a bit pin
        "svgp_notebook_learned": {"variance": 0.220715, "lengthscale": 0.498507, "noise_var": 0.003107},
        "svgp_notebook_epoch_losses": {"phase1": [10413624.614005275, 686034.5295730559, 427065.8343717841, 297071.493696023, 219808.0871498559, 169486.20729875282, 134765.1471133905, 109798.66321648406, 91257.8705670977, 77084.06942481917, 65962.38163622493, 57037.39009905885, 49725.50869601666, 43635.70855486856, 38501.415430223606, 34139.30892930683, 30414.713307491817, 27222.957705478882, 24466.753696665117, 22063.866203795988, 19959.435781693166, 18093.70564938978, 16435.61461383947, 14947.197437326102, 13605.954880888436, 12393.880316263208, 11293.27810727986, 10292.698091923068, 9379.934609293405, 8542.778732654882, 7780.101399774407, 7083.3906599663305, 6442.360608787293, 5856.924952855579, 5319.662670742758, 4827.494923733251, 4375.152951451802, 3958.746627662967, 3574.2727718396727, 3216.389789008766, 2880.8040627817663, 2563.2893900928902, 2259.124250598867, 1963.4009524512699, 1683.301052960261, 1421.08925599032, 1181.4882875552755, 972.8920812023131, 794.3919410633861, 643.6129305537779], "phase2": [122.02590714978953, -861.8691127743712, -1142.8551043268158, -1248.3343954963652, -1319.0632400945233, -1375.485088640635, -1415.3387799226973, -1398.7259993571608, -1406.2506096944428, -1425.3786072098467, -1385.4821177117121, -1148.7904243974, -1248.4710558849933, -1302.58240646708, -1422.9290660653176, -1296.0532055882159, -1432.8777691683824, -1443.8657069101057, -1421.0467725977735, -1411.19388568273, -1427.8889874691674, -1379.492333117903, -1356.5797617962307, -1358.5256191991677, -1405.5467914984783, -1409.6484860247688, -1368.1521038967614, -1351.3528504368003, -1413.816013007459, -1426.6550440932342, -1356.5267725452202, -1425.2884165221458, -1420.5483351285052, -1430.2946033617723, -1380.3330104443605, -1399.0665992260174, -1360.9939473244767, -1419.1421503464217, -1415.0248356293594, -1398.6618957762776, -1402.3061839927834, -1425.2654433536431, -1384.815978968837, -1400.3690408109871, -1402.4205821010662, -1412.2783526889364, -1391.0496208478644, -1390.1175558545679, -1389.4460105298315, -1403.3841449208112]},
        "survey_kats": {"gp_loglik": -18.814420362103, "svgp_elbo": -32.725635407458,
                        "sgp_bound": -20.731336414403,
                        "svgp_pred_mu": [0.13216128, 0.01463357, 0.05120182, 0.42671159, 0.25158748],
                        "svgp_pred_var": [0.84348966, 0.92821441, 0.83251444, 0.46197432, 1.3093171],
                        "gp_pred_mu0": [0.0542666, 0.03815007], "gp_pred_var0": 0.1910888706,
                        "kernel_spot": {"rbf": 0.911064218532, "matern52": 0.875434653540,
                                        "matern32": 0.841188624781, "matern12": 0.670313500731}},
    }
    with open(os.path.join(HERE, 'reference_recorded.json'), 'w') as f:
        json.dump(rec, f, indent=1)


def svgp_notebook_setup(seed=2):
    """Data, inducing inputs and run protocol of examples/notebooks/svgp_regression.ipynb cells 4, 9, 11.  Reproducible from NumPy: the
    data (np.random.seed(0); uniform; randn), the default inducing inputs drawn by SVGPRegression.__init__ (svgp_regression.py:317-320)
    and the randn(20, 1) cell 11 assigns.  NOT reproducible (MXNet RNG): the initial qU_mean / qU_cov_W / raw qU_cov_diag -- MXNet's
    default initialiser Uniform(0.07) (inference_parameters.py:81-88 passes init=None) -- and the DataLoader shuffles; both are drawn
    here from RandomState(seed) and injected through the product's seams."""
    np.random.seed(0)
    X = np.random.uniform(-3., 3., (1000, 1))
    Y = np.sin(X) + np.random.randn(1000, 1) * 0.05
    _ = np.random.randn(20, 1)
    Z0 = np.random.randn(20, 1)
    r = np.random.RandomState(seed)
    U = lambda *sh: r.uniform(-0.07, 0.07, sh)
    init = {'Z': Z0, 'noise_var': np.array([0.01]), 'lengthscale': np.array([1.0]), 'variance': np.array([1.0]),
            'qU_mean': U(20, 1), 'qU_cov_W': U(20, 20), 'qU_cov_diag_raw': U(20)}
    perms = [r.permutation(1000) for _ in range(100)]
    return X, Y, init, perms


def svgp_notebook_raw0(init):
    return {'Z': T(init['Z']), 'noise_var': O.inv_softplus(T(init['noise_var'])), 'lengthscale': O.inv_softplus(T(init['lengthscale'])),
            'variance': O.inv_softplus(T(init['variance'])), 'qU_mean': T(init['qU_mean']), 'qU_cov_W': T(init['qU_cov_W']),
            'qU_cov_diag': T(init['qU_cov_diag_raw'])}


def svgp_notebook():
    """The oracle's run of the notebook protocol (about half a minute on one core): final parameters + per-epoch losses."""
    X, Y, init, perms = svgp_notebook_setup()
    raw, epoch_losses = O.run_svgp_notebook(T(X), T(Y), svgp_notebook_raw0(init), perms)
    out = {k: v.numpy() for k, v in raw.items()}
    out['epoch_losses'] = np.asarray(epoch_losses)
    out['variance'] = O.softplus(raw['variance']).numpy()
    out['lengthscale'] = O.softplus(raw['lengthscale']).numpy()
    out['noise'] = O.softplus(raw['noise_var']).numpy()
    out['full_batch_loss'] = float(O.map_svgp_loss(O.RBF(1), T(X), T(Y), raw, jitter=1e-6))
    save('svgp_notebook_oracle', **out)


if __name__ == '__main__':
    kat_gp()
    kat_svgp()
    kat_sgp()
    kat_kernels()
    kat_svi()
    svgp_notebook()
    reference_recorded()
    print('golden fixtures written to', HERE)
