"""Randomised shape sweep of the exact-GP and sparse-GP training calls and of the three prediction entry points: float64 against the oracle, float32
against float64 on the same (float32-representable) inputs.  usage: fuzz_gp.py [n] [seed]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as O
from mxfusion_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KINDS = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
nrm = lambda a, b: float(np.linalg.norm(np.asarray(a).ravel() - np.asarray(b).ravel()) / max(np.linalg.norm(np.asarray(b).ravel()), 1e-300))
T = O.T
for it in range(n):
    kind = list(KINDS)[rng.randint(4)]
    N = int(rng.choice([5, 33, 64, 100, 129, 256, 500, 700]))
    M = int(rng.choice([3, 16, 50, 64, 130]))
    Nt = int(rng.choice([1, 7, 64, 130]))
    Q = int(rng.choice([1, 2, 3, 5, 8, 12, 17]))
    P = int(rng.choice([1, 2, 3, 9]))
    S = [1, 1, 2, 3][rng.randint(4)]
    ard = bool(rng.randint(2))
    X = r32(rng.uniform(-2., 2., (S, N, Q)))
    Y = r32(np.sin(X[0] @ rng.standard_normal((Q, P))) + 0.05 * rng.standard_normal((N, P)))[None]
    Z = r32(rng.uniform(-2., 2., (M, Q)))
    Xt = r32(rng.uniform(-2., 2., (S, Nt, Q)))
    ls = r32(rng.uniform(0.8, 1.2, (1, Q if ard else 1)) * np.sqrt(Q))
    var, noise = r32(rng.uniform(0.9, 1.3, (1, 1))), r32([[0.05]])
    ok = KINDS[kind](Q, ARD=ard)
    kp = {ok.name + '_lengthscale': T(ls), ok.name + '_variance': T(var)}
    tag = '%s N%d M%d Nt%d Q%d P%d S%d ard%d' % (kind, N, M, Nt, Q, P, S, ard)
    msgs = []
    try:
        # ---- exact GP: log-pdf + gradients, then predictions from its posterior
        lead = {k: T(v).clone().requires_grad_(True) for k, v in (('X', X), ('Y', Y), ('noise', noise), ('ls', ls), ('var', var))}
        ref, post = O.gp_log_pdf(ok, lead['X'], lead['Y'], lead['noise'], {ok.name + '_lengthscale': lead['ls'], ok.name + '_variance': lead['var']},
                                 jitter=1e-6, return_posterior=True)
        gref = torch.autograd.grad(ref.mean(), list(lead.values()))
        res = {}
        for dt in (torch.float64, torch.float32):
            d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
            r = ops.gp_logpdf(kind, d(X), d(Y), d(noise), d(ls), d(var), ard, jitter=1e-6, want_grad=True)
            res[dt] = r
            assert int(r['info'].abs().sum()) == 0
        r64, r32_ = res[torch.float64], res[torch.float32]
        e = nrm(r64['logL'].cpu().numpy(), ref.detach().numpy())
        eg = max(nrm((r64[k] / S).cpu().numpy() if k != 'dY' else (r64[k].sum(0, keepdim=True) / S).cpu().numpy(), g.numpy())
                 for k, g in zip(('dX', 'dY', 'dnoise', 'dls', 'dvar'), gref) if k in ('dX',))
        if e > 1e-9 or eg > 1e-7: msgs.append('gp f64 %.1e/%.1e' % (e, eg))
        e32 = nrm(r32_['logL'].double().cpu().numpy(), r64['logL'].cpu().numpy())
        if e32 > 2e-4: msgs.append('gp f32 %.1e' % e32)
        if S == 1:
            Xc, L, LinvY = post
            for nf in (True, False):
                for full in (False, True):
                    mu_r, var_r = O.gp_predict(ok, T(Xt), T(noise), Xc[None], L[None], LinvY[None], kp, noise_free=nf, diagonal_variance=not full)
                    d = lambda a: torch.as_tensor(a, dtype=torch.float64).cuda()
                    mu, vv = ops.gp_predict(kind, d(X[0]), d(Xt), d(ls[0]), d(var[0]), ard, r64['L'][0], r64['LinvY'][0], d(noise[0]), noise_free=nf, full_cov=full)
                    em, ev = nrm(mu.cpu().numpy(), mu_r.numpy()), nrm(vv.cpu().numpy(), var_r.numpy())
                    if em > 1e-8 or ev > 1e-7: msgs.append('gp_predict nf%d full%d %.1e/%.1e' % (nf, full, em, ev))
        # ---- SVGP prediction (q(u) random)
        qm, qW, qd = r32(0.3 * rng.standard_normal((M, P))), r32(0.3 * rng.standard_normal((M, M)) / np.sqrt(M)), r32(rng.uniform(0.05, 0.5, M))
        for nf in (True, False):
            for full in (False, True):
                mu_r, var_r = O.svgp_predict(ok, T(Xt), T(Z)[None], T(noise), T(qm)[None], T(qW)[None], T(qd)[None], kp, jitter=1e-6, noise_free=nf,
                                             diagonal_variance=not full)
                for dt, tol in ((torch.float64, 1e-8), (torch.float32, 2e-4)):
                    d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
                    mu, vv, info = ops.svgp_predict(kind, d(Z), d(Xt), d(ls[0]), d(var[0]), ard, d(qm), d(qW), d(qd), d(noise[0]), jitter=1e-6, noise_free=nf, full_cov=full)
                    em, ev = nrm(mu.double().cpu().numpy(), mu_r.numpy()), nrm(vv.double().cpu().numpy().reshape(var_r.shape), var_r.numpy())
                    if em > tol or ev > tol * 10: msgs.append('svgp_predict %s nf%d full%d %.1e/%.1e' % ('f64' if dt == torch.float64 else 'f32', nf, full, em, ev))
        # ---- sparse GP (one sample)
        if P <= 8:
            lead = {k: T(v).clone().requires_grad_(True) for k, v in (('X', X[:1]), ('Z', Z[None]), ('noise', noise), ('ls', ls), ('var', var))}
            refs = O.sgp_log_pdf(ok, lead['X'], T(Y), lead['Z'], lead['noise'], {ok.name + '_lengthscale': lead['ls'], ok.name + '_variance': lead['var']}, jitter=1e-6)
            gs = torch.autograd.grad(refs.sum(), list(lead.values()))
            for dt, tol in ((torch.float64, 1e-8), (torch.float32, 5e-4)):
                d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
                r = ops.sgp_logpdf(kind, d(X[0]), d(Y[0]), d(Z), d(noise[0]), d(ls[0]), d(var[0]), ard, jitter=1e-6, want_grad=True)
                e = nrm(r['logL'].double().cpu().numpy(), refs.detach().numpy())
                egs = max(nrm(r[k].double().cpu().numpy(), g.numpy()) for k, g in zip(('dX', 'dZ', 'dnoise', 'dls', 'dvar'), gs))
                if e > tol or egs > tol * 100: msgs.append('sgp %s %.1e/%.1e' % ('f64' if dt == torch.float64 else 'f32', e, egs))
    except Exception as ex:
        msgs.append('EXC %s: %s' % (type(ex).__name__, str(ex)[:200]))
    print(tag, ' | '.join(msgs) if msgs else 'ok', flush=True)
