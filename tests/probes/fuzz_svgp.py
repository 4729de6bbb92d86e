"""Randomised shape sweep of the SVGP module's training call: float32 (whatever path the shape takes: split / padded / generic / whitened / float64
by the guard) against the same call in float64, and float64 against the oracle for the small shapes.  usage: fuzz_svgp.py [n] [seed]"""
import os
import sys
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as O
from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern12, Matern32, Matern52
from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
from mxfusion_amd.modules.gp_modules._fused import Float32Guard

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KINDS = {'rbf': (RBF, O.RBF), 'matern12': (Matern12, O.Matern12), 'matern32': (Matern32, O.Matern32), 'matern52': (Matern52, O.Matern52)}
worst = {}
warnings.simplefilter('ignore')
for it in range(n):
    kind = list(KINDS)[rng.randint(4)]
    S = [1, 1, 2, 3][rng.randint(4)]
    B = int(rng.choice([37, 256, 300, 1000, 1024, 2049, 4096]))
    M = int(rng.choice([7, 64, 100, 128, 130, 200, 256, 384, 500]))
    Q = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 20]))
    P = int(rng.choice([1, 1, 2, 3, 8]))          # (wider Y is blocked in passes of 8 one level up, compute())
    ard = bool(rng.randint(2))
    sampled = S > 1 and bool(rng.randint(2))
    ell = float(rng.choice([0.5, 1.0, 2.0])) * np.sqrt(Q)
    off = float(rng.choice([0., 0., 50., 3000.]))            # inputs at an offset (raw time stamps ...): nothing may depend on it
    X = off + rng.uniform(-2., 2., (S, B, Q))
    Y = np.sin((X[0] - off) @ rng.standard_normal((Q, P))) + 0.05 * rng.standard_normal((B, P))
    Z = off + rng.uniform(-2., 2., (S if sampled else 1, M, Q))
    qm, qW, qd = 0.3 * rng.standard_normal((M, P)), 0.3 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
    ls = rng.uniform(0.8, 1.2, (S if sampled else 1, Q if ard else 1)) * ell
    var, noise = rng.uniform(0.9, 1.3, (S if sampled else 1, 1)), np.array([[0.05]])
    het = (not sampled) and rng.randint(4) == 0          # per-row (P = 1) or per-element noise
    if het:
        noise = rng.uniform(0.02, 0.2, (1, B, 1 if (P == 1 or rng.randint(2)) else P))
    r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)      # every run sees the SAME (float32-representable) inputs: arithmetic, not input rounding
    Y = r32(Y)
    vals = tuple((k, r32(v)) for k, v in (('X', X), ('Z', Z), ('noise', noise), ('qm', qm[None]), ('qW', qW[None]), ('qd', qd[None]), ('ls', ls), ('var', var)))
    scal = float(rng.choice([1.0, 1.0, 8.0]))               # log_pdf_scaling (minibatch steps)
    res = {}
    for dt in (torch.float32, torch.float64):
        kern = KINDS[kind][0](input_dim=Q, ARD=ard, dtype='float32' if dt == torch.float32 else 'float64')
        fn = SVGPRegressionLogPdf.__new__(SVGPRegressionLogPdf)
        fn.jitter, fn.log_pdf_scaling, fn._guard = 1e-6, scal, Float32Guard('fuzz%d' % it)
        lv = {k: torch.as_tensor(v, dtype=dt).cuda().requires_grad_(True) for k, v in vals}
        out = fn._compute_columns(None, lv['X'], torch.as_tensor(Y, dtype=dt).cuda()[None], lv['Z'], lv['noise'], lv['qm'], lv['qW'], lv['qd'], kern,
                                  {kern.name + '_lengthscale': lv['ls'], kern.name + '_variance': lv['var']})
        g = torch.autograd.grad(out.mean(), list(lv.values()))
        torch.cuda.synchronize()
        fn._guard.poll(torch.device('cuda', torch.cuda.current_device()))
        res[dt] = (out.detach().double().cpu().numpy(), [x.double().cpu().numpy() for x in g], fn._guard.tier, fn._guard.cond_max)
    v32, g32, tier, _c = res[torch.float32]
    v64, g64, _, _c2 = res[torch.float64]
    ev = float(np.abs(v32 - v64).max() / np.abs(v64).max())
    eg = max(float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)) for a, b in zip(g32, g64))
    cond = res[torch.float32][3]
    tag = '%s S%d B%d M%d Q%d P%d ard%d smp%d tier%s cond %.1e off %g het%d' % (kind, S, B, M, Q, P, ard, sampled, tier, cond, off, het)
    eo = None
    if B * M <= 300000:
        ok = KINDS[kind][1](Q, ARD=ard)
        lo = {k: O.T(v - off if k in ('X', 'Z') else v) for k, v in vals}
        ref = O.svgp_log_pdf(ok, lo['X'], O.T(Y)[None], lo['Z'], lo['noise'], lo['qm'], lo['qW'], lo['qd'],
                             {ok.name + '_lengthscale': lo['ls'], ok.name + '_variance': lo['var']}, jitter=1e-6, log_pdf_scaling=scal).numpy()
        eo = float(np.abs(v64 - ref).max() / np.abs(ref).max())
    bad = (not np.isfinite(ev)) or ev > 2e-5 or eg > 5e-3 or (eo is not None and eo > max(1e-9, 1e-14 * cond)) or not all(np.isfinite(x).all() for x in g32)
    print('%s  f32-f64 value %.1e grad %.1e  f64-oracle %s %s' % (tag, ev, eg, ('%.1e' % eo) if eo is not None else '-', 'BAD' if bad else ''), flush=True)
