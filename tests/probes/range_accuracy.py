"""float32 SVGP training call (RBF, matrix-pipe reverse pass) against float64 as the inputs span more and more length-scales around the
inducing inputs (Q-dimensional box of half-width R length-scales / sqrt(Q)).  usage: range_accuracy.py"""
import os
import sys
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import _lib, ops
warnings.simplefilter('ignore')
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
dev = torch.cuda.current_device()
for Q in (1, 2, 8):
    for R in (5, 10, 20, 40, 80, 160):
        rng = np.random.RandomState(5)
        B, M = 4096, 256
        h = R / np.sqrt(Q)
        X = r32(rng.uniform(-h, h, (1, B, Q)))
        Y = r32(np.sin(X[0].sum(-1, keepdims=True) / 3.0) + 0.05 * rng.standard_normal((B, 1)))[None]
        Z = r32(X[0, rng.permutation(B)[:M]] + 0.3 * rng.standard_normal((M, Q)))
        qm, qW, qd = r32(0.3 * rng.standard_normal((M, 1))), r32(0.1 * rng.standard_normal((M, M)) / np.sqrt(M)), r32(rng.uniform(0.05, 0.5, M))
        ls, var, noise = r32(np.ones(Q)), r32([1.1]), r32([0.05])
        out = {}
        for key, dt, form in (('exp', torch.float32, _lib.FORM_EXPLICIT), ('whi', torch.float32, _lib.FORM_WHITENED), ('f64', torch.float64, _lib.FORM_EXPLICIT)):
            d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
            _lib.svgp_configure(dev, form, 9)
            try:
                r = ops.svgp_logpdf('rbf', d(X), d(Y), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, gscale=1.0, want_grad=True)
                torch.cuda.synchronize()
            finally:
                _lib.svgp_configure(dev, _lib.FORM_EXPLICIT, 0)
            out[key] = {k: v.double().cpu().numpy() for k, v in r.items()}
        cond = _lib.svgp_cond_slot(dev, 9, reset=True)[0]
        ref = out['f64']
        line = 'Q%d radius %4d cond %.0e ' % (Q, R, cond)
        for key in ('exp', 'whi'):
            o = out[key]
            eg, kg = max((nrm(o[k], ref[k]), k) for k in ref if k.startswith('d'))
            line += ' %s value %.1e grad %.1e(%s)' % (key, abs(o['logL'][0] - ref['logL'][0]) / abs(ref['logL'][0]), eg, kg)
        print(line, flush=True)
