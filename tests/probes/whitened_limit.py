"""Whitened float32 tier against float64 on the same (float32-representable) inputs as cond_1(Kuu) grows, for small / low-dimensional problems
(many inducing points on a line): which condition number still holds 1e-5 on the bound?  usage: whitened_limit.py"""
import os
import sys
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import _lib, ops
warnings.simplefilter('ignore')
dev = torch.cuda.current_device()
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
for kind, Q, M, B in (('matern12', 1, 256, 256), ('matern12', 1, 256, 4096), ('matern32', 1, 256, 1024), ('rbf', 2, 256, 1024), ('rbf', 8, 256, 4096), ('matern52', 2, 128, 512)):
    for ell in (0.05, 0.1, 0.2, 0.4, 0.8, 1.6, 3.2):
        rng = np.random.RandomState(3)
        X = r32(rng.uniform(-2., 2., (1, B, Q)))
        Y = r32(np.sin(X[0] @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1)))[None]
        Z = r32(rng.uniform(-2., 2., (M, Q)))
        qm, qW, qd = r32(0.3 * rng.standard_normal((M, 1))), r32(0.3 * rng.standard_normal((M, M)) / np.sqrt(M)), r32(rng.uniform(0.05, 0.5, M))
        ls, var, noise = r32(np.full(Q, ell * np.sqrt(Q))), r32([1.1]), r32([0.05])
        out = {}
        for dt, form in ((torch.float32, _lib.FORM_WHITENED), (torch.float32, _lib.FORM_EXPLICIT), (torch.float64, _lib.FORM_EXPLICIT)):
            d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
            _lib.svgp_configure(dev, form, 7)
            try:
                r = ops.svgp_logpdf(kind, d(X), d(Y), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, gscale=1.0, want_grad=True)
                torch.cuda.synchronize()
            finally:
                _lib.svgp_configure(dev, _lib.FORM_EXPLICIT, 0)
            out[(dt, form)] = {k: v.double().cpu().numpy() for k, v in r.items()}
        cond = _lib.svgp_cond_slot(dev, 7, reset=True)[0]
        ref = out[(torch.float64, _lib.FORM_EXPLICIT)]
        nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
        res = []
        for form in (_lib.FORM_WHITENED, _lib.FORM_EXPLICIT):
            o = out[(torch.float32, form)]
            ev = abs(o['logL'][0] - ref['logL'][0]) / abs(ref['logL'][0])
            eg, kg = max((nrm(o[k], ref[k]), k) for k in ref if k.startswith('d'))
            res.append('%.1e/%.1e(%s)' % (ev, eg, kg))
        print('%-9s Q%d M%d B%-5d ell %.2f cond %.1e   whitened value/grad %s   explicit %s' % (kind, Q, M, B, ell, cond, res[0], res[1]), flush=True)
