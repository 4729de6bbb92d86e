"""float32 SVGP training call against float64 on the same inputs when the inputs are NOT centred (X, Z = offset + U(-2, 2)): the matrix-pipe
reverse pass forms r2 from |x|^2 + |z|^2 - 2 x.z.  usage: offset_accuracy.py"""
import os
import sys
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
warnings.simplefilter('ignore')
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
for kind in ('rbf', 'matern32'):
    for off in (0., 10., 100., 1000.):
        rng = np.random.RandomState(5)
        B, M, Q = 4096, 256, 5
        X = r32(off + rng.uniform(-2., 2., (1, B, Q)))
        Y = r32(np.sin((X[0] - off) @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1)))[None]
        Z = r32(off + rng.uniform(-2., 2., (M, Q)))
        qm, qW, qd = r32(0.3 * rng.standard_normal((M, 1))), r32(0.3 * rng.standard_normal((M, M)) / np.sqrt(M)), r32(rng.uniform(0.05, 0.5, M))
        ls, var, noise = r32(np.full(Q, 0.3 * np.sqrt(Q))), r32([1.1]), r32([0.05])
        out = {}
        for dt in (torch.float32, torch.float64):
            d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
            r = ops.svgp_logpdf(kind, d(X), d(Y), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, gscale=1.0, want_grad=True)
            torch.cuda.synchronize()
            out[dt] = {k: v.double().cpu().numpy() for k, v in r.items()}
        o, ref = out[torch.float32], out[torch.float64]
        errs = sorted(((nrm(o[k], ref[k]), k) for k in ref if k.startswith('d')), reverse=True)[:3]
        print('%-9s offset %6.0f  value %.1e  worst grads %s' % (kind, off, abs(o['logL'][0] - ref['logL'][0]) / abs(ref['logL'][0]), ' '.join('%s %.1e' % (k, e) for e, k in errs)), flush=True)
