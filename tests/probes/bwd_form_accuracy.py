"""float32 gradients of the SVGP training call against float64 on the same inputs, per kernel kind and input dimension: the matrix-pipe
reverse pass (expansion-form distances, MXF_BWD_MFMA=1) against the difference-form pass (MXF_BWD_MFMA=0; probe build).  usage: bwd_form_accuracy.py"""
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
for kind in ('rbf', 'matern52', 'matern32', 'matern12'):
    for Q in (1, 2, 3, 5, 8):
        rng = np.random.RandomState(5)
        B, M = 4096, 256
        X = r32(rng.uniform(-2., 2., (1, B, Q)))
        Y = r32(np.sin(X[0] @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1)))[None]
        Z = r32(X[0, rng.permutation(B)[:M]] + 1e-3 * rng.standard_normal((M, Q)))        # inducing inputs NEAR data points, as after a few steps from Z = X[:M]
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
        print('%-9s Q%d  value %.1e  worst grads %s' % (kind, Q, abs(o['logL'][0] - ref['logL'][0]) / abs(ref['logL'][0]), ' '.join('%s %.1e' % (k, e) for e, k in errs)), flush=True)
