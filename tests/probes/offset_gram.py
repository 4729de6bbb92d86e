"""float32 Gram (mxf_gram), its reverse mode and the exact-GP call against float64 on the same inputs at an offset.  usage: offset_gram.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
nrm = lambda a, b: float(np.linalg.norm(a.ravel() - b.ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
for kind in ('rbf', 'matern32'):
    for off in (0., 100., 1000., 10000.):
        rng = np.random.RandomState(1)
        N, N2, Q = 512, 384, 5
        X, X2 = r32(off + rng.uniform(-2, 2, (1, N, Q))), r32(off + rng.uniform(-2, 2, (1, N2, Q)))
        ls, var = r32(np.full((1, Q), 1.3)), r32([[1.1]])
        G = rng.standard_normal((1, N, N2))
        Y = r32(np.sin((X[0] - off).sum(-1, keepdims=True)) + 0.05 * rng.standard_normal((N, 1)))[None]
        res = {}
        for dt in (torch.float32, torch.float64):
            d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
            K = ops.gram(kind, d(X), d(X2), d(ls), d(var), True)
            Ks = ops.gram(kind, d(X), None, d(ls), d(var), True)
            g = ops.gram_bwd(kind, d(X), d(X2), d(ls), d(var), True, d(G))
            gp = ops.gp_logpdf(kind, d(X), d(Y), d([[0.05]]), d(ls), d(var), True, jitter=1e-6, want_grad=True)
            res[dt] = dict(K=K, Ks=Ks, dX=g[0], dls=g[2], logL=gp['logL'], gdX=gp['dX'], gdls=gp['dls'])
        a, b = res[torch.float32], res[torch.float64]
        print('%-9s offset %6.0f  ' % (kind, off) + '  '.join('%s %.1e' % (k, nrm(a[k].double().cpu().numpy(), b[k].cpu().numpy())) for k in a), flush=True)
