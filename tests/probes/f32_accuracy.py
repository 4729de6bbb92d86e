"""float32 streaming SVGP step vs float64 on the same inputs, as a function of the conditioning of Kuu (length-scale sweep).
usage: f32_accuracy.py [B] [M]   -- prints ELBO and gradient agreement per length-scale and the condition number of Kuu + jitter I."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S, Q, P = 2, 8, 1
rng = np.random.default_rng(0)
X0 = rng.uniform(-3., 3., (B, Q))
w = rng.standard_normal(Q)
Y = np.sin(X0 @ w)[:, None] + 0.05 * rng.standard_normal((B, 1))
Z = X0[rng.permutation(B)[:M]].copy()
X = X0[None] + 0.1 * rng.standard_normal((S, B, Q))
qm = 0.3 * rng.standard_normal((M, P))
qW = 0.05 * rng.standard_normal((M, M)) / np.sqrt(M) * 8
qd = rng.uniform(0.05, 0.5, M)
noise = np.array([0.02])
var = np.array([1.0])


def run(dt, ls, jitter):
    d = lambda a: torch.as_tensor(np.asarray(a), dtype=dt).cuda()
    r = ops.svgp_logpdf('rbf', d(X), d(Y[None]), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=jitter, gscale=1.0 / S,
                        want_grad=True)
    torch.cuda.synchronize()
    return {k: v.double().cpu().numpy() for k, v in r.items()}


print('B=%d M=%d S=%d  (float32 streaming + f64 core vs float64 throughout)' % (B, M, S))
for jitter in (1e-6, 1e-4):
    for l in (1.0, 1.5, 2.2, 3.0, 4.4, 6.2):
        ls = np.full(Q, l)
        Zt = torch.as_tensor(Z) / l
        Kuu = torch.exp(-0.5 * torch.cdist(Zt, Zt) ** 2) + jitter * torch.eye(M, dtype=torch.float64)
        ev = torch.linalg.eigvalsh(Kuu)
        cond = float(ev[-1] / ev[0])
        r64 = run(torch.float64, ls, jitter)
        r32 = run(torch.float32, ls, jitter)
        rel = np.abs(r32['logL'] - r64['logL']).max() / np.abs(r64['logL']).max()
        line = 'jit %.0e l=%.1f cond(Kuu)=%.2e info=%d/%d ELBO %.6e rel %.2e |' % (jitter, l, cond, int(r64['info'].sum()), int(r32['info'].sum()),
                                                                                 r64['logL'][0], rel)
        for k in ('dX', 'dZ', 'dmu', 'dW', 'dSdiag', 'dls', 'dvar', 'dnoise'):
            a, b = r32[k].ravel(), r64[k].ravel()
            line += ' %s %.1e' % (k, np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        print(line, flush=True)
