import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from mxfusion_amd import ops
side, M, Dh = 23, 512, 2
g = np.stack(np.meshgrid(*[np.linspace(-1.2, 1.2, side)] * Dh, indexing='ij'), -1).reshape(-1, Dh)[:M]
for dt in (torch.float32,):
    Z = torch.as_tensor(g + 0.01 * np.random.default_rng(3).standard_normal((M, Dh)), device='cuda', dtype=dt)
    ls = torch.full((Dh,), 2.4 / 22, device='cuda', dtype=dt); var = torch.ones(1, device='cuda', dtype=dt)
    noise = torch.full((1,), 0.01, device='cuda', dtype=dt); mu = torch.zeros(M, 1, device='cuda', dtype=dt)
    W = torch.zeros(M, M, device='cuda', dtype=dt); sd = torch.ones(M, device='cuda', dtype=dt)
    for S, B in ((4, 131072), (4, 131072), (8, 65536), (4, 131088), (3, 131072)):
        X = torch.tanh(torch.randn(S, B, Dh, device='cuda', dtype=dt)); Y = torch.randn(1, B, 1, device='cuda', dtype=dt)
        for wg in (True,):
            r = ops.svgp_logpdf('rbf', X, Y, Z, noise, mu, W, sd, ls, var, True, jitter=1e-5, gscale=1.0 / S, want_grad=wg)
            print(dt, S, B, 'grad' if wg else 'fwd ', 'info', r['info'].cpu().tolist(), 'logL', [float(v) for v in r['logL'][:2]], flush=True)
