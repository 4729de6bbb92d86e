import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from mxfusion_amd import ops
torch.manual_seed(0)
for dt in (torch.float64, torch.float32):
    for n in (64, 128, 448, 449, 511, 512, 513, 576, 1024, 1536):
        A = torch.randn(n, n, device='cuda', dtype=torch.float64)
        K = (A @ A.T / n + torch.eye(n, device='cuda', dtype=torch.float64)).to(dt)
        L, info = ops.potrf_(K[None].clone())
        ref = torch.linalg.cholesky(K.double())
        err = float((L[0].double() - ref).abs().max())
        print(dt, n, 'info', info.cpu().tolist(), 'err %.2e' % err, flush=True)
# the lattice RBF Kuu of the deep GP workload
side, M, Dh = 23, 512, 2
g = np.stack(np.meshgrid(*[np.linspace(-1.2, 1.2, side)] * Dh, indexing='ij'), -1).reshape(-1, Dh)[:M]
Z = torch.as_tensor(g + 0.01 * np.random.default_rng(3).standard_normal((M, Dh)), device='cuda')[None]
ls = torch.full((1, Dh), 2.4 / 22, device='cuda', dtype=torch.float64); var = torch.ones(1, 1, device='cuda', dtype=torch.float64)
K = ops.gram('rbf', Z, None, ls, var, True, jitter=1e-5)
ev = torch.linalg.eigvalsh(K[0])
print('lattice Kuu eig min %.3e max %.3e' % (float(ev.min()), float(ev.max())))
L, info = ops.potrf_(K.clone()); print('info', info.cpu().tolist())
