"""Accuracy of the split (bf16-plane) SVGP training step at the bench size vs the float64 step, for MXF_SPLIT_NPROD = 6 / 3.
usage: nprod_probe.py {f64|f32} out.npz ; then nprod_probe.py cmp a.npz b.npz"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
if sys.argv[1] == 'cmp':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        print('%-8s normwise %.2e   max-elt/max %.2e' % (k, np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-300), np.abs(x - y).max() / max(np.abs(x).max(), 1e-300)))
    sys.exit(0)
import torch
from mxfusion_amd import ops
N, Q, M, S, P = 65536, 8, 1024, 4, 1
rng = np.random.default_rng(0)
X0 = rng.uniform(-3., 3., (N, Q))
w = rng.standard_normal(Q)
Y = np.sin(X0 @ w)[:, None] + 0.05 * rng.standard_normal((N, 1))
Z = X0[rng.permutation(N)[:M]].copy()
X = X0[None] + 0.1 * rng.standard_normal((S, N, Q))
qm, qW, qd = rng.standard_normal((M, P)) * 0.3, rng.standard_normal((M, M)) * 0.02, rng.random(M) + 0.5
ls, var, noise = np.ones(Q) * 1.0 + 0.2 * rng.random(Q), np.array([1.2]), np.array([0.02])
dt = torch.float64 if sys.argv[1] == 'f64' else torch.float32
d = lambda a: torch.as_tensor(a, dtype=dt).cuda()
args = (d(X), d(Y[None]), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    r = ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
    torch.cuda.synchronize(); t1 = time.time()
print(sys.argv[1], 'NPROD', os.environ.get('MXF_SPLIT_NPROD'), 'step %.2f ms' % ((t1 - t0) * 1e3), 'info', int(r['info'].abs().sum()))
np.savez(sys.argv[2], **{k: v.double().cpu().numpy() for k, v in r.items() if k != 'info'})
