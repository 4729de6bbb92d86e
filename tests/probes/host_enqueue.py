"""Host time to enqueue one SVGP training call vs its device time (is the step launch-bound?).  usage: host_enqueue.py [S]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from mxfusion_amd import ops
N, Q, M, P = 65536, 8, 1024, 1
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
X0 = rng.uniform(-3., 3., (N, Q)); w = rng.standard_normal(Q)
Y = np.sin(X0 @ w)[:, None] + 0.05 * rng.standard_normal((N, 1))
Z = X0[rng.permutation(N)[:M]].copy()
X = X0[None] + 0.1 * rng.standard_normal((S, N, Q))
d = lambda a: torch.as_tensor(a, dtype=torch.float32).cuda()
args = (d(X), d(Y[None]), d(Z), d([0.02]), d(rng.standard_normal((M, P)) * 0.3), d(rng.standard_normal((M, M)) * 0.02), d(rng.random(M) + 0.5), d(np.ones(Q)), d([1.2]), True)
for it in range(3):
    r = ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
torch.cuda.synchronize()
hs, ds = [], []
for it in range(10):
    t0 = time.perf_counter()
    r = ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(t1 - t0); ds.append(t2 - t0)
print('S=%d: host enqueue %.3f ms, call + device %.3f ms' % (S, np.median(hs) * 1e3, np.median(ds) * 1e3))
