"""Device time of ONE training call of the library (mxf_svgp_logpdf with the reverse mode) at S samples, back to back without host
synchronisation in between -- the step without the host layer's head and tail.  usage: svgp_call_time.py [S]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
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
run = lambda: ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
for it in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print('S=%d: library training call %.3f ms (20 back to back)' % (S, e0.elapsed_time(e1) / 20))
