"""One SVGP training call (S samples) captured into a hipGraph and replayed: with the host out of the way a rocprofv3 kernel trace of the
replays shows the device-side critical path.  usage: graph_timeline.py [S]"""
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
call = lambda: ops.svgp_logpdf('rbf', *args, jitter=1e-6, scaling=1.0, gscale=1.0 / S, want_grad=True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    call(); call()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = call()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print('S=%d graph replay %.3f ms per call' % (S, (time.perf_counter() - t0) / 10 * 1e3))
