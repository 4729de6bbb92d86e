"""The SVGP training call (float32, N = 65 536, M = 1 024, Q = 8, S samples) per kernel kind: the Matern kinds run the difference-form reverse pass
since r04 (MXF_BWD_MFMA=2 in the probe build puts them back on the matrix-pipe pass).  usage: matern_step_time.py [S]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, M, Q = 65536, 1024, 8
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.rand(S, B, Q, device='cuda', generator=g) * 6 - 3
Y = torch.sin(X[0].sum(-1, keepdim=True))[None]
Z = X[0, :M].clone()
qm, qW, qd = torch.zeros(M, 1, device='cuda'), torch.zeros(M, M, device='cuda'), torch.ones(M, device='cuda')
ls, var, noise = torch.ones(Q, device='cuda'), torch.ones(1, device='cuda'), torch.full((1,), 0.01, device='cuda')
for kind in ('rbf', 'matern52', 'matern32', 'matern12'):
    f = lambda: ops.svgp_logpdf(kind, X, Y, Z, noise, qm, qW, qd, ls, var, True, jitter=1e-6, gscale=1.0 / S, want_grad=True)
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    print('%-9s S=%d  %.2f ms per call' % (kind, S, e0.elapsed_time(e1) / 5), flush=True)
