"""Per-row noise (N, 1) SVGP training call at the bench shape: the streaming form (r04) against the generic materialised-dKuf path
(probe build, MXF_SVGP_HET_STREAM=0) and the homoscedastic call.  usage: [MXF_GP_LIB=...probe.so MXF_SVGP_HET_STREAM=0] het_time.py [S]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, M, Q = 65536, 1024, 8
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.rand(S, B, Q, device='cuda', generator=g) * 6 - 3
Y = torch.sin(X[0] @ torch.randn(Q, 1, device='cuda', generator=g))[None]
Z = X[0, :M].contiguous()
qm, qW, qd = torch.zeros(M, 1, device='cuda'), torch.zeros(M, M, device='cuda'), torch.ones(M, device='cuda')
ls, var = torch.ones(Q, device='cuda'), torch.ones(1, device='cuda')
for name, noise in (('homoscedastic (1,)', torch.full((1,), 0.02, device='cuda')), ('per-row (N, 1)', torch.rand(B, 1, device='cuda', generator=g) * 0.05 + 0.01)):
    fn = lambda: ops.svgp_logpdf('rbf', X, Y, Z, noise, qm, qW, qd, ls, var, True, jitter=1e-6, gscale=1.0 / S, want_grad=True)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print('%-20s S=%d: %.2f ms per training call' % (name, S, e0.elapsed_time(e1) / 5), flush=True)
