"""Gram kernel time at N=65536, Q=8 per kind and dtype (HIP events).  usage: gram_time.py [f32|f64]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
N, Q = 65536, 8
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.float64
X = torch.rand(1, N, Q, device='cuda', dtype=dt) * 6 - 3
ls = torch.ones(1, Q, device='cuda', dtype=dt); var = torch.ones(1, 1, device='cuda', dtype=dt)
out = torch.empty(1, N, N, device='cuda', dtype=dt)
for kind in ('rbf', 'matern12', 'matern32', 'matern52'):
    for _ in range(2):
        ops.gram(kind, X, None, ls, var, True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gram(kind, X, None, ls, var, True, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%s %s: %.3f ms  %.2f TB/s' % (kind, sys.argv[1] if len(sys.argv) > 1 else 'f32', ms, N * N * out.element_size() / ms / 1e9))
