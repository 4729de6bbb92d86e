"""Time of the M x M float64 products of the SVGP core (small-tile GEMM).  usage: gemm_small_time.py [n]"""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from mxfusion_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = torch.randn(1, n, n, device='cuda', dtype=torch.float64); B = torch.randn(1, n, n, device='cuda', dtype=torch.float64)
out = torch.empty(1, n, n, device='cuda', dtype=torch.float64)
for ta, tb in ((False, False), (False, True), (True, False), (True, True)):
    for _ in range(3):
        ops.gemm(A, B, transA=ta, transB=tb, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(A, B, transA=ta, transB=tb, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('n=%d transA=%d transB=%d: %.1f us  %.1f TF' % (n, ta, tb, ms * 1e3, 2.0 * n ** 3 / ms / 1e9))
