"""The trailing update of the blocked Cholesky alone: C (n x n, lower tiles) -= A A^T with A (n x 512), float64 (MXF_GEMM_LOWER=1 makes mxf_gemm
skip the tiles above the diagonal).  usage: MXF_GEMM_LOWER=1 python syrk_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
for n, K in ((7168, 512), (6144, 512), (4096, 512), (2048, 512), (7168, 1024), (7168, 2048)):
    A = torch.randn(1, n, K, device='cuda', dtype=torch.float64)
    C = torch.zeros(1, n, n, device='cuda', dtype=torch.float64)
    for _ in range(2):
        ops.gemm(A, A, transB=True, alpha=-1.0, beta=1.0, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(A, A, transB=True, alpha=-1.0, beta=1.0, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = n * (n + 128) / 2 * K * 2
    print('n=%d K=%d: %.3f ms, %.1f TFLOP/s (lower tiles)' % (n, K, ms, fl / ms / 1e9))
