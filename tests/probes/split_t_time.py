"""Time of the T-shaped split GEMM (1024 x SB x 1024) in the f16x2 format.  usage: split_t_time.py [SB] [K]"""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from mxfusion_amd import ops
M = 1024
SB = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
K = int(sys.argv[2]) if len(sys.argv) > 2 else M
A = torch.randn(M, K, device='cuda'); B = torch.rand(SB, K, device='cuda')
pa, pb = ops.f16x2_split(A), ops.f16x2_split(B)
del B
out = torch.empty(M, SB, device='cuda')
for _ in range(2):
    ops.gemm_f16x2_planes(pa, pb, M, SB, K, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.gemm_f16x2_planes(pa, pb, M, SB, K, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print('T GEMM %d x %d x %d: %.2f ms, %.0f TF f32-equivalent, NPROD=%s' % (M, SB, K, ms, 2.0 * M * K * SB / ms / 1e9, os.environ.get('MXF_SPLIT_NPROD')))
