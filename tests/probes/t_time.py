"""Time of the T-shaped split GEMM (1024 x SB x 1024, output in 16-column blocks: the persistent 128 x 256 kernel) and of the Psi2 shape.
usage: t_time.py [SB]      (MXF_GP_LIB selects an experiment build)"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
M, SB = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
pa = ops.f16x2_split(torch.randn(M, M, device='cuda'))
B = torch.rand(SB, M, device='cuda'); pb = ops.f16x2_split(B); del B
out = torch.empty(M, SB, device='cuda')
C = torch.rand(M, SB, device='cuda'); pc = ops.f16x2_split(C); del C
psi = torch.zeros(M, M, device='cuda')
for name, fn in (('T', lambda: ops.gemm_f16x2_planes(pa, pb, M, SB, M, out=out, blocked=True)),
                 ('Psi2', lambda: ops.gemm_f16x2_planes(pc, pc, M, M, SB, out=psi, lower_only=True))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.3f ms' % (name, e0.elapsed_time(e1) / 5))
