import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mxfusion_amd import ops
A = torch.rand(1024, 1024, device='cuda') - 0.3
B = torch.rand(262144, 1024, device='cuda')
out = torch.empty(1024, 262144, device='cuda')
for _ in range(3):
    ops.gemm_f32x3(A, B, out=out)
torch.cuda.synchronize()
