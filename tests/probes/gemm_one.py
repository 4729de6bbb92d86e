import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
M, N, K = 1024, 262144, 1024
A = torch.randn(1, M, K, device='cuda', dtype=torch.float32); B = torch.randn(1, K, N, device='cuda', dtype=torch.float32)
out = torch.empty(1, M, N, device='cuda', dtype=torch.float32)
for _ in range(3): ops.gemm(A, B, out=out)
torch.cuda.synchronize()
