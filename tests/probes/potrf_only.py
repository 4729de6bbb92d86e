import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
n = 1024
A = torch.randn(1, n, n, device='cuda', dtype=torch.float64); K = A @ A.transpose(1, 2) / n + torch.eye(n, device='cuda', dtype=torch.float64)
work = K.clone()
for _ in range(3):
    work.copy_(K); ops.potrf_(work)
torch.cuda.synchronize()
