"""One-kernel workload for the PMC passes: RBF Gram N=65536, Q=8, float32 (3 launches)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
N, Q = 65536, 8
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.float64
X = torch.rand(1, N, Q, device='cuda', dtype=dt) * 6 - 3
ls = torch.ones(1, Q, device='cuda', dtype=dt); var = torch.ones(1, 1, device='cuda', dtype=dt)
out = torch.empty(1, N, N, device='cuda', dtype=dt)
for _ in range(3):
    ops.gram('rbf', X, None, ls, var, True, out=out)
torch.cuda.synchronize()
