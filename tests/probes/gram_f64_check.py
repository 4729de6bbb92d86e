import sys, os, torch
sys.path.insert(0, os.getcwd())
from mxfusion_amd import ops
torch.manual_seed(0)
for (N, N2, Q) in ((1000, 1000, 8), (513, 770, 3), (4096, 4096, 8), (300, 2, 1), (2048, 2048, 16)):
    X = torch.rand(2, N, Q, device='cuda', dtype=torch.float64) * 6 - 3
    Z = torch.rand(2, N2, Q, device='cuda', dtype=torch.float64) * 6 - 3
    ls = torch.rand(2, Q, device='cuda', dtype=torch.float64) + 0.5
    var = torch.rand(2, 1, device='cuda', dtype=torch.float64) + 0.5
    for sq in (False, True):
        K = ops.gram('rbf', X, None if sq else Z, ls, var, True)
        A = X / ls[:, None, :]; B = A if sq else Z / ls[:, None, :]
        ref = var[:, :, None] * torch.exp(-0.5 * torch.cdist(A, B) ** 2)
        print(N, N2, Q, sq, 'max abs err %.3e' % float((K - ref).abs().max()), 'diag err %.3e' % (float((torch.diagonal(K, dim1=1, dim2=2) - var).abs().max()) if sq else 0.0))
# far apart points: underflow region
X = torch.zeros(1, 64, 8, device='cuda', dtype=torch.float64); X[0, :, 0] = torch.arange(64, device='cuda') * 3.0
ls = torch.ones(1, 8, device='cuda', dtype=torch.float64); var = torch.ones(1, 1, device='cuda', dtype=torch.float64)
K = ops.gram('rbf', X, None, ls, var, True)
ref = torch.exp(-0.5 * torch.cdist(X, X) ** 2)
rel = ((K - ref).abs() / ref.clamp_min(1e-300)).max()
print('underflow sweep: max rel err %.3e' % float(rel), 'min nonzero', float(K[K > 0].min()), float(ref[ref > 0].min()))
