"""potrf(n) on a matrix whose row stride is n + pad doubles (pad = 0: 64 KB rows at n = 8192, every row of a tile column in the same L2 / HBM
channel) -- is the power-of-two leading dimension what the rows-below kernel and the f64 products pay for?   usage: r06_lda_pad.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
X = torch.randn(n, 8, device='cuda', dtype=torch.float64)
K = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(n, device='cuda', dtype=torch.float64)
ref = torch.linalg.cholesky(K)
for pad in (0, 16, 32, 64, 528):
    bufs = [torch.zeros(1, n, n + pad, device='cuda', dtype=torch.float64) for _ in range(6)]
    for b in bufs:
        b[0, :, :n] = K
    v = bufs[0][:, :, :n]
    L, info = ops.potrf_(v)
    err = float((L[0] - ref).abs().max())
    for b in bufs:
        b[0, :, :n] = K
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in bufs:
        ops.potrf_(b[:, :, :n])
    e1.record(); torch.cuda.synchronize()
    print('n=%d pad=%d: %.3f ms, max err %.2e, info %d' % (n, pad, e0.elapsed_time(e1) / len(bufs), err, int(info[0])), flush=True)
