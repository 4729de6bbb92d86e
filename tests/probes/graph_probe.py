"""probe: which library calls survive hipGraph capture (torch.cuda.graph)."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()
import torch
from mxfusion_amd import ops

def cap(name, f, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm): f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print('capturing', name, flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = f()
    print('  captured', flush=True)
    g.replay(); torch.cuda.synchronize()
    print('  replayed ok', name, flush=True)
    return out

dt = torch.float32
B, M, Q, S = 4096, 256, 8, 2
X = torch.rand(S, B, Q, device='cuda', dtype=dt); Y = torch.rand(1, B, 1, device='cuda', dtype=dt)
Z = torch.rand(M, Q, device='cuda', dtype=dt); ls = torch.ones(Q, device='cuda', dtype=dt); var = torch.ones(1, device='cuda', dtype=dt)
noise = torch.full((1,), 0.1, device='cuda', dtype=dt); mu = torch.zeros(M, 1, device='cuda', dtype=dt)
W = torch.zeros(M, M, device='cuda', dtype=dt); sd = torch.ones(M, device='cuda', dtype=dt)
cap('gram', lambda: ops.gram('rbf', X, None, ls[None], var[None], True))
A = torch.rand(1, 512, 512, device='cuda', dtype=torch.float64); A = A @ A.transpose(1, 2) + 512 * torch.eye(512, device='cuda', dtype=torch.float64)
cap('potrf', lambda: ops.potrf_(A.clone()))
cap('trtri', lambda: ops.trtri(torch.tril(A)))
cap('svgp fwd', lambda: ops.svgp_logpdf('rbf', X, Y, Z, noise, mu, W, sd, ls, var, True, jitter=1e-6, want_grad=False))
r = cap('svgp fwd+bwd', lambda: ops.svgp_logpdf('rbf', X, Y, Z, noise, mu, W, sd, ls, var, True, jitter=1e-6, gscale=0.5, want_grad=True))
print(float(r['logL'][0]), flush=True)
