"""V = L^-1 Kuf of the whitened tier at the bench shape (M = 1024, S B = 2 097 152): the planes-output product alone, with the transposed
planes, and with the U partial sums (mxf_gemm_f16x2_planes_out); ms per launch (torch events on the current stream)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from mxfusion_amd import ops

M, N, K = 1024, 65536 * 32, 1024
g = torch.Generator(device='cuda').manual_seed(1)
A = torch.tril(torch.randn(M, K, device='cuda', generator=g))
B = torch.rand(N, K, device='cuda', generator=g) ** 3
a = torch.randn(M, device='cuda', generator=g)
As, Bs = ops.f16x2_split(A), ops.f16x2_split(B)
del B
ref = None


def run(**kw):
    for _ in range(2):
        ops.gemm_f16x2_planes_out(As, Bs, M, N, K, alpha=8.0, a_lower=True, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_f16x2_planes_out(As, Bs, M, N, K, alpha=8.0, a_lower=True, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


print('planes only        %.3f ms' % run())
print('+ transposed       %.3f ms' % run(transposed=True))
print('+ transposed + U   %.3f ms' % run(a=a))
