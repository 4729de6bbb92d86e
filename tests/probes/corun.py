"""r03: can a Gram-type pass (VALU + HBM writes) hide behind a split GEMM (matrix pipe; one 512-thread workgroup per CU, half the register
file free)?  T-shaped and Psi2-shaped GEMM alone, an 8.6 GB float32 Gram alone, and both at once on two streams (wall time).
usage: corun.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
M, SB = 1024, 2097152
pa = ops.f16x2_split(torch.randn(M, M, device='cuda'))
B = torch.rand(SB, M, device='cuda'); pb = ops.f16x2_split(B); del B
out = torch.empty(M, SB, device='cuda')
C = torch.rand(M, SB, device='cuda'); pc = ops.f16x2_split(C); del C
psi = torch.zeros(M, M, device='cuda')
N, Q = 46336, 8
X = torch.rand(1, N, Q, device='cuda') * 6 - 3
ls = torch.ones(1, Q, device='cuda'); var = torch.ones(1, 1, device='cuda')
G = torch.empty(1, N, N, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
gemms = {'T': lambda: ops.gemm_f16x2_planes(pa, pb, M, SB, M, out=out, blocked=True),
         'Psi2': lambda: ops.gemm_f16x2_planes(pc, pc, M, M, SB, out=psi, lower_only=True)}
gram = lambda: ops.gram('rbf', X, None, ls, var, True, out=G)


def wall(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def on(stream, fn):
    with torch.cuda.stream(stream):
        fn()


tg = wall(lambda: on(s2, gram))
print('Gram alone (%.1f GB): %.3f ms' % (N * N * 4 / 1e9, tg))
for name, g in gemms.items():
    ta = wall(lambda: on(s1, g))

    def both(first_gemm=True):
        if first_gemm:
            on(s1, g); on(s2, gram)
        else:
            on(s2, gram); on(s1, g)
        s1.synchronize(); s2.synchronize()
    tb = wall(lambda: both(True)); tc = wall(lambda: both(False))
    print('%s alone %.3f ms; with the Gram pass at once: %.3f (GEMM first) / %.3f (Gram first); sum %.3f' % (name, ta, tb, tc, ta + tg))
