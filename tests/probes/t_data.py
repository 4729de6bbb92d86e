"""Is the split GEMM power-bound?  Time of the T-shaped product (1024 x SB x 1024, blocked output) for operands of different bit activity:
uniform random planes (what bench.py's roofline_mfma uses), all-zero B planes, and the B planes of the bench model itself (RBF covariances
of 8-dimensional uniform inputs at length-scale 1: almost all of them underflow the scaled f16 terms).  usage: t_data.py [SB]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
M, SB = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
pa = ops.f16x2_split(torch.randn(M, M, device='cuda'))
out = torch.empty(M, SB, device='cuda')


def timed(pb, reps=5):
    fn = lambda: ops.gemm_f16x2_planes(pa, pb, M, SB, M, out=out, blocked=True)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = torch.rand(SB, M, device='cuda'); pb = ops.f16x2_split(B); del B
print('uniform [0,1) B: %.3f ms' % timed(pb)); del pb
B = torch.zeros(SB, M, device='cuda'); pb = ops.f16x2_split(B); del B
print('zero B:          %.3f ms' % timed(pb)); del pb
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.rand(SB, 8, device='cuda', generator=g) * 6 - 3
Z = X[torch.randperm(SB, device='cuda', generator=g)[:M]].contiguous()
for ell in (1.0, 2.2):
    one = lambda v, n: torch.full((1, n), v, device='cuda')
    K = ops.gram('rbf', X[None], Z[None], one(ell, 8), one(1.0, 1), True)[0]        # (SB, M) covariances
    print('  ell %.1f: fraction of covariances below 2^-14: %.3f, mean %.2e' % (ell, float((K < 2 ** -14).float().mean()), float(K.mean())))
    pb = ops.f16x2_split(K); del K
    print('RBF covariances, ell %.1f: %.3f ms' % (ell, timed(pb))); del pb
