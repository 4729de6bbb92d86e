"""float64 building blocks of the exact-GP step (BASELINE configs[1]) timed alone: potrf, trtri, L^-T L^-1 (lower), and square GEMMs, per n.
Rates in TFLOP/s on the algorithmic counts n^3/3, n^3/3, n^3/3 (lower half of 2 n^3 / ... counted as n^3/3), 2 n^3."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
torch.manual_seed(0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (1024, 2048, 4096, 8192):
    X = torch.randn(n, 8, device='cuda', dtype=torch.float64)
    K = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(n, device='cuda', dtype=torch.float64)
    bufs = [K[None].clone() for _ in range(6)]
    it = iter(bufs)
    t_potrf = timeit(lambda: ops.potrf_(next(it)), reps=5)
    L = bufs[0]
    t_trtri = timeit(lambda: ops.trtri(L))
    Li = ops.trtri(L)
    t_ltl = timeit(lambda: ops.gemm(Li, Li, transA=True))
    t_gemm = timeit(lambda: ops.gemm(K[None], K[None]))
    f = n ** 3 / 3.0
    print('n=%5d  potrf %.3f ms (%.1f TF)  trtri %.3f ms (%.1f TF)  L^-T L^-1 (full product as called here) %.3f ms  gemm n^3 %.3f ms (%.1f TF)'
          % (n, t_potrf, f / t_potrf / 1e9, t_trtri, f / t_trtri / 1e9, t_ltl, t_gemm, 2.0 * n ** 3 / t_gemm / 1e9), flush=True)
