import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1024, 8192):
    A = torch.randn(1, n, n, device='cuda', dtype=torch.float64); K = A @ A.transpose(1, 2) / n + torch.eye(n, device='cuda', dtype=torch.float64)
    work = K.clone()
    def f():
        work.copy_(K); ops.potrf_(work)
    t = bench(f) ; tc = bench(lambda: work.copy_(K))
    L = work.clone()
    t2 = bench(lambda: ops.trtri(L))
    B = torch.randn(1, n, 1, device='cuda', dtype=torch.float64)
    t3 = bench(lambda: ops.trsm_(L, B.clone()))
    print(f'n={n}: potrf {t-tc:.3f} ms ({n**3/3/(t-tc)/1e9:.1f} TF)  trtri {t2:.3f} ms  trsm(nrhs=1) {t3:.3f} ms', flush=True)
