"""probe: latency of the fused potrf panel kernel: n = 64 (diagonal factor only), n = 192 (one panel with 128 rows below + ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mxfusion_amd import ops
for dt in (torch.float64, torch.float32):
    for n in (64, 128, 192, 1024):
        A = torch.randn(n, n, device='cuda', dtype=torch.float64)
        K = (A @ A.T / n + torch.eye(n, device='cuda', dtype=torch.float64)).to(dt)[None]
        W = K.clone()
        for _ in range(3): W.copy_(K); ops.potrf_(W)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        tot = 0.0
        for _ in range(reps):
            W.copy_(K)
            e0.record(); ops.potrf_(W); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print(dt, n, 'potrf %.1f us' % (tot / reps * 1e3), flush=True)
