import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
N, Q = 65536, 8
for dt in (torch.float32, torch.float64):
    X = torch.rand(1, N, Q, device='cuda', dtype=dt) * 6 - 3
    ls = torch.ones(1, Q, device='cuda', dtype=dt); var = torch.ones(1, 1, device='cuda', dtype=dt)
    out = torch.empty(1, N, N, device='cuda', dtype=dt)
    for kind in ('rbf', 'matern52'):
        ops.gram(kind, X, None, ls, var, True, out=out); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gram(kind, X, None, ls, var, True, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"TR={os.environ.get('MXF_GRAM_TR','-')} NT={os.environ.get('MXF_GRAM_NT','-')} {kind} {str(dt)[6:]}: {ms:.3f} ms {N*N*out.element_size()/ms/1e6:.0f} GB/s", flush=True)
    del out; torch.cuda.empty_cache()
