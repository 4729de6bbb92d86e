import time, torch, numpy as np, sys
sys.path.insert(0, '.')
from mxfusion_amd import ops
for dt in (torch.float32, torch.float64):
    for N in (8192, 32768, 65536):
        X = (torch.rand(1, N, 8, device='cuda', dtype=dt) * 6 - 3)
        ls = torch.ones(1, 8, device='cuda', dtype=dt); var = torch.ones(1, 1, device='cuda', dtype=dt)
        out = torch.empty(1, N, N, device='cuda', dtype=dt)
        for kind in ('rbf', 'matern52'):
            ops.gram(kind, X, None, ls, var, True, out=out); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gram(kind, X, None, ls, var, True, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gb = N * N * out.element_size() / 1e9
            print(f'{kind} {dt} N={N}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s', flush=True)
        del out
for dt, fl in ((torch.float32, 'f32'), (torch.float64, 'f64')):
    for (M, N, K) in ((4096, 4096, 4096), (1024, 1024, 65536), (1024, 65536, 1024)):
        A = torch.randn(1, M, K, device='cuda', dtype=dt); B = torch.randn(1, K, N, device='cuda', dtype=dt)
        out = torch.empty(1, M, N, device='cuda', dtype=dt)
        ops.gemm(A, B, out=out); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.gemm(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f'gemm {fl} {M}x{N}x{K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TFLOP/s', flush=True)
