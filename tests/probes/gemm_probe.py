import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops, _lib
def bench(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tag = os.path.basename(_lib.LIB_PATH)
dts = ((torch.float32, 'f32'),) if len(sys.argv) > 1 and sys.argv[1] == 'f32' else ((torch.float32, 'f32'), (torch.float64, 'f64'))
for dt, fl in dts:
    big = 2097152 if dt == torch.float32 else 1048576
    for (M, N, K, ta, tb) in ((4096, 4096, 4096, 0, 0), (1024, big, 1024, 0, 0), (1024, 1024, big, 1, 0), (64, 1024, 960, 0, 0)):
        A = torch.randn(1, K, M, device='cuda', dtype=dt) if ta else torch.randn(1, M, K, device='cuda', dtype=dt)
        B = torch.randn(1, N, K, device='cuda', dtype=dt) if tb else torch.randn(1, K, N, device='cuda', dtype=dt)
        out = torch.empty(1, M, N, device='cuda', dtype=dt)
        ms = bench(lambda: ops.gemm(A, B, bool(ta), bool(tb), out=out))
        print(f'[{tag}] gemm {fl} {M}x{N}x{K} ta={ta} tb={tb}: {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TFLOP/s', flush=True)
        del A, B, out; torch.cuda.empty_cache()
