"""probe: f32x3 (bf16-split) GEMM vs plain f32 MFMA GEMM -- accuracy against float64 and time, at the SVGP step's two shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mxfusion_amd import ops

def tms(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

torch.manual_seed(0)
for (M, N, K, lower) in [(256, 384, 1024, False), (130, 70, 50, False), (1024, 1024, 65536, True), (1024, 262144, 1024, False), (1024, 1024, 2097152, True),
                         (1024, 2097152, 1024, False)]:
    A = torch.rand(M, K, device='cuda') * 2 - 0.7
    B = torch.exp(-torch.rand(N, K, device='cuda') * 6)
    if M * N * K <= 1024 * 1024 * 65536:
        ref = A.double() @ B.double().T
        C3 = ops.gemm_f32x3(A, B, lower_only=lower)
        C1 = ops.gemm(A[None], B[None], transB=True)[0]
        msk = torch.tril(torch.ones(M, N, device='cuda', dtype=torch.bool)) if lower else torch.ones(M, N, device='cuda', dtype=torch.bool)
        den = ref.abs().max()
        e3 = ((C3.double() - ref).abs() * msk).max() / den
        e1 = ((C1.double() - ref).abs() * msk).max() / den
        print('M=%d N=%d K=%d lower=%d  max err / max|C|: f32x3 %.2e   f32-mfma %.2e' % (M, N, K, lower, e3, e1), flush=True)
    if K >= 1024 and M >= 1024:
        out = torch.zeros(M, N, device='cuda')
        t3 = tms(lambda: ops.gemm_f32x3(A, B, out=out, lower_only=lower))
        fl = 2.0 * M * N * K * (0.5625 if lower else 1.0)
        print('   f32x3 (incl. splitting both operands) %.2f ms = %.0f TF' % (t3, fl / t3 / 1e9), flush=True)
