"""GPU parity: MFMA GEMM (f32 32x32x2 / f64 16x16x4) through the C ABI vs torch CPU float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K,S', [(1, 1, 1, 1), (3, 5, 7, 2), (128, 128, 16, 1), (130, 257, 33, 2), (64, 200, 4100, 1),
                                     (300, 3, 1024, 1)])
def test_gemm_vs_cpu(dtype, tol, ta, tb, M, N, K, S):
    from mxfusion_amd import ops
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.randn(S, K, M) if ta else rng.randn(S, M, K)
    B = rng.randn(1, N, K) if tb else rng.randn(1, K, N)    # B broadcast over S, and asymmetric (transpose-detecting)
    C0 = rng.randn(S, M, N)
    ref = 0.7 * (np.swapaxes(A, 1, 2) if ta else A) @ (np.swapaxes(B, 1, 2) if tb else B) - 0.3 * C0
    out = torch.as_tensor(C0, dtype=dtype).cuda()
    ops.gemm(torch.as_tensor(A, dtype=dtype).cuda(), torch.as_tensor(B, dtype=dtype).cuda(), ta, tb, 0.7, -0.3, out=out)
    scale = np.sqrt(K)
    assert np.allclose(out.cpu().numpy(), ref, rtol=tol, atol=tol * scale)
    out2 = ops.gemm(torch.as_tensor(A, dtype=dtype).cuda(), torch.as_tensor(B, dtype=dtype).cuda(), ta, tb)
    assert np.allclose(out2.cpu().numpy(), (ref + 0.3 * C0) / 0.7, rtol=tol, atol=tol * scale)
