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


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K,S', [(1152, 1280, 208, 1),      # > 64 tiles of 128 x 128, short K: one launch, no split
                                     (1152, 1152, 1040, 2),     # batched, long K: split-K with the atomic epilogue
                                     (2048, 1024, 48, 1)])      # three 16-wide k blocks only: one pipelined trip, both look-ahead requests clamped (13 blocks above: remainder 1)
def test_gemm_float64_lds_dma_kernel(ta, tb, M, N, K, S):
    """gemm_f64_dma_kernel (128-aligned float64 shapes with more than 64 tiles; all four operand layouts, B broadcast over the batch,
    alpha / beta, split-K) against numpy."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(S, K, M) if ta else rng.randn(S, M, K)
    B = rng.randn(1, N, K) if tb else rng.randn(1, K, N)
    C0 = rng.randn(S, M, N)
    ref = 0.7 * (np.swapaxes(A, 1, 2) if ta else A) @ (np.swapaxes(B, 1, 2) if tb else B) - 0.3 * C0
    out = torch.as_tensor(C0).cuda()
    ops.gemm(torch.as_tensor(A).cuda(), torch.as_tensor(B).cuda(), ta, tb, 0.7, -0.3, out=out)
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * np.sqrt(K))


@pytest.mark.parametrize('M,N,K,lower', [(128, 128, 16, False), (256, 384, 1024, False), (130, 70, 50, False), (1, 5, 7, False),
                                          (1024, 1024, 4096, True), (300, 300, 333, True), (512, 8192, 512, False)])
def test_gemm_f32x3_is_f32_accurate(M, N, K, lower):
    """mxf_gemm_f32x3 (f32 operands split exactly into three bf16 terms, six bf16 MFMA products, f32 accumulate): the error against
    float64 must be at the level of the plain f32-MFMA kernel (and of an f32 accumulation of K terms), for aligned, ragged, tiny,
    split-K and lower-only shapes, with both signs and a wide dynamic range in the operands."""
    from mxfusion_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K)
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 0.7) * torch.exp(torch.randn(M, 1, device='cuda', generator=g))
    B = torch.exp(-torch.rand(N, K, device='cuda', generator=g) * 8) * (torch.rand(N, K, device='cuda', generator=g) - 0.3)
    ref = A.double() @ B.double().T
    C3 = ops.gemm_f32x3(A, B, lower_only=lower)
    C1 = ops.gemm(A[None], B[None], transB=True)[0]
    msk = torch.tril(torch.ones(M, N, device='cuda')) if lower else torch.ones(M, N, device='cuda')
    scale = (A.double().abs() @ B.double().abs().T)                      # elementwise condition-free scale sum |a||b|
    e3 = float((((C3.double() - ref).abs() / scale) * msk).max())
    e1 = float((((C1.double() - ref).abs() / scale) * msk).max())
    assert e3 < 4e-7, (e3, e1)                                              # a few f32 ulps of sum|a||b|
    assert e3 < 2.5 * e1 + 1e-7, (e3, e1)                                   # never materially worse than the f32 MFMA kernel
    if lower:
        assert float((C3 * (1 - msk)).abs().max()) == 0.0                  # strictly-upper blocks / entries untouched
    # alpha / beta and the planes-level entry points
    out = torch.full((M, N), 2.0, device='cuda')
    ops.gemm_f32x3_planes(ops.f32x3_split(A), ops.f32x3_split(B), M, N, K, alpha=0.5, beta=1.0, out=out)
    assert float(((out.double() - (0.5 * ref + 2.0)).abs() / (scale + 1.0)).max()) < 4e-7


@pytest.mark.parametrize('M,N,K,lower', [(128, 128, 16, False), (256, 384, 1024, False), (130, 70, 50, False), (1, 5, 7, False),
                                          (1024, 1024, 4096, True), (300, 300, 333, True), (512, 8192, 512, False),
                                          # the 128 x 256 kernel (M % 128 == 0, N % 256 == 0): one, odd and even k-block counts, split-K,
                                          # lower-only with rectangular tiles on the diagonal
                                          (128, 256, 16, False), (128, 256, 48, False), (256, 512, 1040, False), (384, 768, 2064, False),
                                          (1024, 1024, 80, True), (256, 256, 100000, True), (128, 512, 70000, False)])
@pytest.mark.parametrize('mag', [1.0, 3e-6, 7e5])
def test_gemm_f16x2_is_f32_accurate(M, N, K, lower, mag):
    """mxf_gemm_f16x2 (two power-of-two-scaled f16 terms per operand, three f16 MFMA products, f32 accumulate): error against float64
    at the level of the f32-MFMA kernel for aligned, ragged, tiny, split-K and lower-only shapes, operands of both signs with a wide
    dynamic range, and operand magnitudes far outside the f16 range (the per-operand scale comes from its max-abs word)."""
    from mxfusion_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K)
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 0.7) * torch.exp(torch.randn(M, 1, device='cuda', generator=g)) * mag
    B = torch.exp(-torch.rand(N, K, device='cuda', generator=g) * 8) * (torch.rand(N, K, device='cuda', generator=g) - 0.3) / mag ** 0.5
    ref = A.double() @ B.double().T
    C2 = ops.gemm_f16x2(A, B, lower_only=lower)
    C1 = ops.gemm(A[None], B[None], transB=True)[0]
    msk = torch.tril(torch.ones(M, N, device='cuda')) if lower else torch.ones(M, N, device='cuda')
    scale = (A.double().abs() @ B.double().abs().T)
    e2 = float((((C2.double() - ref).abs() / scale) * msk).max())
    e1 = float((((C1.double() - ref).abs() / scale) * msk).max())
    assert e2 < 6e-7, (e2, e1)                                              # 2^-22 per product at worst, a few f32 ulps of sum|a||b| in practice
    assert e2 < 2.5 * e1 + 2e-7, (e2, e1)
    if lower:
        assert float((C2 * (1 - msk)).abs().max()) == 0.0
    out = torch.full((M, N), 2.0, device='cuda') * float(scale.max())
    out0 = out.clone()
    ops.gemm_f16x2(A, B, alpha=0.5, beta=1.0, out=out, lower_only=lower)
    # (split-K partial sums are added to the large beta * C term one by one: a few more f32 roundings of that term)
    assert float((((out.double() - (0.5 * ref + out0.double())).abs() / (scale + out0.double().abs())) * msk).max()) < 2e-6
    # the planes-level entry points (operands split once, reused)
    Cp = ops.gemm_f16x2_planes(ops.f16x2_split(A), ops.f16x2_split(B), M, N, K, lower_only=lower)
    assert torch.equal(Cp * msk, C2 * msk) or float((((Cp.double() - ref).abs() / scale) * msk).max()) < 6e-7


@pytest.mark.parametrize('M,N,K', [(128, 256, 16), (256, 512, 1040), (1024, 8192, 1024), (130, 48, 50), (384, 65536, 64)])
def test_gemm_f16x2_planes_blocked_output_is_the_row_major_product_reordered(M, N, K):
    """lower_only = 2 of mxf_gemm_f16x2_planes: the full product with C in 16-column blocks, element (m, n) at ((n / 16) * M + m) * 16 + n % 16
    -- the layout the SVGP training step keeps T = H0 Kuf in (both GEMM kernels; the persistent 128 x 256 one walks several work items per
    workgroup at the larger shapes).  Must equal the row-major product bit for bit where the same kernel computes both, to rounding otherwise."""
    from mxfusion_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.rand(N, K, device='cuda', generator=g) - 0.3
    pa, pb = ops.f16x2_split(A), ops.f16x2_split(B)
    C = ops.gemm_f16x2_planes(pa, pb, M, N, K)
    out = torch.full((M * N,), float('nan'), device='cuda')
    ops.gemm_f16x2_planes(pa, pb, M, N, K, out=out.view(M, N), blocked=True)
    Cb = out.view(N // 16, M, 16).permute(1, 0, 2).reshape(M, N)
    assert not torch.isnan(Cb).any()
    ref = A.double() @ B.double().T
    scale = A.double().abs() @ B.double().abs().T
    assert float(((Cb.double() - ref).abs() / scale).max()) < 6e-7
    assert float(((Cb - C).abs() / scale.float()).max()) < 3e-7


def test_gemm_f16x2_zero_operand():
    from mxfusion_amd import ops
    A = torch.zeros(128, 64, device='cuda')
    B = torch.randn(128, 64, device='cuda')
    assert float(ops.gemm_f16x2(A, B).abs().max()) == 0.0


@pytest.mark.parametrize('xt,sync,pp', [(4, 0, 0), (4, 1, 0), (4, 2, 0), (8, 0, 0), (8, 2, 0), (16, 0, 0), (16, 1, 0), (16, 2, 0), (16, 0, 1), (16, 1, 1), (16, 2, 1)])
def test_wide_split_gemm_variants(xt, sync, pp):
    """Every form of the wide f16x2 kernel (128 rows; 256 rows by four 512-register waves; 256 rows by eight waves in two row halves, with and
    without the ping-pong phases) with every rendezvous mode (none, per column strip / k split, per XCD), on
    shapes where the kernel walks several work items per workgroup (T-like: 8 persistent rounds) and where the tiles of a k split meet
    inside the k loop (Psi2-like), against float64.  The variants are knobs of the PROBE build of the library (the shipped one has them
    compiled in), hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = os.path.join(root, 'mxfusion_amd', 'libmxf_gp_probe.so')
    if not os.path.exists(probe):
        pytest.skip('probe build of the library absent (make -C mxfusion_amd/csrc probe)')
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from mxfusion_amd import ops\n"
        "g = torch.Generator(device='cuda').manual_seed(5)\n"
        "M, N, K = 1024, 131072, 1024\n"
        "A = torch.randn(M, K, device='cuda', generator=g); B = torch.rand(N, K, device='cuda', generator=g) - 0.3\n"
        "pa, pb = ops.f16x2_split(A), ops.f16x2_split(B)\n"
        "out = torch.full((M * N,), float('nan'), device='cuda')\n"
        "for _ in range(3): ops.gemm_f16x2_planes(pa, pb, M, N, K, out=out.view(M, N), blocked=True)\n"
        "Cb = out.view(N // 16, M, 16).permute(1, 0, 2).reshape(M, N)\n"
        "idx = torch.randint(0, N, (2048,), device='cuda', generator=g)\n"
        "ref = A.double() @ B[idx].double().T; sc = A.double().abs() @ B[idx].double().abs().T\n"
        "assert not torch.isnan(Cb).any()\n"
        "e1 = float(((Cb[:, idx].double() - ref).abs() / sc).max()); assert e1 < 6e-7, e1\n"
        "K2 = 400000\n"
        "C = torch.rand(M, K2, device='cuda', generator=g) - 0.2; pc = ops.f16x2_split(C)\n"
        "P = torch.zeros(M, M, device='cuda')\n"
        "for _ in range(3): ops.gemm_f16x2_planes(pc, pc, M, M, K2, out=P, lower_only=True)\n"
        "ref = torch.tril(C.double() @ C.double().T); sc = C.double().abs() @ C.double().abs().T\n"
        "e2 = float(((P.double() - ref).abs() / sc).max())\n"
        "assert e2 < 2e-6, e2      # f32 accumulation of K2 / splits = 16 000 terms per partial sum, ~25 partial sums added atomically\n"
        "assert float(torch.triu(P, 1).abs().max()) == 0.0\n"
        "print('errors', e1, e2); print('ok')\n") % root
    env = dict(os.environ, MXF_GP_LIB=probe, MXF_SPLIT_XT=str(xt), MXF_SPLIT_SYNC=str(sync), MXF_SPLIT_PP=str(pp))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


@pytest.mark.parametrize('M,N,K,blocked,with_u', [(256, 256, 48, False, False), (256, 512, 64, True, True), (1024, 8192, 1024, True, True),
                                                   (512, 2048, 1040, False, True), (1024, 65536, 1024, True, True), (256, 1024, 2048, False, True)])
def test_gemm_f16x2_kmajor_second_operand(M, N, K, blocked, with_u):
    """mxf_gemm_f16x2_planes_kmajor (gemm_bt.hip, r06): C = A Bt with Bt (K x N) split AS STORED -- its 16 x 16 tiles [k][n] are turned into
    MFMA fragments by LDS-DMA + ds_read_b64_tr_b16 -- against float64 and against the row-operand product on the explicitly transposed copy
    (same arithmetic: three f16 products per k block, f32 accumulation); the fused row U = w^T Bt against float64.  Shapes: one k trip with
    clamped look-ahead (48), remainders 1 and 2 of the three-slot ring (64, 1040), several persistent rounds per workgroup, the T shape of
    BASELINE configs[3] (1024 x 65536 x 1024), K = 2048 (the largest the fused U row takes)."""
    from mxfusion_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g) * torch.exp(torch.randn(M, 1, device='cuda', generator=g))
    Bt = torch.rand(K, N, device='cuda', generator=g) - 0.3
    w = torch.randn(K, device='cuda', generator=g) * 3.0 if with_u else None
    pa, pbt = ops.f16x2_split(A), ops.f16x2_split(Bt)
    out = torch.full((M * N,), float('nan'), device='cuda')
    r = ops.gemm_f16x2_planes_kmajor(pa, pbt, M, N, K, alpha=0.75, out=out.view(M, N), blocked=blocked, w=w)
    C = out.view(N // 16, M, 16).permute(1, 0, 2).reshape(M, N) if blocked else out.view(M, N)
    assert not torch.isnan(C).any()
    idx = torch.randint(0, N, (1024,), device='cuda', generator=g)
    ref = 0.75 * (A.double() @ Bt[:, idx].double())
    scale = A.double().abs() @ Bt[:, idx].double().abs()
    assert float(((C[:, idx].double() - ref).abs() / scale).max()) < 6e-7
    # the row-operand kernel on the transposed copy forms the same products in the same order
    C2 = ops.gemm_f16x2_planes(pa, ops.f16x2_split(Bt.T.contiguous()), M, N, K, alpha=0.75)
    assert float(((C - C2).abs() / (A.abs() @ Bt.abs())).max()) < 3e-7
    if with_u:
        U = r[1]
        uref = w.double() @ Bt.double()
        uscale = w.double().abs() @ Bt.double().abs()
        assert float(((U.double() - uref).abs() / uscale).max()) < 6e-7
