"""GPU parity: blocked Cholesky / triangular solves / inverse / log-det / Gram reverse mode / elementwise
kernels (through the C ABI) vs the CPU oracle (torch float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402


def _spd(rng, S, n, cond_shift=1.0):
    A = rng.randn(S, n, n)
    return A @ np.swapaxes(A, 1, 2) / n + cond_shift * np.eye(n)[None]


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-11), (torch.float32, 2e-4)])
@pytest.mark.parametrize('S,n', [(1, 1), (2, 5), (1, 64), (3, 65), (1, 200), (2, 513), (1, 1100),
                                 # n % 64 == 0, 128 <= n <= 1024 in float64: the one-launch tile-dataflow kernel (batched)
                                 (3, 128), (2, 192), (1, 640), (2, 1024),
                                 # larger n % 64 == 0 in float64: the same kernel once per 512-column outer panel
                                 (1, 1536), (2, 2048),
                                 # 64 block rows below the first outer panel: chain launch + potrf_rows_kernel, the next panel's rows-below head
                                 # update on the auxiliary stream (production knobs)
                                 (1, 4608)])
def test_potrf_trsm_trtri_logdet(dtype, tol, S, n):
    from mxfusion_amd import ops
    rng = np.random.RandomState(n)
    A = _spd(rng, S, n)
    Lref = np.linalg.cholesky(A)
    L, info = ops.potrf_(_dev(A, dtype))
    assert int(info.abs().sum()) == 0
    Lh = L.cpu().numpy()
    assert np.allclose(Lh, Lref, rtol=tol, atol=tol)
    assert np.all(np.triu(Lh, 1) == 0)          # MXNet potrf returns a clean lower triangle
    sld = ops.sumlogdiag(L).cpu().numpy()
    assert np.allclose(sld, np.log(np.diagonal(Lref, axis1=1, axis2=2)).sum(-1), rtol=tol, atol=tol * n)
    for nrhs in (1, 3, 300):
        B = rng.randn(S, n, nrhs)
        for tr in (False, True):
            Lt = np.swapaxes(Lref, 1, 2) if tr else Lref
            ref = np.linalg.solve(Lt, B)
            X = ops.trsm_(_dev(Lref, dtype), _dev(B, dtype), transpose=tr).cpu().numpy()
            assert np.allclose(X, ref, rtol=tol * 10, atol=tol * 10 * np.abs(ref).max()), (nrhs, tr)
    Li = ops.trtri(_dev(Lref, dtype)).cpu().numpy()
    assert np.allclose(Li, np.linalg.inv(Lref), rtol=tol * 10, atol=tol * 10)
    # L broadcast over samples (stride 0)
    B = rng.randn(2, n, 4)
    X = ops.trsm_(_dev(Lref[:1], dtype), _dev(B, dtype)).cpu().numpy()
    assert np.allclose(X, np.linalg.solve(Lref[:1], B), rtol=tol * 10, atol=tol * 10 * np.abs(X).max())


def test_potrf_tiles_not_positive_definite_reports_info():
    """the one-launch kernel (float64, n = 256): a negative pivot in the third block row of batch item 1 is reported (first failing index,
    1-based), the factor stays finite, item 0 is untouched; and a launch right behind it starts from clean hand-off counters."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(5)
    A = _spd(rng, 2, 256)
    A[1, 150, 150] = -5.0
    L, info = ops.potrf_(_dev(A, torch.float64))
    L2, info2 = ops.potrf_(_dev(A[:1], torch.float64))
    info = info.cpu().numpy()
    assert info[0] == 0 and info[1] == 151, info
    assert torch.isfinite(L).all()
    assert np.allclose(L[0].cpu().numpy(), np.linalg.cholesky(A[0]), rtol=1e-11, atol=1e-11)
    assert int(info2[0]) == 0 and np.allclose(L2[0].cpu().numpy(), np.linalg.cholesky(A[0]), rtol=1e-11, atol=1e-11)


def test_potrf_panel_form_not_positive_definite_reports_info():
    """the same in the panel form (n = 1024: two outer panels, trailing update in between): a negative pivot in the SECOND panel of batch item 1."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(6)
    A = _spd(rng, 2, 1024)
    A[1, 700, 700] = -5.0
    L, info = ops.potrf_(_dev(A, torch.float64))
    info = info.cpu().numpy()
    assert info[0] == 0 and info[1] == 701, info
    assert np.allclose(L[0].cpu().numpy(), np.linalg.cholesky(A[0]), rtol=1e-11, atol=1e-11)
    # the leading 700 x 700 block of the failing item is still its Cholesky factor
    assert np.allclose(L[1, :700, :700].cpu().numpy(), np.linalg.cholesky(A[1, :700, :700]), rtol=1e-11, atol=1e-11)


def test_potrf_tiles_scratch_rings_wrap_and_streams_do_not_collide():
    """The tile kernel hands tiles on through two per-handle rings (progress counters, inverse blocks of the diagonal factors): 400
    back-to-back factorisations wrap both; two streams factoring at the same time (as the SVGP step does with Kuu and Su) use disjoint
    regions.  Every result is checked against the first one / against LAPACK."""
    from mxfusion_amd import ops
    rng = np.random.RandomState(11)
    A = _spd(rng, 1, 1024)
    ref = np.linalg.cholesky(A[0])
    Ad = _dev(A)
    first = ops.potrf_(Ad.clone())[0]
    assert np.allclose(first[0].cpu().numpy(), ref, rtol=1e-11, atol=1e-11)
    last = None
    for _ in range(400):
        last, info = ops.potrf_(Ad.clone())
    # (not bit-identical run to run: the trailing update between the two outer panels is a split-K product with float64 atomics)
    assert int(info[0]) == 0 and torch.allclose(last, first, rtol=1e-12, atol=1e-13)
    B = _dev(_spd(rng, 1, 768))
    refB = np.linalg.cholesky(B[0].cpu().numpy())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(20):
        a, b2 = Ad.clone(), B.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            la, _ = ops.potrf_(a)
        with torch.cuda.stream(s2):
            lb, _ = ops.potrf_(b2)
        outs.append((la, lb))
    torch.cuda.synchronize()
    for la, lb in outs:
        assert torch.allclose(la, first, rtol=1e-12, atol=1e-13)
        assert np.allclose(lb[0].cpu().numpy(), refB, rtol=1e-11, atol=1e-11)


def test_potrf_outer_panel_split_launch_batched():
    """The two-launch form of an outer panel (chain rows, then the rows below without hand-offs) only engages with >= 64 block rows below
    a panel (n >= 4608); MXF_POTRF_SPLIT_ROWS=2 (a knob of the PROBE build of the library, read once per process, hence the subprocess)
    forces it at n = 1536 with a batch of 2."""
    import os
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mxfusion_amd', 'libmxf_gp_probe.so')
    if not os.path.exists(probe):
        pytest.skip('probe build of the library absent (make -C mxfusion_amd/csrc probe)')
    code = (
        "import numpy as np, torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mxfusion_amd import ops\n"
        "rng = np.random.RandomState(3)\n"
        "A = rng.randn(2, 1536, 1536); A = A @ np.swapaxes(A, 1, 2) / 1536 + np.eye(1536)[None]\n"
        "L, info = ops.potrf_(torch.as_tensor(A).cuda())\n"
        "assert int(info.abs().sum()) == 0\n"
        "assert np.allclose(L.cpu().numpy(), np.linalg.cholesky(A), rtol=1e-11, atol=1e-11)\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MXF_POTRF_SPLIT_ROWS='2', MXF_GP_LIB=probe), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_potrf_not_positive_definite_reports_info():
    from mxfusion_amd import ops, _lib
    A = np.eye(70)[None].repeat(2, 0)
    A[1, 66, 66] = -1.0
    L, info = ops.potrf_(_dev(A))
    assert info.cpu().tolist() == [0, 67]
    with pytest.raises(_lib.MXFError):
        ops.check_info(info)


KINDS = {'rbf': O.RBF, 'matern12': O.Matern12, 'matern32': O.Matern32, 'matern52': O.Matern52}


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 2e-4)])
@pytest.mark.parametrize('kind', list(KINDS))
@pytest.mark.parametrize('N,N2,Q,S,ard', [(7, 5, 3, 1, True), (70, 300, 8, 2, True), (130, None, 5, 2, False), (64, 257, 16, 1, True),
                                          (33, 700, 1, 1, True), (100, 260, 2, 1, False), (90, 1000, 12, 1, True)])
def test_gram_bwd_vs_autograd(kind, N, N2, Q, S, ard, dtype, tol):
    from mxfusion_amd import ops
    rng = np.random.RandomState(N + Q)
    X = rng.uniform(-2, 2, (S, N, Q))
    X2 = None if N2 is None else rng.uniform(-2, 2, (1, N2, Q))      # X2 broadcast over S: its gradient sums over S
    ls = rng.rand(1, Q if ard else 1) + 0.8
    var = rng.rand(1, 1) + 0.5
    dK = rng.randn(S, N, N if N2 is None else N2)
    k = KINDS[kind](Q, ARD=ard)
    tX, tls, tvar = [O.T(a).clone().requires_grad_(True) for a in (X, ls, var)]
    tX2 = None if X2 is None else O.T(X2).clone().requires_grad_(True)
    K = k.K(tX, tX2, **{k.name + '_lengthscale': tls, k.name + '_variance': tvar})
    (K * O.T(dK)).sum().backward()
    if kind == 'matern12' and Q == 1:
        tol = max(tol, 1e-6)     # |x - z| of the closest pairs from the reference's expansion form x^2 - 2xz + z^2 carries ~1e-8 relative noise
    dX, dX2, dls, dvar = ops.gram_bwd(kind, _dev(X, dtype), None if X2 is None else _dev(X2, dtype), _dev(ls, dtype), _dev(var, dtype), ard,
                                      _dev(dK, dtype))
    for got, ref, name in ((dX, tX.grad, 'dX'), (dX2, None if tX2 is None else tX2.grad, 'dX2'), (dls, tls.grad, 'dls'),
                           (dvar, tvar.grad, 'dvar')):
        if ref is None:
            assert got is None
            continue
        r = ref.numpy()
        assert np.allclose(got.cpu().numpy(), r, rtol=tol, atol=tol * max(1., np.abs(r).max())), name


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_elementwise_kernels(dtype, tol):
    from mxfusion_amd import ops
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(1000) * 3, [-80., -30., 0., 30., 80.]])
    y = ops.softplus(_dev(x, dtype)).cpu().numpy()
    assert np.allclose(y, O.softplus(O.T(x)).numpy(), rtol=tol, atol=1e-30)
    dy = rng.randn(x.size)
    dx = ops.softplus_bwd_(_dev(x, dtype), _dev(dy, dtype), torch.zeros(x.size, dtype=dtype).cuda()).cpu().numpy()
    assert np.allclose(dx, dy / (1 + np.exp(-x)), rtol=tol, atol=tol)
    # Normal reparameterisation + log-pdf with reverse mode (normal.py:52-92, factor_graph.py:223)
    S, n = 5, 333
    mean, var, eps = rng.randn(n), rng.rand(n) + 0.2, rng.randn(S, n)
    tm, tv = O.T(mean).requires_grad_(True), O.T(var).requires_grad_(True)
    xs = O.normal_draw(tm, tv, O.T(eps))
    xg = ops.normal_reparam(_dev(mean, dtype), _dev(var, dtype), _dev(eps, dtype))
    assert np.allclose(xg.cpu().numpy(), xs.detach().numpy(), rtol=tol, atol=tol)
    xs_leaf = xs.detach().clone().requires_grad_(True)
    pm, pv = O.T(rng.randn(n)).requires_grad_(True), O.T(rng.rand(n) + 0.3).requires_grad_(True)
    lp = O.factor_sum(O.normal_log_pdf(pm, pv, xs_leaf)) * -1.0      # scale = -1/S (posterior term of variational.py:107)
    lp.backward()
    out = torch.zeros(1, dtype=dtype).cuda()
    dxa, dma, dva = [torch.zeros(sh, dtype=dtype).cuda() for sh in ((S, n), (n,), (n,))]
    ops.normal_logpdf_(_dev(xs_leaf.detach().numpy(), dtype), _dev(pm.detach().numpy(), dtype), _dev(pv.detach().numpy(), dtype),
                       -1.0 / S, out, dxa, dma, dva)
    assert np.allclose(out.cpu().numpy(), float(lp), rtol=max(tol, 1e-12) * 10)
    assert np.allclose(dxa.cpu().numpy(), xs_leaf.grad.numpy(), rtol=tol * 10, atol=tol * 10)
    assert np.allclose(dma.cpu().numpy(), pm.grad.numpy(), rtol=tol * 10, atol=tol * 10)
    assert np.allclose(dva.cpu().numpy(), pv.grad.numpy(), rtol=tol * 100, atol=tol * 100)
    # scalar (broadcast) prior mean/var as in Normal.define_variable(mean=0, variance=1)
    out2 = torch.zeros(1, dtype=dtype).cuda()
    ops.normal_logpdf_(xg, _dev([0.], dtype), _dev([1.], dtype), 1.0 / S, out2)
    ref2 = float(O.factor_sum(O.normal_log_pdf(O.T([0.]), O.T([1.]), xs.detach())))
    assert np.allclose(out2.cpu().numpy(), ref2, rtol=max(tol, 1e-12) * 10)
    # reparam reverse mode
    gx = rng.randn(S, n)
    (xs * O.T(gx)).sum().backward()
    dm, dv = torch.zeros(n, dtype=dtype).cuda(), torch.zeros(n, dtype=dtype).cuda()
    ops.normal_reparam_bwd_(_dev(var, dtype), _dev(eps, dtype), _dev(gx, dtype), dm, dv)
    assert np.allclose(dm.cpu().numpy(), tm.grad.numpy(), rtol=tol * 10, atol=tol * 10)
    assert np.allclose(dv.cpu().numpy(), tv.grad.numpy(), rtol=tol * 10, atol=tol * 10)
    # MXNet Adam, 3 steps vs the oracle's restatement
    w = rng.randn(257)
    opt = O.MXNetAdam(0.05)
    wd, m, v = _dev(w, dtype), torch.zeros(257, dtype=dtype).cuda(), torch.zeros(257, dtype=dtype).cuda()
    p = {'w': O.T(w)}
    for t in range(1, 4):
        g = rng.randn(257)
        p = opt.step(p, {'w': O.T(g)}, batch_size=4)
        ops.adam_step_(wd, _dev(g, dtype), m, v, 0.05, t, rescale_grad=0.25)
    assert np.allclose(wd.cpu().numpy(), p['w'].numpy(), rtol=tol * 10, atol=tol * 10)
    # MXNet SGD (python/mxnet/optimizer: sgd_update / sgd_mom_update), plain and with momentum + weight decay, 3 steps in numpy float64
    for momentum, wdecay in ((0.0, 0.0), (0.9, 1e-2)):
        w0 = rng.randn(300)
        wr, mr = w0.copy(), np.zeros(300)
        wdev = _dev(w0, dtype)
        mdev = torch.zeros(300, dtype=dtype).cuda() if momentum else None
        for t in range(3):
            g = rng.randn(300)
            gi = 0.25 * g + wdecay * wr
            if momentum:
                mr = momentum * mr - 0.05 * gi
                wr = wr + mr
            else:
                wr = wr - 0.05 * gi
            ops.sgd_step_(wdev, _dev(g, dtype), mdev, 0.05, momentum=momentum, wd=wdecay, rescale_grad=0.25)
        assert np.allclose(wdev.cpu().numpy(), wr, rtol=tol * 10, atol=tol * 10)
    # the other rules the Trainer seam takes by name (mxf_opt_step): MXNet 'rmsprop' / 'adagrad' / 'adadelta' / 'nag' vs the oracle's restatement,
    # 4 steps with weight decay and rescale_grad = 1 / batch_size
    for kind, p1 in (('rmsprop', 0.9), ('adagrad', 0.0), ('adadelta', 0.9), ('nag', 0.8)):
        w0 = rng.randn(300)
        rule = O.MXNetRule(kind, 0.05, p1=p1, wd=1e-2)
        wr = O.T(w0)
        wdev, s1, s2 = _dev(w0, dtype), torch.zeros(300, dtype=dtype).cuda(), torch.zeros(300, dtype=dtype).cuda()
        for t in range(4):
            g = rng.randn(300)
            wr = rule.step(wr, O.T(g), batch_size=4)
            ops.opt_step_(kind, wdev, _dev(g, dtype), s1, s2 if kind == 'adadelta' else None, 0.05, p1, rule.eps, wd=1e-2, rescale_grad=0.25)
        assert np.allclose(wdev.cpu().numpy(), wr.numpy(), rtol=tol * 20, atol=tol * 20), kind


def test_potrf_is_race_free_under_cu_contention():
    """Regression: every workgroup of the fused potrf panel kernel re-factors the diagonal block from the un-factored values while
    workgroup 0 writes the factor back in place; when other kernels hold the CUs (the SVGP step runs big GEMMs on side streams) the
    late workgroups used to read the already-factored block.  Factor repeatedly while another stream saturates the chip."""
    from mxfusion_amd import ops
    torch.manual_seed(3)
    n = 1024
    A = torch.randn(n, n, device='cuda', dtype=torch.float64)
    K = A @ A.T / n + torch.eye(n, device='cuda', dtype=torch.float64)
    ref = torch.linalg.cholesky(K)
    big_a = torch.randn(1, 4096, 4096, device='cuda')
    big_b = torch.randn(1, 4096, 16384, device='cuda')
    out = torch.empty(1, 4096, 16384, device='cuda')
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = 0
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.gemm(big_a, big_b, out=out)
        L, info = ops.potrf_(K[None].clone())
        torch.cuda.synchronize()
        if int(info.abs().sum()) != 0 or float((L[0] - ref).abs().max()) > 1e-11:
            bad += 1
    assert bad == 0, bad
