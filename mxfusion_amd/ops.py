"""Tensor-level wrappers over the C ABI (include/mxf_gp.h).  PyTorch is used for device memory and
streams only; every array computation is a HIP kernel in libmxf_gp.so."""
import torch

from . import _lib
from ._lib import F32, F64, WRITE, ACC_ADD, ACC_MUL  # noqa: F401

KIND = {'rbf': _lib.K_RBF, 'matern12': _lib.K_MATERN12, 'matern32': _lib.K_MATERN32,
        'matern52': _lib.K_MATERN52, 'linear': _lib.K_LINEAR, 'bias': _lib.K_BIAS, 'white': _lib.K_WHITE}


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.MXFError('mxfusion_amd ops need device (HIP) tensors; got a %s tensor. There is no CPU '
                            'fallback.' % t.device)


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError('mxfusion_amd supports float32/float64, got %s' % t.dtype)


def _h(t):
    _require_gpu(t)
    return _lib.handle(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _ss(t):
    """sample stride in elements (0 => broadcast over S); the per-sample block must be contiguous."""
    if t is None:
        return 0
    return 0 if t.shape[0] == 1 else t.stride(0)


def _c(t):
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def num_samples(*ts):
    return max([t.shape[0] for t in ts if t is not None] + [1])


def gram(kind, X, X2, lengthscale, variance, ard, diag_add=None, jitter=0.0, out=None, mode=WRITE):
    """K(X, X2) with the sample axis: X (S|1,N,Q), X2 (S|1,N2,Q) or None, lengthscale (S|1,Q|1),
    variance (S|1,1), diag_add (S|1,1) -> (S,N,N2).  Kernel.K (kernels/kernel.py:96-123)."""
    X, X2, lengthscale, variance, diag_add = _c(X), _c(X2), _c(lengthscale), _c(variance), _c(diag_add)
    S = num_samples(X, X2, lengthscale, variance, diag_add)
    N, Q = X.shape[-2], X.shape[-1]
    N2 = N if X2 is None else X2.shape[-2]
    if out is None:
        out = torch.empty((S, N, N2), dtype=X.dtype, device=X.device)
    if N == 0 or N2 == 0:       # empty operand: nothing to compute (an empty tensor has a null data pointer, which the ABI reads as "no X2")
        return out
    _lib.call('mxf_gram', _h(X), KIND[kind] if isinstance(kind, str) else kind, _dt(X), S, N, N2, Q,
              _p(X), _ss(X), _p(X2), _ss(X2), _p(lengthscale), int(bool(ard)), _ss(lengthscale),
              _p(variance), _ss(variance), _p(diag_add), _ss(diag_add), float(jitter), mode,
              _p(out), out.stride(-2), out.stride(0) if out.dim() == 3 else 0, _stream())
    return out


def gram2(kind1, kind2, op, X, X2, ls1, var1, ard1, ls2, var2, ard2, diag_add=None, jitter=0.0, out=None):
    """k1(X, X2) + k2(X, X2) (op = ACC_ADD) or their product (ACC_MUL) for two stationary kernels in ONE pass and one write (mxf_gram2):
    AddKernel / MultiplyKernel._compute_K (add_kernel.py:44-68, multiply_kernel.py:44-67).  Shapes as gram()."""
    X, X2, ls1, var1, ls2, var2, diag_add = _c(X), _c(X2), _c(ls1), _c(var1), _c(ls2), _c(var2), _c(diag_add)
    S = num_samples(X, X2, ls1, var1, ls2, var2, diag_add)
    N, Q = X.shape[-2], X.shape[-1]
    N2 = N if X2 is None else X2.shape[-2]
    if out is None:
        out = torch.empty((S, N, N2), dtype=X.dtype, device=X.device)
    if N == 0 or N2 == 0:
        return out
    k = lambda kk: KIND[kk] if isinstance(kk, str) else kk
    _lib.call('mxf_gram2', _h(X), k(kind1), k(kind2), op, _dt(X), S, N, N2, Q, _p(X), _ss(X), _p(X2), _ss(X2), _p(ls1), int(bool(ard1)), _ss(ls1),
              _p(var1), _ss(var1), _p(ls2), int(bool(ard2)), _ss(ls2), _p(var2), _ss(var2), _p(diag_add), _ss(diag_add), float(jitter),
              _p(out), out.stride(-2), out.stride(0) if out.dim() == 3 else 0, _stream())
    return out


def gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None):
    """Batched C = alpha op(A) op(B) + beta C on (S|1, m, k) operands -- linalg.gemm2 / syrk."""
    A, B = _c(A), _c(B)
    S = num_samples(A, B)
    M = A.shape[-1] if transA else A.shape[-2]
    K = A.shape[-2] if transA else A.shape[-1]
    N = B.shape[-2] if transB else B.shape[-1]
    if out is None:
        out = torch.empty((S, M, N), dtype=A.dtype, device=A.device)
    _lib.call('mxf_gemm', _h(A), _dt(A), int(transA), int(transB), M, N, K, float(alpha), _p(A), A.stride(-2), _ss(A),
              _p(B), B.stride(-2), _ss(B), float(beta), _p(out), out.stride(-2), out.stride(0), S, _stream())
    return out


def gemm_f32x3(A, B, alpha=1.0, beta=0.0, out=None, lower_only=False):
    """C = alpha A B^T + beta C for float32 A (M,K), B (N,K) on the bf16 matrix pipe with three-term splitting (f32-equivalent)."""
    A, B = _c(A), _c(B)
    if A.dtype != torch.float32 or B.dtype != torch.float32 or A.dim() != 2 or B.dim() != 2:
        raise ValueError('gemm_f32x3: 2-D float32 operands')
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.zeros((M, N), dtype=A.dtype, device=A.device) if lower_only else torch.empty((M, N), dtype=A.dtype, device=A.device)
    _lib.call('mxf_gemm_f32x3', _h(A), M, N, K, float(alpha), _p(A), A.stride(0), _p(B), B.stride(0), float(beta), _p(out), out.stride(0),
              int(bool(lower_only)), _stream())
    return out


def gemm_f16x2(A, B, alpha=1.0, beta=0.0, out=None, lower_only=False):
    """C = alpha A B^T + beta C for float32 A (M,K), B (N,K) on the f16 matrix pipe: two scaled f16 terms per operand, three products
    (f32-equivalent normwise; half the matrix-pipe work of gemm_f32x3)."""
    A, B = _c(A), _c(B)
    if A.dtype != torch.float32 or B.dtype != torch.float32 or A.dim() != 2 or B.dim() != 2:
        raise ValueError('gemm_f16x2: 2-D float32 operands')
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.zeros((M, N), dtype=A.dtype, device=A.device) if lower_only else torch.empty((M, N), dtype=A.dtype, device=A.device)
    _lib.call('mxf_gemm_f16x2', _h(A), M, N, K, float(alpha), _p(A), A.stride(0), _p(B), B.stride(0), float(beta), _p(out), out.stride(0),
              int(bool(lower_only)), _stream())
    return out


def f16x2_split(X):
    """The two scaled f16 planes of a 2-D float32 operand and its max-abs word (see gemm_f16x2_planes)."""
    X = _c(X)
    R, K = X.shape
    if X.dtype != torch.float32 or X.dim() != 2:
        raise ValueError('f16x2_split: 2-D float32 operand')
    n = _lib.load().mxf_f32x3_plane_elems(R, K)
    planes = torch.empty(2 * n, dtype=torch.int16, device=X.device)
    word = torch.zeros(1, dtype=torch.int32, device=X.device)
    _lib.call('mxf_f16x2_split', _h(X), R, K, _p(X), X.stride(0), _p(planes), _p(word), _stream())
    return planes, word


def gemm_f16x2_planes(A_split, B_split, M, N, K, alpha=1.0, beta=0.0, out=None, lower_only=False, blocked=False):
    """C = alpha A B^T + beta C from operands already split by f16x2_split (reuse across products).  blocked: the full product with C in
    16-column blocks, element (m, n) at ((n // 16) * M + m) * 16 + n % 16 of `out` (the layout the SVGP training step keeps T in)."""
    (pa, wa), (pb, wb) = A_split, B_split
    if out is None:
        out = torch.zeros((M, N), dtype=torch.float32, device=pa.device) if lower_only else torch.empty((M, N), dtype=torch.float32, device=pa.device)
    _lib.call('mxf_gemm_f16x2_planes', _h(pa), M, N, K, float(alpha), _p(pa), _p(wa), _p(pb), _p(wb), float(beta), _p(out), out.stride(0),
              2 if blocked else int(bool(lower_only)), _stream())
    return out


def gemm_f16x2_planes_kmajor(A_split, Bt_split, M, N, K, alpha=1.0, out=None, blocked=False, w=None):
    """C (M x N) = alpha A (M x K) Bt (K x N) from split operands, Bt_split = f16x2_split of the (K x N) matrix itself -- the product that lets
    the SVGP step form T = H0 Kuf from the planes of Kuf that Psi2 reads (mxf_gemm_f16x2_planes_kmajor).  w (K,) float32: also returns
    U[n] = sum_k w[k] Bt[k][n] from the same launch."""
    (pa, wa), (pb, wb) = A_split, Bt_split
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=pa.device)
    U = torch.empty(N, dtype=torch.float32, device=pa.device) if w is not None else None
    wc = _c(w) if w is not None else None
    _lib.call('mxf_gemm_f16x2_planes_kmajor', _h(pa), M, N, K, float(alpha), _p(pa), _p(wa), _p(pb), _p(wb), _p(out), int(bool(blocked)),
              _p(wc) if wc is not None else None, _p(U) if U is not None else None, _stream())
    return (out, U) if w is not None else out


def gemm_f16x2_planes_out(A_split, B_split, M, N, K, alpha=1.0, a_lower=False, transposed=False, a=None):
    """alpha A B^T written directly as the two f16 planes (unscaled hi + lo) of the (M x N) operand whose contraction index is its column
    (mxf_gemm_f16x2_planes_out: chained split products; M % 128 == 0, N % 256 == 0).  Returns the int16 planes tensor; transposed=True: also
    the planes of the (N x M) transpose from the same launch; with a (M,) float32 also U[n] = sum_m a[m] (hi + lo)(m, n)."""
    (pa, wa), (pb, wb) = A_split, B_split
    n = _lib.load().mxf_f32x3_plane_elems(M, N)
    out = torch.empty(2 * n, dtype=torch.int16, device=pa.device)
    transposed = transposed or a is not None
    outT = torch.empty(2 * _lib.load().mxf_f32x3_plane_elems(N, M), dtype=torch.int16, device=pa.device) if transposed else None
    U = torch.empty(N, dtype=torch.float32, device=pa.device) if a is not None else None
    _lib.call('mxf_gemm_f16x2_planes_out', _h(pa), M, N, K, float(alpha), _p(pa), _p(wa), _p(pb), _p(wb), _p(out), _p(outT),
              _p(_c(a)) if a is not None else None, _p(U), int(bool(a_lower)), _stream())
    if not transposed:
        return out
    return (out, outT) if U is None else (out, outT, U)


def f16x2_planes_transpose(planes, R, K, a=None, scale=None):
    """Planes of an (R x K) operand -> planes of its transpose (K x R) (mxf_f16x2_planes_transpose); with a (R,) and scale (1,) float32 also
    U[k] = scale * sum_r a[r] x(r, k).  Returns planes_T or (planes_T, U)."""
    n = _lib.load().mxf_f32x3_plane_elems(K, R)
    out = torch.empty(2 * n, dtype=torch.int16, device=planes.device)
    U = torch.empty(K, dtype=torch.float32, device=planes.device) if a is not None else None
    _lib.call('mxf_f16x2_planes_transpose', _h(planes), R, K, _p(planes), _p(out), _p(_c(a)) if a is not None else None,
              _p(_c(scale)) if scale is not None else None, _p(U), _stream())
    return out if U is None else (out, U)


def planes_to_dense(planes, R, K):
    """Decode two f16 planes (k16-blocked, element (r, k) at ((k // 16) * R + r) * 16 + k % 16) to the float64 matrix hi + lo (test helper)."""
    K16 = (K + 15) // 16
    n = K16 * R * 16
    v = planes.view(torch.float16)
    dec = lambda q: v[q * n:(q + 1) * n].view(K16, R, 16).permute(1, 0, 2).reshape(R, K16 * 16)[:, :K].double()
    return dec(0) + dec(1)


def f32x3_split(X):
    """Three-term bf16 split planes of a 2-D float32 matrix (operand format of gemm_f32x3_planes); returns an int16 tensor."""
    X = _c(X)
    if X.dtype != torch.float32 or X.dim() != 2:
        raise ValueError('f32x3_split: 2-D float32 operand')
    R, K = X.shape
    n = _lib.load().mxf_f32x3_plane_elems(R, K)
    planes = torch.empty(3 * n, dtype=torch.int16, device=X.device)
    _lib.call('mxf_f32x3_split', _h(X), R, K, _p(X), X.stride(0), _p(planes), _stream())
    return planes


def gemm_f32x3_planes(A_planes, B_planes, M, N, K, alpha=1.0, beta=0.0, out=None, lower_only=False):
    """C (M,N) = alpha A B^T + beta C from operands split with f32x3_split (A: (M,K), B: (N,K))."""
    if out is None:
        out = torch.zeros((M, N), dtype=torch.float32, device=A_planes.device) if lower_only else torch.empty((M, N), dtype=torch.float32, device=A_planes.device)
    _lib.call('mxf_gemm_f32x3_planes', _h(A_planes), M, N, K, float(alpha), _p(A_planes), _p(B_planes), float(beta), _p(out), out.stride(0),
              int(bool(lower_only)), _stream())
    return out


def gram_bwd(kind, X, X2, lengthscale, variance, ard, dK, need=('X', 'X2', 'ls', 'var')):
    """Reverse mode of gram(); returns (dX, dX2, dls, dvar) shaped like their primals (summed over S for
    broadcast operands)."""
    X, X2, lengthscale, variance, dK = _c(X), _c(X2), _c(lengthscale), _c(variance), _c(dK)
    S = dK.shape[0]
    N, Q = X.shape[-2], X.shape[-1]
    N2 = N if X2 is None else X2.shape[-2]
    dX = torch.zeros_like(X) if 'X' in need else None
    dX2 = torch.zeros_like(X2) if (X2 is not None and 'X2' in need) else None
    dls = torch.zeros_like(lengthscale) if 'ls' in need else None
    dvar = torch.zeros_like(variance) if 'var' in need else None
    _lib.call('mxf_gram_bwd', _h(X), KIND[kind] if isinstance(kind, str) else kind, _dt(X), S, N, N2, Q,
              _p(X), _ss(X), _p(X2), _ss(X2), _p(lengthscale), int(bool(ard)), _ss(lengthscale), _p(variance), _ss(variance),
              _p(dK), dK.stride(-2), dK.stride(0), _p(dX), _p(dX2), _p(dls), _p(dvar), _stream())
    return dX, dX2, dls, dvar


def potrf_(A, info=None):
    """in-place batched lower Cholesky of A (S,n,n); returns (A, info) with info an int32 device tensor."""
    S, n = A.shape[0], A.shape[-1]
    if info is None:
        info = torch.zeros(S, dtype=torch.int32, device=A.device)
    _lib.call('mxf_potrf', _h(A), _dt(A), S, n, _p(A), A.stride(-2), A.stride(0), _p(info), _stream())
    return A, info


def merge_info(*infos):
    """Combine LAPACK-style info words of several factorisations WITHOUT masking: a negative word (an internal failure of the library, e.g.
    a lost workgroup hand-off of the tile Cholesky) wins over everything, otherwise the largest positive one (first non-positive-definite
    leading minor); 0 only if all are 0.  (A plain sum turns -1 + 1 into "fine".)"""
    st = torch.stack(torch.broadcast_tensors(*[i.reshape(-1) for i in infos]))      # (one word per sample, or one for all of them)
    lo, hi = st.min(0).values, st.max(0).values
    return torch.where(lo < 0, lo, hi)


def check_info(info, what='potrf'):
    """Host sync + raise on a non-positive-definite matrix (MXNet raises lazily at the next blocking read); a NEGATIVE word is an internal
    failure of the factorisation kernels and always an error."""
    bad = info.nonzero()
    if bad.numel():
        s = int(bad[0, 0])
        v = int(info.reshape(-1)[s])
        if v < 0:
            raise _lib.MXFError('%s: internal failure of the factorisation of sample %d (info %d: a workgroup hand-off was lost)' % (what, s, v))
        raise _lib.MXFError('%s: matrix of sample %d is not positive definite (leading minor %d)' % (what, s, v))


def trsm_(L, B, transpose=False):
    """B <- op(L)^-1 B in place; L (S|1,n,n) lower, B (S,n,nrhs)."""
    L = _c(L)
    S, n, nrhs = B.shape[0], B.shape[-2], B.shape[-1]
    _lib.call('mxf_trsm', _h(B), _dt(B), int(bool(transpose)), S, n, nrhs, _p(L), L.stride(-2), _ss(L), _p(B), B.stride(-2),
              B.stride(0), _stream())
    return B


def trtri(L):
    L = _c(L)
    S, n = L.shape[0], L.shape[-1]
    out = torch.empty_like(L)
    _lib.call('mxf_trtri', _h(L), _dt(L), S, n, _p(L), L.stride(-2), L.stride(0), _p(out), out.stride(-2), out.stride(0), _stream())
    return out


def sumlogdiag(L):
    L = _c(L)
    S, n = L.shape[0], L.shape[-1]
    out = torch.empty(S, dtype=L.dtype, device=L.device)
    _lib.call('mxf_sumlogdiag', _h(L), _dt(L), S, n, _p(L), L.stride(-2), L.stride(0), _p(out), _stream())
    return out


def make_diagonal(a):
    """(..., n) -> (..., n, n) diagonal embed (mxf_make_diagonal)."""
    a = _c(a)
    n = a.shape[-1]
    batch = a.numel() // n if n else 0
    out = torch.empty(tuple(a.shape) + (n,), dtype=a.dtype, device=a.device)
    _lib.call('mxf_make_diagonal', _h(a), _dt(a), batch, n, _p(a), _p(out), _stream())
    return out


def diag_of(g):
    """(..., n, n) -> (..., n) diagonal (mxf_diag_of)."""
    g = _c(g)
    n = g.shape[-1]
    batch = g.numel() // (n * n) if n else 0
    out = torch.empty(tuple(g.shape[:-1]), dtype=g.dtype, device=g.device)
    _lib.call('mxf_diag_of', _h(g), _dt(g), batch, n, _p(g), _p(out), _stream())
    return out


def softplus(x):
    x = _c(x)
    y = torch.empty_like(x)
    _lib.call('mxf_softplus_fwd', _h(x), _dt(x), x.numel(), _p(x), _p(y), _stream())
    return y


def softplus_bwd_(x, dy, dx_acc):
    _lib.call('mxf_softplus_bwd', _h(x), _dt(x), x.numel(), _p(_c(x)), _p(_c(dy)), _p(dx_acc), _stream())
    return dx_acc


def normal_reparam(mean, var, eps):
    """x[s] = mean + eps[s]*sqrt(var); mean/var (n...), eps (S, n...)."""
    mean, var, eps = _c(mean), _c(var), _c(eps)
    x = torch.empty_like(eps)
    _lib.call('mxf_normal_reparam', _h(eps), _dt(eps), eps.shape[0], mean.numel(), _p(mean), _p(var), _p(eps), _p(x), _stream())
    return x


def normal_reparam_bwd_(var, eps, dx, dmean_acc, dvar_acc):
    _lib.call('mxf_normal_reparam_bwd', _h(eps), _dt(eps), eps.shape[0], var.numel(), _p(_c(var)), _p(_c(eps)), _p(_c(dx)),
              _p(dmean_acc), _p(dvar_acc), _stream())


def normal_logpdf_(x, mean, var, scale, out_acc, dx_acc=None, dmean_acc=None, dvar_acc=None):
    """out_acc += scale * sum_{s,i} log N(x[s,i]|mean[i],var[i]) (+ reverse mode into the *_acc buffers)."""
    x, mean, var = _c(x), _c(mean), _c(var)
    S = x.shape[0]
    n = x.numel() // S
    _lib.call('mxf_normal_logpdf', _h(x), _dt(x), S, n, _p(x), _p(mean), mean.numel(), _p(var), var.numel(), float(scale),
              _p(out_acc), _p(dx_acc), _p(dmean_acc), _p(dvar_acc), _stream())
    return out_acc


def adam_step_(w, g, m, v, lr, t, beta1=0.9, beta2=0.999, epsilon=1e-8, rescale_grad=1.0):
    _lib.call('mxf_adam_step', _h(w), _dt(w), w.numel(), _p(w), _p(g), _p(m), _p(v), float(lr), float(beta1), float(beta2),
              float(epsilon), float(rescale_grad), int(t), _stream())


def uniform_sum(g):
    """sum(g) if all entries of the (small) vector g agree to 1e-6 relative, NaN otherwise -- one launch, no host sync (mxf_uniform_sum)."""
    g = _c(g)
    out = torch.empty((), dtype=g.dtype, device=g.device)
    _lib.call('mxf_uniform_sum', _h(g), _dt(g), g.numel(), _p(g), _p(out), _stream())
    return out


def sgd_step_(w, g, mom, lr, momentum=0.0, wd=0.0, rescale_grad=1.0):
    """MXNet SGD on a flat buffer (mxf_sgd_step); mom = None for plain SGD."""
    _lib.call('mxf_sgd_step', _h(w), _dt(w), w.numel(), _p(w), _p(g), None if mom is None else _p(mom), float(lr), float(momentum), float(wd),
              float(rescale_grad), _stream())


OPT_KIND = {'rmsprop': 1, 'adagrad': 2, 'adadelta': 3, 'nag': 4}


def opt_step_(kind, w, g, s1, s2, lr, p1, epsilon, wd=0.0, rescale_grad=1.0):
    """MXNet 'rmsprop' / 'adagrad' / 'adadelta' / 'nag' on a flat buffer (mxf_opt_step); s1, s2: the rule's state buffers (s2: adadelta only)."""
    _lib.call('mxf_opt_step', _h(w), OPT_KIND[kind], _dt(w), w.numel(), _p(w), _p(g), _p(s1), None if s2 is None else _p(s2), float(lr), float(p1),
              float(epsilon), float(wd), float(rescale_grad), _stream())


def gp_logpdf(kind, X, Y, noise_var, lengthscale, variance, ard, jitter=0.0, want_grad=False):
    """GPRegressionLogPdf.compute (gp_regression.py:42-76).  X (S|1,N,Q), Y (S|1,N,P) [minus mean], noise_var (S|1,1),
    lengthscale (S|1,Q|1), variance (S|1,1).  Returns dict(logL (S,), L (S,N,N), LinvY (S,N,P), info, grads...)."""
    X, Y, noise_var, lengthscale, variance = _c(X), _c(Y), _c(noise_var), _c(lengthscale), _c(variance)
    S = num_samples(X, Y, noise_var, lengthscale, variance)
    N, Q, P = X.shape[-2], X.shape[-1], Y.shape[-1]
    dev, dt = X.device, X.dtype
    out = {'logL': torch.empty(S, dtype=dt, device=dev), 'L': torch.empty((S, N, N), dtype=dt, device=dev),
           'LinvY': torch.empty((S, N, P), dtype=dt, device=dev), 'info': torch.zeros(S, dtype=torch.int32, device=dev)}
    g = {}
    if want_grad:
        g = {'dX': torch.empty((S, N, Q), dtype=dt, device=dev), 'dY': torch.empty((S, N, P), dtype=dt, device=dev),
             'dnoise': torch.empty((S, 1), dtype=dt, device=dev),
             'dls': torch.empty((S, lengthscale.shape[-1]), dtype=dt, device=dev), 'dvar': torch.empty((S, 1), dtype=dt, device=dev)}
    _lib.call('mxf_gp_logpdf', _h(X), KIND[kind], _dt(X), S, N, Q, P, _p(X), _ss(X), _p(Y), _ss(Y), _p(noise_var), _ss(noise_var),
              _p(lengthscale), int(bool(ard)), _ss(lengthscale), _p(variance), _ss(variance), float(jitter),
              _p(out['logL']), _p(out['L']), _p(out['LinvY']), _p(out['info']), int(want_grad),
              _p(g.get('dX')), _p(g.get('dY')), _p(g.get('dnoise')), _p(g.get('dls')), _p(g.get('dvar')), _stream())
    out.update(g)
    return out


def svgp_logpdf(kind, X, Y, Z, noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance, ard, jitter=0.0,
                scaling=1.0, gscale=1.0, want_grad=False):
    """SVGPRegressionLogPdf.compute (svgp_regression.py:43-109).  X (S|1,B,Q), Y (S|1,B,P) [minus mean],
    Z (M,Q), noise_var (1,) | (P,) | (B,1) | (B,P), qU_mean (M,P), qU_cov_W (M,M), qU_cov_diag (M,) [positive], lengthscale (Q|1,), variance (1,).
    Returns dict(logL (S,), info, and -- if want_grad -- the gradients of gscale*sum_s logL[s])."""
    X, Y, Z = _c(X), _c(Y), _c(Z)
    noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance = [_c(t) for t in (noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance)]
    S = num_samples(X, Y)
    B, Q, P, M = X.shape[-2], X.shape[-1], Y.shape[-1], Z.shape[-2]
    dev, dt = X.device, X.dtype
    out = {'logL': torch.empty(S, dtype=dt, device=dev), 'info': torch.zeros(1, dtype=torch.int32, device=dev)}
    # noise_var: (1,) homoscedastic -> streaming fused path; (P,), (B,1) or (B,P) -> mxf_svgp_logpdf_het (svgp_regression.py:61-67)
    if noise_var.dim() == 1:
        noise_var = noise_var.reshape(1, -1)
    nrows, ncols = noise_var.shape
    if nrows not in (1, B) or ncols not in (1, P):
        raise ValueError('svgp_logpdf: noise_var must be (1|B, 1|P), got %s' % (tuple(noise_var.shape),))
    het = nrows * ncols > 1
    g = {}
    if want_grad:
        E = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
        g = {'dX': E(*X.shape), 'dY': E(*Y.shape), 'dZ': E(M, Q), 'dnoise': E(nrows, ncols), 'dmu': E(M, P), 'dW': E(M, M), 'dSdiag': E(M),
             'dls': E(lengthscale.numel()), 'dvar': E(1)}
    tail = (_p(qU_mean), _p(qU_cov_W), _p(qU_cov_diag), _p(lengthscale), int(bool(ard)), _p(variance), float(jitter),
            float(scaling), float(gscale), _p(out['logL']), _p(out['info']), int(want_grad),
            _p(g.get('dX')), _p(g.get('dY')), _p(g.get('dZ')), _p(g.get('dnoise')), _p(g.get('dmu')), _p(g.get('dW')),
            _p(g.get('dSdiag')), _p(g.get('dls')), _p(g.get('dvar')), _stream())
    head = (_h(X), KIND[kind], _dt(X), S, B, M, Q, P, _p(X), _ss(X), _p(Y), _ss(Y), _p(Z), _p(noise_var))
    if het:
        _lib.call('mxf_svgp_logpdf_het', *head, nrows, ncols, *tail)
    else:
        _lib.call('mxf_svgp_logpdf', *head, *tail)
    out.update(g)
    return out


def svgp_logpdf_sampled(kind, X, Y, Z, noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance, ard, jitter=0.0, scaling=1.0,
                        gscale=1.0, want_grad=False):
    """mxf_svgp_logpdf_sampled: the homoscedastic bound with any operand sampled.  Every operand carries a leading sample axis of size S or 1:
    X (S|1,B,Q) Y (S|1,B,P) Z (S|1,M,Q) noise_var (S|1,1) qU_mean (S|1,M,P) qU_cov_W (S|1,M,M) qU_cov_diag (S|1,M) lengthscale (S|1,Q|1)
    variance (S|1,1).  Returns logL (S,), info (S,) and -- if want_grad -- per-sample gradients (S, ...) of gscale * logL[s]."""
    ops_in = [_c(t) for t in (X, Y, Z, noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance)]
    X, Y, Z, noise_var, qU_mean, qU_cov_W, qU_cov_diag, lengthscale, variance = ops_in
    S = max(t.shape[0] for t in ops_in)
    if any(t.shape[0] not in (1, S) for t in ops_in):
        raise ValueError('svgp_logpdf_sampled: sample axes must be 1 or %d' % S)
    B, Q, P, M = X.shape[-2], X.shape[-1], Y.shape[-1], Z.shape[-2]
    if noise_var.numel() != noise_var.shape[0]:
        raise ValueError('svgp_logpdf_sampled: homoscedastic noise (S|1, 1) only')
    dev, dt = X.device, X.dtype
    st = lambda t: 0 if t.shape[0] == 1 else t[0].numel()
    lsn = lengthscale[0].numel()
    out = {'logL': torch.empty(S, dtype=dt, device=dev), 'info': torch.zeros(S, dtype=torch.int32, device=dev)}
    g = {}
    if want_grad:
        E = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
        g = {'dX': E(S, B, Q), 'dY': E(S, B, P), 'dZ': E(S, M, Q), 'dnoise': E(S, 1), 'dmu': E(S, M, P), 'dW': E(S, M, M), 'dSdiag': E(S, M),
             'dls': E(S, lsn), 'dvar': E(S, 1)}
    _lib.call('mxf_svgp_logpdf_sampled', _h(X), KIND[kind], _dt(X), S, B, M, Q, P, _p(X), st(X), _p(Y), st(Y), _p(Z), st(Z), _p(noise_var), st(noise_var),
              _p(qU_mean), st(qU_mean), _p(qU_cov_W), st(qU_cov_W), _p(qU_cov_diag), st(qU_cov_diag), _p(lengthscale), int(bool(ard)), st(lengthscale),
              _p(variance), st(variance), float(jitter), float(scaling), float(gscale), _p(out['logL']), _p(out['info']), int(want_grad),
              _p(g.get('dX')), _p(g.get('dY')), _p(g.get('dZ')), _p(g.get('dnoise')), _p(g.get('dmu')), _p(g.get('dW')), _p(g.get('dSdiag')),
              _p(g.get('dls')), _p(g.get('dvar')), _stream())
    out.update(g)
    return out


def _device_index(device):
    """CUDA device index of `device` (None / torch.device('cuda') without an index -> the current device; int / torch.device / str)."""
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, str):
        device = torch.device(device)
    if isinstance(device, torch.device):
        if device.type != 'cuda':
            raise ValueError('mxfusion_amd: %r is not a GPU device' % (device,))
        return device.index if device.index is not None else torch.cuda.current_device()
    return int(device)


def svgp_cond_nowait(device=None, reset=False):
    """Running maximum of the condition numbers published by the finished svgp_logpdf training calls of this thread on `device`; does not
    synchronise (mxf_svgp_cond_nowait)."""
    return _lib.svgp_cond_nowait(_device_index(device), reset)


def svgp_last_cond(device=None):
    """1-norm condition number of Kuu + jitter I seen by the last svgp_logpdf training call of this thread on `device` (mxf_svgp_last_cond;
    synchronises).  The float32 streaming form is valid up to ~3e3 (include/mxf_gp.h)."""
    idx = _device_index(device)
    return _lib.svgp_last_cond(idx)


# ---- gradient exchange through the C ABI (RCCL); mxfusion_amd's own loops use torch.distributed, these mirror what a reference-side
# binder without PyTorch calls (include/mxf_gp.h) -------------------------------------------------------------
def comm_unique_id(device=None):
    """128-byte rendezvous id (rank 0 creates it and hands it to the other ranks)."""
    import ctypes
    idx = _device_index(device)
    buf = ctypes.create_string_buffer(128)
    _lib.call('mxf_comm_unique_id', _lib.handle(idx), ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw


def comm_init(nranks, rank, unique_id, device=None):
    import ctypes
    idx = _device_index(device)
    assert len(unique_id) == 128
    buf = ctypes.create_string_buffer(bytes(unique_id), 128)
    _lib.call('mxf_comm_init', _lib.handle(idx), int(nranks), int(rank), ctypes.cast(buf, ctypes.c_void_p))


def comm_destroy(device=None):
    idx = _device_index(device)
    _lib.call('mxf_comm_destroy', _lib.handle(idx))


def allreduce_sum_(t):
    """In-place sum over the ranks of the handle's communicator (mxf_allreduce_sum), ordered on the current stream."""
    assert t.is_contiguous()
    _lib.call('mxf_allreduce_sum', _h(t), _dt(t), _p(t), t.numel(), _stream())
    return t


def bcast_(t, root=0):
    assert t.is_contiguous()
    _lib.call('mxf_bcast', _h(t), _dt(t), _p(t), t.numel(), int(root), _stream())
    return t


def svgp_logpdf_mat(Kuu, Kuf, Kdiag, Y, noise_var, qU_mean, qU_cov_W, qU_cov_diag, jitter=0.0, scaling=1.0, gscale=1.0, want_grad=False):
    """SVGP bound from materialised Grams: Kuu (M,M) without jitter, Kuf (M,B), Kdiag (B,), Y (B,P) or (S,B,P) [S samples of the outputs over
    the same inputs], noise_var (1,) | (P,) | (B,1) | (B,P).  Returns dict(logL (S,), info, and -- if want_grad -- dKuu, dKuf, dKdiag, dY,
    dnoise, dmu, dW, dSdiag of gscale * sum_s logL[s])."""
    Kuu, Kuf, Kdiag, Y, noise_var, qU_mean, qU_cov_W, qU_cov_diag = [_c(t) for t in (Kuu, Kuf, Kdiag, Y, noise_var, qU_mean, qU_cov_W, qU_cov_diag)]
    M, B, P = Kuf.shape[-2], Kuf.shape[-1], Y.shape[-1]
    S = Y.shape[0] if Y.dim() == 3 else 1
    dev, dt = Kuf.device, Kuf.dtype
    if noise_var.dim() == 1:
        noise_var = noise_var.reshape(1, -1)
    nrows, ncols = noise_var.shape
    if nrows not in (1, B) or ncols not in (1, P):
        raise ValueError('svgp_logpdf_mat: noise_var must be (1|B, 1|P), got %s' % (tuple(noise_var.shape),))
    out = {'logL': torch.empty(S, dtype=dt, device=dev), 'info': torch.zeros(1, dtype=torch.int32, device=dev)}
    g = {}
    if want_grad:
        E = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
        g = {'dKuu': E(M, M), 'dKuf': E(M, B), 'dKdiag': E(B), 'dY': E(*Y.shape), 'dnoise': E(nrows, ncols), 'dmu': E(M, P), 'dW': E(M, M),
             'dSdiag': E(M)}
    _lib.call('mxf_svgp_logpdf_mat', _h(Kuf), _dt(Kuf), S, B, M, P, _p(Kuu), _p(Kuf), _p(Kdiag), _p(Y), B * P if S > 1 else 0, _p(noise_var),
              nrows, ncols, _p(qU_mean), _p(qU_cov_W), _p(qU_cov_diag), float(jitter), float(scaling), float(gscale), _p(out['logL']),
              _p(out['info']), int(want_grad), _p(g.get('dKuu')), _p(g.get('dKuf')), _p(g.get('dKdiag')), _p(g.get('dY')), _p(g.get('dnoise')),
              _p(g.get('dmu')), _p(g.get('dW')), _p(g.get('dSdiag')), _stream())
    out.update(g)
    return out


def coldot(A, B):
    """out[s, n] = sum_m A[s, m, n] * B[s, m, n]  (F.sum(A*B, axis=-2))."""
    A, B = _c(A), _c(B)
    S, M, N = max(A.shape[0], B.shape[0]), A.shape[-2], A.shape[-1]
    out = torch.empty((S, N), dtype=A.dtype, device=A.device)
    _lib.call('mxf_coldot', _h(A), _dt(A), S, M, N, _p(A), A.stride(-2), _ss(A), _p(B), B.stride(-2), _ss(B), _p(out), _stream())
    return out


def kdiag(kind, X, lengthscale, variance, ard):
    """Kernel.Kdiag through the C ABI (mxf_kdiag): X (S,N,Q) -> (S,N).  For 'linear', `lengthscale` carries the variances."""
    X = _c(X)
    S, N, Q = X.shape
    ls = None if lengthscale is None else _c(lengthscale).reshape(-1, _c(lengthscale).shape[-1])
    var = None if variance is None else _c(variance).reshape(-1)
    out = torch.empty((S, N), dtype=X.dtype, device=X.device)
    sls = 0 if ls is None or ls.shape[0] == 1 else ls.shape[-1]
    svar = 0 if var is None or var.numel() == 1 else 1
    _lib.call('mxf_kdiag', _h(X), KIND[kind], _dt(X), S, N, Q, _p(X), N * Q, _p(ls), int(bool(ard)), sls, _p(var), svar, _p(out), _stream())
    return out


def gp_predict(kind, X_cond, X_test, lengthscale, variance, ard, L, LinvY, noise_var, noise_free=False, full_cov=False):
    """GPRegressionMeanVariancePrediction.compute (gp_regression.py:146-196) as one C-ABI call (mxf_gp_predict): one posterior
    (L (N,N), LinvY (N,P)), X_test (S,Nt,Q) -> mean (S,Nt,P), var (S,Nt) or (S,Nt,Nt)."""
    X_cond, X_test, lengthscale, variance, L, LinvY = [_c(t) for t in (X_cond, X_test, lengthscale, variance, L, LinvY)]
    S, Nt, Q = X_test.shape
    N, P = X_cond.shape[-2], LinvY.shape[-1]
    mean = torch.empty((S, Nt, P), dtype=X_test.dtype, device=X_test.device)
    var = torch.empty((S, Nt, Nt) if full_cov else (S, Nt), dtype=X_test.dtype, device=X_test.device)
    _lib.call('mxf_gp_predict', _h(X_test), KIND[kind], _dt(X_test), S, N, Nt, Q, P, _p(X_cond), _p(X_test), _p(lengthscale), int(bool(ard)),
              _p(variance), _p(L), L.stride(-2), _p(LinvY), None if noise_var is None else _p(_c(noise_var)), int(bool(noise_free)),
              int(bool(full_cov)), _p(mean), _p(var), _stream())
    return mean, var


def svgp_predict(kind, Z, X_test, lengthscale, variance, ard, qU_mean, qU_cov_W, qU_cov_diag, noise_var, jitter=0.0, noise_free=False,
                 full_cov=False):
    """SVGPRegressionMeanVariancePrediction.compute (svgp_regression.py:121-189) as one C-ABI call (mxf_svgp_predict).
    Returns mean (S,Nt,P), var (S,Nt) or (S,Nt,Nt), info (2,) int32."""
    Z, X_test, lengthscale, variance, qU_mean, qU_cov_W, qU_cov_diag = [_c(t) for t in (Z, X_test, lengthscale, variance, qU_mean, qU_cov_W,
                                                                                        qU_cov_diag)]
    S, Nt, Q = X_test.shape
    M, P = Z.shape[-2], qU_mean.shape[-1]
    mean = torch.empty((S, Nt, P), dtype=X_test.dtype, device=X_test.device)
    var = torch.empty((S, Nt, Nt) if full_cov else (S, Nt), dtype=X_test.dtype, device=X_test.device)
    info = torch.zeros(2, dtype=torch.int32, device=X_test.device)
    _lib.call('mxf_svgp_predict', _h(X_test), KIND[kind], _dt(X_test), S, M, Nt, Q, P, _p(Z), _p(X_test), _p(lengthscale), int(bool(ard)),
              _p(variance), _p(qU_mean), _p(qU_cov_W), _p(qU_cov_diag), None if noise_var is None else _p(_c(noise_var)), float(jitter),
              int(bool(noise_free)), int(bool(full_cov)), _p(mean), _p(var), _p(info), _stream())
    return mean, var, info


def sgp_logpdf(kind, X, Y, Z, noise_var, lengthscale, variance, ard, jitter=0.0, gscale=1.0, want_grad=False):
    """SparseGPRegressionLogPdf.compute (sparsegp_regression.py:42-108) for ONE sample: X (B,Q), Y (B,P), Z (M,Q), noise_var (1,),
    lengthscale (Q|1,), variance (1,).  Returns dict(logL (1,), wv (M,P), L (M,M), LA (M,M), info, gradients...)."""
    X, Y, Z, noise_var, lengthscale, variance = [_c(t) for t in (X, Y, Z, noise_var, lengthscale, variance)]
    B, Q, P, M = X.shape[-2], X.shape[-1], Y.shape[-1], Z.shape[-2]
    dev, dt = X.device, X.dtype
    E = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
    out = {'logL': E(1), 'wv': E(M, P), 'L': E(M, M), 'LA': E(M, M), 'info': torch.zeros(1, dtype=torch.int32, device=dev)}
    g = {}
    if want_grad:
        g = {'dX': E(B, Q), 'dY': E(B, P), 'dZ': E(M, Q), 'dnoise': E(1), 'dls': E(lengthscale.numel()), 'dvar': E(1)}
    _lib.call('mxf_sgp_logpdf', _h(X), KIND[kind], _dt(X), B, M, Q, P, _p(X), _p(Y), _p(Z), _p(noise_var), _p(lengthscale),
              int(bool(ard)), _p(variance), float(jitter), float(gscale), _p(out['logL']), _p(out['wv']), _p(out['L']), _p(out['LA']),
              _p(out['info']), int(want_grad), _p(g.get('dX')), _p(g.get('dY')), _p(g.get('dZ')), _p(g.get('dnoise')), _p(g.get('dls')),
              _p(g.get('dvar')), _stream())
    out.update(g)
    return out
