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
    _lib.call('mxf_gram', _h(X), KIND[kind] if isinstance(kind, str) else kind, _dt(X), S, N, N2, Q,
              _p(X), _ss(X), _p(X2), _ss(X2), _p(lengthscale), int(bool(ard)), _ss(lengthscale),
              _p(variance), _ss(variance), _p(diag_add), _ss(diag_add), float(jitter), mode,
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
