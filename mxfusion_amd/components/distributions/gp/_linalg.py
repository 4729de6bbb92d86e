"""autograd bridges over the dense linear algebra of the C ABI (mxf_potrf / mxf_trsm / mxf_trtri / mxf_gemm) with closed-form
reverse modes.  The log-pdf algorithms never differentiate through a Cholesky factor (their reverse modes are closed forms of the whole
expression); the reparameterised GP draws do, through CholFn."""
import math

import torch

from .... import ops


class CholLogPdfFn(torch.autograd.Function):
    """logL[s] = -c * sumlogdiag(chol(K[s])) - 1/2 (|L^-1 Y|^2 + N P log 2pi), c = P unless `logdet_mult` is given
    (gp.py:113-122, gp_regression.py:61-70).  Reverse mode: dK = 1/2 (alpha alpha^T - c K^-1), dY = -alpha, alpha = K^-1 Y.
    Also returns the factor L, L^-1 Y and the potrf info word."""

    @staticmethod
    def forward(ctx, K, Y, logdet_mult=None):
        SK, N, P = K.shape[0], K.shape[-1], Y.shape[-1]
        S = max(SK, Y.shape[0])                                   # either operand may be shared over the sample axis
        c = float(P if logdet_mult is None else logdet_mult)
        L, info = ops.potrf_(K.contiguous().clone())
        LinvY = ops.trsm_(L, Y.expand(S, N, P).contiguous().clone())
        logL = -c * ops.sumlogdiag(L) - 0.5 * ((LinvY ** 2).reshape(S, -1).sum(-1) + N * P * math.log(2 * math.pi))
        if any(ctx.needs_input_grad[:2]):
            Linv = ops.trtri(L)
            alpha = ops.gemm(Linv, LinvY, transA=True)
            dK = ops.gemm(alpha, alpha, transB=True, alpha=0.5)
            Kinv = ops.gemm(Linv, Linv, transA=True)
            dK = dK - (0.5 * c) * Kinv
            ctx.save_for_backward(dK, alpha)
            ctx.yshape, ctx.kshape = Y.shape, K.shape
        ctx.mark_non_differentiable(L, LinvY, info)
        return logL, L, LinvY, info

    @staticmethod
    def backward(ctx, g, *_):
        dK, alpha = ctx.saved_tensors
        gK = dK * g.reshape(-1, 1, 1)
        gY = -alpha * g.reshape(-1, 1, 1)
        if ctx.kshape[0] == 1 and gK.shape[0] > 1:
            gK = gK.sum(0, keepdim=True)
        if ctx.yshape[0] == 1 and gY.shape[0] > 1:
            gY = gY.sum(0, keepdim=True)
        return gK, gY, None


class CholFn(torch.autograd.Function):
    """L = chol(K) through mxf_potrf (lower, clean upper triangle), differentiable: with Gl = tril(dL) the reverse mode is
        dK = sym(L^-T Phi(L^T Gl) L^-1),   Phi(A) = tril(A) with a halved diagonal,   sym(S) = (S + S^T) / 2
    (the blocked form of Murray 2016, "Differentiation of the Cholesky decomposition") -- one mxf_trtri and three mxf_gemm.  Used where
    the reference differentiates through linalg.potrf and no closed form of the whole expression exists: the reparameterised draws
    L eps of the GP distributions and of the sampling algorithms (gp.py:124-153, cond_gp.py:185-223, gp_regression.py:92-135)."""

    @staticmethod
    def forward(ctx, K):
        L, info = ops.potrf_(K.contiguous().clone())
        ctx.save_for_backward(L)
        ctx.mark_non_differentiable(info)
        return L, info

    @staticmethod
    def backward(ctx, G, *_):
        L, = ctx.saved_tensors
        Linv = ops.trtri(L)
        P = torch.tril(ops.gemm(L, torch.tril(G).contiguous(), transA=True))
        P = P - 0.5 * torch.diag_embed(torch.diagonal(P, dim1=-2, dim2=-1))
        S = ops.gemm(ops.gemm(Linv, P, transA=True), Linv)
        return 0.5 * (S + S.transpose(-1, -2))


def chol(K):
    """(L, info): differentiable when K requires grad (CholFn), the plain mxf_potrf on a copy otherwise."""
    if _needs_grad(K):
        return CholFn.apply(K)
    return ops.potrf_(K.contiguous().clone())


class SpdInverseFn(torch.autograd.Function):
    """A = K^-1 for symmetric positive definite K (potrf + trtri + gemm); dK = -A G A."""

    @staticmethod
    def forward(ctx, K):
        L, info = ops.potrf_(K.clone())
        Linv = ops.trtri(L)
        A = ops.gemm(Linv, Linv, transA=True)
        ctx.save_for_backward(A)
        ctx.mark_non_differentiable(info)
        return A, info

    @staticmethod
    def backward(ctx, G, *_):
        A, = ctx.saved_tensors
        T = ops.gemm(A, G.contiguous())
        return ops.gemm(T, A, alpha=-1.0)


class MatmulFn(torch.autograd.Function):
    """C = op(A) op(B) through mxf_gemm (batched over the sample axis), with its two reverse-mode GEMMs."""

    @staticmethod
    def forward(ctx, A, B, transA, transB):
        ctx.tA, ctx.tB = transA, transB
        ctx.save_for_backward(A, B)
        return ops.gemm(A, B, transA=transA, transB=transB)

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        G = G.contiguous()
        tA, tB = ctx.tA, ctx.tB
        gA = gB = None
        if ctx.needs_input_grad[0]:
            gA = ops.gemm(B, G, transA=tB, transB=True) if tA else ops.gemm(G, B, transB=not tB)
            if A.shape[0] == 1 and gA.shape[0] > 1:
                gA = gA.sum(0, keepdim=True)
        if ctx.needs_input_grad[1]:
            gB = ops.gemm(G, A, transA=True, transB=tA) if tB else ops.gemm(A, G, transA=not tA)
            if B.shape[0] == 1 and gB.shape[0] > 1:
                gB = gB.sum(0, keepdim=True)
        return gA, gB, None, None


def matmul(A, B, transA=False, transB=False):
    return MatmulFn.apply(A, B, transA, transB)


class TrsmFn(torch.autograd.Function):
    """X = op(L)^-1 B through mxf_trsm (L lower triangular, shared over the sample axis when its leading extent is 1).
    Reverse mode: dB = op(L)^-T G (a second mxf_trsm); dL = -tril(dB X^T) (op = identity) or -tril(X dB^T) (op = transpose)."""

    @staticmethod
    def forward(ctx, L, B, transpose):
        ctx.transpose = bool(transpose)
        X = ops.trsm_(L, B.contiguous().clone(), transpose=transpose)
        ctx.save_for_backward(L, X)
        return X

    @staticmethod
    def backward(ctx, G):
        L, X = ctx.saved_tensors
        dB = ops.trsm_(L, G.contiguous().clone(), transpose=not ctx.transpose)
        dL = None
        if ctx.needs_input_grad[0]:
            dL = ops.gemm(X, dB, transB=True, alpha=-1.0) if ctx.transpose else ops.gemm(dB, X, transB=True, alpha=-1.0)
            dL = torch.tril(dL)
            if L.shape[0] == 1 and dL.shape[0] > 1:
                dL = dL.sum(0, keepdim=True)
        return dL, (dB if ctx.needs_input_grad[1] else None), None


class ColdotFn(torch.autograd.Function):
    """out[s, n] = sum_m A[s, m, n] B[s, m, n] through mxf_coldot; dA = B * g[:, None, :], dB = A * g[:, None, :]."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        return ops.coldot(A, B)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = g.unsqueeze(-2)
        gA = gB = None
        if ctx.needs_input_grad[0]:
            gA = B * g
            if A.shape[0] == 1 and gA.shape[0] > 1:
                gA = gA.sum(0, keepdim=True)
        if ctx.needs_input_grad[1]:
            gB = A * g
            if B.shape[0] == 1 and gB.shape[0] > 1:
                gB = gB.sum(0, keepdim=True)
        return gA, gB


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def trsm(L, B, transpose=False):
    """op(L)^-1 B; differentiable when an operand requires grad, the plain in-place-on-a-copy call otherwise."""
    if _needs_grad(L, B):
        return TrsmFn.apply(L, B, transpose)
    return ops.trsm_(L, B.contiguous().clone(), transpose=transpose)


def gemm(A, B, transA=False, transB=False):
    if _needs_grad(A, B):
        return MatmulFn.apply(A, B, transA, transB)
    return ops.gemm(A, B, transA=transA, transB=transB)


def coldot(A, B):
    if _needs_grad(A, B):
        return ColdotFn.apply(A, B)
    return ops.coldot(A, B)


class _InverseCache(object):
    """L^-1 (mxf_trtri) of a shared lower factor, kept while the factor tensor is unchanged (same storage, same version counter)."""

    def __init__(self):
        self._key, self._inv, self._held = None, None, None

    def get(self, L):
        key = (L.data_ptr(), L._version, tuple(L.shape), L.dtype)
        if key != self._key:
            with torch.no_grad():
                self._inv = ops.trtri(L.detach())
            self._key, self._held = key, L        # holding the factor keeps its storage from being recycled under the same address
        return self._inv


def solve_shared(L, B, cache):
    """L^-1 B for a factor shared by all samples inside a differentiable loop (the PILCO rollout: hundreds of solves against the SAME
    factor with one right-hand side per trajectory).  mxf_trsm is a chain of N / 64 dependent panel steps however few columns B has;
    with the cached explicit inverse the solve -- and its reverse pass -- is one GEMM each."""
    return gemm(cache.get(L), B)
