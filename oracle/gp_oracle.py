"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

An op-for-op float64 restatement (PyTorch-CPU tensors, so that ``torch.autograd``
also yields reference gradients) of the Gaussian-process + SVI hot path of
amzn/MXFusion v0.3.1.  Every function cites the reference ``file:line`` it follows
(paths relative to the reference repo root).  Arrays carry the reference's leading
sample axis ``S`` exactly as the reference does (``(S, N, Q)``, ``(S, N, N)`` ...).

Who may use this file: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* / reported CPU baseline,
never as the thing measured or shipped.  The product package ``mxfusion_amd`` must
not import it (``tests/test_layout.py`` enforces this).

Parity pin status
-----------------
The reference is Python on Apache MXNet (>=1.3, un-vendored) and neither MXNet nor
GPy (the reference tests' own oracle) is installable here, so the reference cannot
be executed.  The oracle is pinned by:

* **Reference-recorded outputs** (true MXFusion-on-MXNet outputs stored in the
  reference repo): the 10-checkpoint loss trajectory of 100 Adam iterations printed
  in ``examples/notebooks/gp_regression.ipynb`` cell 12 and the learned parameters in
  cell 14 -- pins a1,a2,a4,a7,a19,a20 (Gram, exact-GP log-pdf, softplus, MAP, batch
  loop, MXNet-Adam) end to end; the GPy optimum of cell 16.
  See ``tests/golden/make_golden.py`` / ``tests/test_oracle_golden.py``.
* **Independent closed forms** on the reference tests' own seeded inputs
  (``testing/modules/*_test.py``): SciPy ``multivariate_normal.logpdf`` for the exact
  GP, the Hensman-2013 bound written with explicit inverses for SVGP, the Titsias
  bound via ``log N(y|0,Qff+s2 I)`` for the sparse GP.  These pin a10/a11/a13
  *values* but not by executing reference code: for the SVGP/SGP rows the parity
  status is "pinned to closed forms on the reference tests' inputs, reference
  execution unavailable".
"""
import math

import numpy as np
import torch

DT = torch.float64
LOG2PI = math.log(2.0 * math.pi)


def T(a, dtype=DT):
    """numpy/python -> torch tensor (no copy when already a tensor of that dtype)."""
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a), dtype=dtype)


# ----------------------------------------------------------------------------
# Sample-axis helpers: mxfusion/components/variables/runtime_variable.py
# ----------------------------------------------------------------------------
def add_sample_dimension(a):
    """runtime_variable.py:20-31 -- F.expand_dims(array, axis=0)."""
    return a.unsqueeze(0)


def expectation(a):
    """runtime_variable.py:53-60 -- F.mean(array, axis=0)."""
    return a.mean(dim=0)


def _num_samples(a):
    if isinstance(a, dict):
        return max(v.shape[0] for v in a.values())
    return a.shape[0]


def _as_samples(a, n):
    # runtime_variable.py:82-99: broadcast_axis(axis=0,size=n) unless already S>1
    if a.shape[0] > 1:
        return a
    return a.expand((n,) + tuple(a.shape[1:])).clone()  # materialised, like MXNet


def arrays_as_samples(arrays):
    """runtime_variable.py:102-118."""
    mx = max(_num_samples(a) for a in arrays)
    if mx > 1:
        return [{k: _as_samples(v, mx) for k, v in a.items()} if isinstance(a, dict)
                else _as_samples(a, mx) for a in arrays]
    return arrays


# ----------------------------------------------------------------------------
# Positive transformation: mxfusion/components/variables/var_trans.py:63-91
# ----------------------------------------------------------------------------
def softplus(x):
    """var_trans.py:75 -- Activation(softrelu) = log(1+exp(x)), offset 0."""
    return torch.nn.functional.softplus(x, beta=1.0, threshold=1e6)


def inv_softplus(y):
    """var_trans.py:91 -- log(expm1(y))."""
    return torch.log(torch.expm1(y))


# ----------------------------------------------------------------------------
# Kernels: mxfusion/components/distributions/gp/kernels/*.py
# ----------------------------------------------------------------------------
def syrk(a, transpose=False):
    """MXNet linalg.syrk: A A^T (transpose=False) or A^T A; full symmetric output."""
    return a.transpose(-1, -2) @ a if transpose else a @ a.transpose(-1, -2)


def gemm2(a, b, ta=False, tb=False):
    """MXNet linalg.gemm2."""
    a = a.transpose(-1, -2) if ta else a
    b = b.transpose(-1, -2) if tb else b
    return a @ b


def potrf(a):
    """MXNet linalg.potrf: lower Cholesky, batched over leading axes."""
    return torch.linalg.cholesky(a)


def trsm(L, B, transpose=False):
    """MXNet linalg.trsm(A,B,transpose): solves op(A) X = B, A lower triangular."""
    if transpose:
        return torch.linalg.solve_triangular(L.transpose(-1, -2), B, upper=True)
    return torch.linalg.solve_triangular(L, B, upper=False)


def trmm(L, B):
    """MXNet linalg.trmm: L B with L lower triangular."""
    return torch.tril(L) @ B


def sumlogdiag(L):
    """MXNet linalg.sumlogdiag."""
    return torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)


def make_diagonal(d):
    """util/customop.py:22-57 -- (...,M) -> (...,M,M) diagonal embed."""
    return torch.diag_embed(d)


class Kernel(object):
    """kernels/kernel.py:96-147 -- strips the ``<name>_`` prefix, applies active_dims."""
    name = 'kern'

    def __init__(self, input_dim, name, active_dims=None):
        self.input_dim = input_dim
        self.name = name
        self.active_dims = active_dims

    def _fetch(self, kernel_params):
        off = len(self.name) + 1
        return {k[off:]: v for k, v in kernel_params.items() if k.startswith(self.name + '_')}

    def _slice(self, X):
        if self.active_dims is not None and X is not None:
            return X[..., list(self.active_dims)]
        return X

    def K(self, X, X2=None, **kernel_params):
        return self._compute_K(self._slice(X), X2=self._slice(X2), **self._fetch(kernel_params))

    def Kdiag(self, X, **kernel_params):
        return self._compute_Kdiag(self._slice(X), **self._fetch(kernel_params))

    def param_names(self):
        raise NotImplementedError

    def __add__(self, other):
        return AddKernel([self, other])

    def __mul__(self, other):
        return MultiplyKernel([self, other])


class Stationary(Kernel):
    def __init__(self, input_dim, ARD=False, name='stationary', active_dims=None):
        super().__init__(input_dim, name, active_dims)
        self.ARD = ARD

    def param_names(self):
        return [self.name + '_lengthscale', self.name + '_variance']

    def _compute_R2(self, X, lengthscale, variance, X2=None):
        """kernels/stationary.py:74-107 (expansion form, no clipping)."""
        lengthscale = lengthscale.unsqueeze(-2)
        if X2 is None:
            xsc = X / lengthscale
            amat = syrk(xsc) * -2
            dg_a = (xsc ** 2).sum(-1)
            amat = amat + dg_a.unsqueeze(-1)
            amat = amat + dg_a.unsqueeze(-2)
        else:
            x1sc = X / lengthscale
            x2sc = X2 / lengthscale
            amat = gemm2(x1sc, x2sc, False, True) * -2
            dg1 = (x1sc ** 2).sum(-1, keepdim=True)
            amat = amat + dg1
            dg2 = (x2sc ** 2).sum(-1).unsqueeze(-2)
            amat = amat + dg2
        return amat

    def _compute_Kdiag(self, X, lengthscale, variance):
        """kernels/stationary.py:109-124 -- zeros(X.shape[:-1]) + variance."""
        return torch.zeros(X.shape[:-1], dtype=X.dtype) + variance


class RBF(Stationary):
    def __init__(self, input_dim, ARD=False, name='rbf', active_dims=None):
        super().__init__(input_dim, ARD, name, active_dims)

    def _compute_K(self, X, lengthscale, variance, X2=None):
        """kernels/rbf.py:71-72."""
        R2 = self._compute_R2(X, lengthscale, variance, X2=X2)
        return torch.exp(R2 / -2) * variance.unsqueeze(-1)


class Matern52(Stationary):
    def __init__(self, input_dim, ARD=False, name='matern52', active_dims=None):
        super().__init__(input_dim, ARD, name, active_dims)

    def _compute_K(self, X, lengthscale, variance, X2=None):
        """kernels/matern.py:84-88."""
        R2 = self._compute_R2(X, lengthscale, variance, X2=X2)
        R = torch.sqrt(torch.clamp(R2, min=1e-14))
        return ((1 + math.sqrt(5) * R + 5 / 3. * R2) * torch.exp(-math.sqrt(5) * R)) * variance.unsqueeze(-2)


class Matern32(Stationary):
    def __init__(self, input_dim, ARD=False, name='matern32', active_dims=None):
        super().__init__(input_dim, ARD, name, active_dims)

    def _compute_K(self, X, lengthscale, variance, X2=None):
        """kernels/matern.py:116-120."""
        R2 = self._compute_R2(X, lengthscale, variance, X2=X2)
        R = torch.sqrt(torch.clamp(R2, min=1e-14))
        return ((1 + math.sqrt(3) * R) * torch.exp(-math.sqrt(3) * R)) * variance.unsqueeze(-2)


class Matern12(Stationary):
    def __init__(self, input_dim, ARD=False, name='matern12', active_dims=None):
        super().__init__(input_dim, ARD, name, active_dims)

    def _compute_K(self, X, lengthscale, variance, X2=None):
        """kernels/matern.py:148-151."""
        R = torch.sqrt(torch.clamp(self._compute_R2(X, lengthscale, variance, X2=X2), min=1e-14))
        return torch.exp(-R) * variance.unsqueeze(-2)


class Linear(Kernel):
    def __init__(self, input_dim, ARD=False, name='linear', active_dims=None):
        super().__init__(input_dim, name, active_dims)
        self.ARD = ARD

    def param_names(self):
        return [self.name + '_variances']

    def _compute_K(self, X, variances, X2=None):
        """kernels/linear.py:59-89."""
        if self.ARD:
            var_sqrt = torch.sqrt(variances).unsqueeze(-2)
            if X2 is None:
                return syrk(X * var_sqrt)
            return gemm2(X * var_sqrt, X2 * var_sqrt, False, True)
        A = syrk(X) if X2 is None else gemm2(X, X2, False, True)
        return A * variances.unsqueeze(-1)

    def _compute_Kdiag(self, X, variances):
        """kernels/linear.py:91-103."""
        return ((X ** 2) * variances.unsqueeze(-2)).sum(-1)


class Bias(Kernel):
    def __init__(self, input_dim, name='bias', active_dims=None):
        super().__init__(input_dim, name, active_dims)

    def param_names(self):
        return [self.name + '_variance']

    def _compute_K(self, X, variance, X2=None):
        """kernels/static.py:56-74 -- variance broadcast to (S,N,N2)."""
        if X2 is None:
            X2 = X
        return variance.reshape(-1, 1, 1).expand(X.shape[0] if variance.shape[0] == 1 else variance.shape[0],
                                                 X.shape[-2], X2.shape[-2]).clone()

    def _compute_Kdiag(self, X, variance):
        """kernels/static.py:76-88."""
        return variance.reshape(-1, 1).expand(max(X.shape[0], variance.shape[0]), X.shape[-2]).clone()


class White(Kernel):
    def __init__(self, input_dim, name='white', active_dims=None):
        super().__init__(input_dim, name, active_dims)

    def param_names(self):
        return [self.name + '_variance']

    def _compute_K(self, X, variance, X2=None):
        """kernels/static.py:125-150 -- eye*variance if X2 is None else zeros."""
        S = max(X.shape[0], variance.shape[0])
        if X2 is None:
            eye = torch.eye(X.shape[-2], dtype=X.dtype).unsqueeze(0).expand(S, -1, -1)
            return eye * variance.reshape(-1, 1, 1)
        return torch.zeros((S, X.shape[-2], X2.shape[-2]), dtype=X.dtype)

    def _compute_Kdiag(self, X, variance):
        """kernels/static.py:152-164."""
        return variance.reshape(-1, 1).expand(max(X.shape[0], variance.shape[0]), X.shape[-2]).clone()


def rename_duplicate_names(names):
    """util/util.py:65-100: [(index, new name)] -- a repeated name gets the first free ``<prefix><count>`` (rbf, rbf -> rbf, rbf0)."""
    import re
    all_names = set(names)
    if len(all_names) == len(names):
        return []
    cur, prog, renames = set(), re.compile(r'^(.*)(\d+)$'), []
    for i, n in enumerate(names):
        if n in cur:
            res = prog.match(n)
            prefix, count = (n, 0) if res is None else (res.groups()[0], int(res.groups()[1]) + 1)
            while prefix + str(count) in all_names:
                count += 1
            renames.append((i, prefix + str(count)))
            all_names.add(prefix + str(count))
        else:
            cur.add(n)
    return renames


class _Combination(Kernel):
    """kernels/kernel.py:317-373 -- sub-kernel parameters are prefixed ``<comb>_<sub>_``; duplicate sub-kernel names are renamed
    (:333-335); input_dim is the largest of the sub-kernels' (:332)."""

    def __init__(self, sub_kernels, name):
        sub_kernels = list(sub_kernels)
        for i, n in rename_duplicate_names([k.name for k in sub_kernels]):
            sub_kernels[i].name = n
        super().__init__(max(k.input_dim for k in sub_kernels), name, None)
        self.sub_kernels = sub_kernels

    def param_names(self):
        return [self.name + '_' + n for k in self.sub_kernels for n in k.param_names()]


class AddKernel(_Combination):
    def __init__(self, sub_kernels, name='add'):
        """kernels/add_kernel.py:36-46: a sum of sums is flattened into one sum."""
        flat = []
        for k in sub_kernels:
            flat.extend(k.sub_kernels if isinstance(k, AddKernel) else [k])
        super().__init__(flat, name)

    def _compute_K(self, X, X2=None, **params):
        """kernels/add_kernel.py:44-68."""
        K = self.sub_kernels[0].K(X, X2, **params)
        for k in self.sub_kernels[1:]:
            K = K + k.K(X, X2, **params)
        return K

    def _compute_Kdiag(self, X, **params):
        """kernels/add_kernel.py:70-88."""
        K = self.sub_kernels[0].Kdiag(X, **params)
        for k in self.sub_kernels[1:]:
            K = K + k.Kdiag(X, **params)
        return K


class MultiplyKernel(_Combination):
    def __init__(self, sub_kernels, name='mul'):
        super().__init__(sub_kernels, name)

    def _compute_K(self, X, X2=None, **params):
        """kernels/multiply_kernel.py:44-67."""
        K = self.sub_kernels[0].K(X, X2, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.K(X, X2, **params)
        return K

    def _compute_Kdiag(self, X, **params):
        """kernels/multiply_kernel.py:69-87."""
        K = self.sub_kernels[0].Kdiag(X, **params)
        for k in self.sub_kernels[1:]:
            K = K * k.Kdiag(X, **params)
        return K


# ----------------------------------------------------------------------------
# Normal: mxfusion/components/distributions/normal.py
# ----------------------------------------------------------------------------
def normal_log_pdf(mean, variance, rv, log_pdf_scaling=1.0):
    """normal.py:52-70 (elementwise)."""
    logvar = LOG2PI / -2 + torch.log(variance) / -2
    return (logvar + (rv - mean) ** 2 / (-2 * variance)) * log_pdf_scaling


def normal_draw(mean, variance, eps):
    """normal.py:72-92 -- eps*sqrt(variance)+mean; eps has shape (S,)+rv_shape."""
    return eps * torch.sqrt(variance) + mean


# ----------------------------------------------------------------------------
# Exact GP regression: mxfusion/modules/gp_modules/gp_regression.py
# ----------------------------------------------------------------------------
def gp_log_pdf(kern, X, Y, noise_var, kern_params, jitter=0., mean=None, return_posterior=False):
    """gp_regression.py:42-76.  X:(S,N,Q) Y:(S,N,P) noise_var:(S,1) -> (S,)."""
    D = Y.shape[-1]
    N = X.shape[-2]
    X, Y, noise_var, kern_params = arrays_as_samples([X, Y, noise_var, kern_params])
    eye = torch.eye(N, dtype=X.dtype).unsqueeze(0)
    K = kern.K(X, **kern_params) + eye * noise_var.unsqueeze(-2)
    if jitter > 0.:
        K = K + eye * jitter
    L = potrf(K)
    if mean is not None:
        Y = Y - mean
    LinvY = trsm(L, Y)
    logdet_l = sumlogdiag(torch.abs(L))
    tmp = (LinvY ** 2 + LOG2PI).reshape(Y.shape[0], -1).sum(-1)
    logL = -logdet_l * D - tmp / 2
    if return_posterior:
        # gp_regression.py:72-75 persists sample 0 only
        return logL, (X[0].detach(), L[0].detach(), LinvY[0].detach())
    return logL


def gp_predict(kern, Xt, noise_var, X_cond, L, LinvY, kern_params, mean=None,
               noise_free=True, diagonal_variance=True):
    """gp_regression.py:146-196.  All arrays carry the S axis (posterior ones S=1)."""
    N = Xt.shape[-2]
    Xt, noise_var, X_cond, L, LinvY, kern_params = arrays_as_samples(
        [Xt, noise_var, X_cond, L, LinvY, kern_params])
    Kxt = kern.K(X_cond, Xt, **kern_params)
    LinvKxt = trsm(L, Kxt)
    mu = gemm2(LinvKxt, LinvY, True, False)
    if mean is not None:
        mu = mu + mean
    if diagonal_variance:
        Ktt = kern.Kdiag(Xt, **kern_params)
        var = Ktt - (LinvKxt ** 2).sum(-2)
        if not noise_free:
            var = var + noise_var
    else:
        Ktt = kern.K(Xt, **kern_params)
        var = Ktt - syrk(LinvKxt, True)
        if not noise_free:
            var = var + torch.eye(N, dtype=Xt.dtype).unsqueeze(0) * noise_var.unsqueeze(-2)
    return mu, var


def gp_sample_prior(kern, X, noise_var, kern_params, eps, mean=None):
    """gp_regression.py:92-135 -- Y = L eps (+mean); eps:(S,N,P)."""
    N = X.shape[-2]
    X, noise_var, kern_params = arrays_as_samples([X, noise_var, kern_params])
    K = kern.K(X, **kern_params) + torch.eye(N, dtype=X.dtype).unsqueeze(0) * noise_var.unsqueeze(-2)
    L = potrf(K)
    y = trmm(L.expand(eps.shape[0], -1, -1), eps)
    return y + mean if mean is not None else y


def gp_predict_sample(kern, Xt, noise_var, X_cond, L, LinvY, kern_params, eps, mean=None,
                      noise_free=True, diagonal_variance=True, jitter=0.):
    """gp_regression.py:213-275."""
    mu, var = gp_predict(kern, Xt, noise_var, X_cond, L, LinvY, kern_params, mean=mean,
                         noise_free=noise_free, diagonal_variance=diagonal_variance)
    if diagonal_variance:
        return mu + eps * torch.sqrt(var.unsqueeze(-1))
    N = Xt.shape[-2]
    if jitter > 0.:
        var = var + torch.eye(N, dtype=Xt.dtype).unsqueeze(0) * jitter
    Lc = potrf(var)
    return mu + trmm(Lc.expand(eps.shape[0], -1, -1), eps)


# ----------------------------------------------------------------------------
# SVGP: mxfusion/modules/gp_modules/svgp_regression.py
# ----------------------------------------------------------------------------
def svgp_log_pdf(kern, X, Y, Z, noise_var, mu, S_W, S_diag, kern_params, jitter=0.,
                 log_pdf_scaling=1., mean=None):
    """svgp_regression.py:43-109.
    X:(S,B,Q) Y:(S,B,P) Z:(S,M,Q) noise_var:(S,1) or (S,B,P) mu:(S,M,P) S_W:(S,M,M) S_diag:(S,M) -> (S,)."""
    D = Y.shape[-1]
    M = Z.shape[-2]
    X, Y, Z, noise_var, mu, S_W, S_diag, kern_params = arrays_as_samples(
        [X, Y, Z, noise_var, mu, S_W, S_diag, kern_params])
    if noise_var.dim() == 2:
        noise_var = noise_var.unsqueeze(-2)
    if noise_var.shape[-1] == 1:
        beta_sum = D * (1 / noise_var).sum(-1)
    else:
        beta_sum = (1 / noise_var).sum(-1)
    Kuu = kern.K(Z, **kern_params)
    if jitter > 0.:
        Kuu = Kuu + torch.eye(M, dtype=Z.dtype).unsqueeze(0) * jitter
    Kuf = kern.K(Z, X, **kern_params)
    Kff_diag = kern.Kdiag(X, **kern_params)
    S = syrk(S_W) + make_diagonal(S_diag)
    if mean is not None:
        Y = Y - mean
    psi1Y = gemm2(Kuf, Y / noise_var, False, False)
    L = potrf(Kuu)
    Ls = potrf(S)
    LinvLs = trsm(L, Ls)
    Linvmu = trsm(L, mu)
    LinvKuf = trsm(L, Kuf)
    KfuKuuInvmu = gemm2(LinvKuf, Linvmu, True, False)
    KfuKuuInvLs = gemm2(LinvKuf, LinvLs, True, False)
    LinvKufY = trsm(L, psi1Y)
    KL_u = (M / 2. + sumlogdiag(Ls)) * D - sumlogdiag(L) * D \
        - (LinvLs ** 2).sum(-1).sum(-1) / 2. * D \
        - (Linvmu ** 2).sum(-1).sum(-1) / 2.
    logL = -((Y ** 2) / noise_var + LOG2PI + torch.log(noise_var)).sum(-1).sum(-1) / 2.
    logL = logL - (Kff_diag * beta_sum).sum(-1) / 2.
    logL = logL - ((KfuKuuInvmu ** 2) / noise_var).sum(-1).sum(-1) / 2.
    logL = logL - ((KfuKuuInvLs ** 2) * beta_sum.unsqueeze(-1)).sum(-1).sum(-1) / 2.
    logL = logL + ((LinvKuf ** 2) * beta_sum.unsqueeze(-2)).sum(-1).sum(-1) / 2.
    logL = logL + (Linvmu * LinvKufY).sum(-1).sum(-1)
    return log_pdf_scaling * logL + KL_u


def svgp_predict(kern, Xt, Z, noise_var, mu, S_W, S_diag, kern_params, jitter=0., mean=None,
                 noise_free=True, diagonal_variance=True):
    """svgp_regression.py:121-189 (note: no arrays_as_samples; var gets a trailing axis :170)."""
    N = Xt.shape[-2]
    M = Z.shape[-2]
    S = syrk(S_W) + make_diagonal(S_diag)
    Kuu = kern.K(Z, **kern_params)
    if jitter > 0.:
        Kuu = Kuu + torch.eye(M, dtype=Z.dtype) * jitter
    L = potrf(Kuu)
    Ls = potrf(S)
    LinvLs = trsm(L, Ls)
    Linvmu = trsm(L, mu)
    LinvSLinvT = syrk(LinvLs)
    wv = trsm(L, Linvmu, transpose=True)
    Kxt = kern.K(Z, Xt, **kern_params)
    mu_t = gemm2(Kxt, wv, True, False)
    if mean is not None:
        mu_t = mu_t + mean
    LinvKxt = trsm(L, Kxt)
    if diagonal_variance:
        Ktt = kern.Kdiag(Xt, **kern_params)
        tmp = gemm2(LinvSLinvT, LinvKxt)
        var = Ktt - (LinvKxt ** 2).sum(-2) + (tmp * LinvKxt).sum(-2)
        var = var.unsqueeze(-1)
        if not noise_free:
            var = var + noise_var
    else:
        Ktt = kern.K(Xt, **kern_params)
        tmp = gemm2(LinvSLinvT, LinvKxt)
        var = Ktt - syrk(LinvKxt, True) + gemm2(LinvKxt, tmp, True, False)
        var = var.unsqueeze(-1)
        if not noise_free:
            var = var + torch.eye(N, dtype=Xt.dtype).reshape(1, N, N, 1) * noise_var.unsqueeze(-2)
    return mu_t, var


def svgp_predict_sample(kern, Xt, Z, noise_var, mu, S_W, S_diag, kern_params, eps, jitter=0., mean=None,
                        noise_free=True, diagonal_variance=True):
    """svgp_regression.py:203-280: the moments of :226-260 (jitter on Kuu, :236-238), then mu + eps sqrt(var) (:254-255) or
    mu + chol(cov) eps (:262-272; no jitter on the predictive covariance); eps:(S,Nt,P)."""
    mu_t, var = svgp_predict(kern, Xt, Z, noise_var, mu, S_W, S_diag, kern_params, jitter=jitter, mean=mean,
                             noise_free=noise_free, diagonal_variance=diagonal_variance)
    if diagonal_variance:
        return mu_t + eps * torch.sqrt(var)
    Lc = potrf(var[..., 0])
    return mu_t + trmm(Lc.expand(eps.shape[0], -1, -1), eps)


def svgp_log_pdf_suffstats(kern, X, Y, Z, noise_var, mu, S_W, S_diag, kern_params, jitter=0.,
                           log_pdf_scaling=1., mean=None):
    """SURVEY Appendix A.5 (NOT in the reference): the same bound written in the streaming
    sufficient statistics (Psi2, psi1, kappa, upsilon).  Homoscedastic only.  Used to check the
    algebra the HIP path relies on against ``svgp_log_pdf``."""
    P = Y.shape[-1]
    M = Z.shape[-2]
    X, Y, Z, noise_var, mu, S_W, S_diag, kern_params = arrays_as_samples(
        [X, Y, Z, noise_var, mu, S_W, S_diag, kern_params])
    B = X.shape[-2]
    s2 = noise_var.reshape(-1)
    beta = 1. / s2
    Kuu = kern.K(Z, **kern_params) + torch.eye(M, dtype=Z.dtype).unsqueeze(0) * jitter
    Kuf = kern.K(Z, X, **kern_params)
    if mean is not None:
        Y = Y - mean
    Psi2 = Kuf @ Kuf.transpose(-1, -2)
    psi1 = Kuf @ Y
    kappa = kern.Kdiag(X, **kern_params).sum(-1)
    ups = (Y ** 2).sum(-1).sum(-1)
    Su = syrk(S_W) + make_diagonal(S_diag)
    Ki = torch.linalg.inv(Kuu)
    w = Ki @ mu
    H = 0.5 * P * beta.reshape(-1, 1, 1) * (Ki - Ki @ Su @ Ki)
    G = H - 0.5 * beta.reshape(-1, 1, 1) * (w @ w.transpose(-1, -2))
    data = -0.5 * (beta * ups + B * P * (LOG2PI + torch.log(s2))) - 0.5 * P * beta * kappa \
        + (Psi2 * G).sum(-1).sum(-1) + beta * (psi1 * w).sum(-1).sum(-1)
    negKL = 0.5 * P * (M + torch.logdet(Su) - torch.logdet(Kuu) - (Ki * Su.transpose(-1, -2)).sum(-1).sum(-1)) \
        - 0.5 * (mu * w).sum(-1).sum(-1)
    return log_pdf_scaling * data + negKL


# ----------------------------------------------------------------------------
# Sparse (Titsias) GP: mxfusion/modules/gp_modules/sparsegp_regression.py
# ----------------------------------------------------------------------------
def sgp_log_pdf(kern, X, Y, Z, noise_var, kern_params, jitter=0., mean=None, return_posterior=False):
    """sparsegp_regression.py:42-108."""
    D = Y.shape[-1]
    M = Z.shape[-2]
    X, Y, Z, noise_var, kern_params = arrays_as_samples([X, Y, Z, noise_var, kern_params])
    noise_var_m = noise_var.unsqueeze(-2)
    Kuu = kern.K(Z, **kern_params)
    if jitter > 0.:
        Kuu = Kuu + torch.eye(M, dtype=Z.dtype).unsqueeze(0) * jitter
    Kuf = kern.K(Z, X, **kern_params)
    Kff_diag = kern.Kdiag(X, **kern_params)
    L = potrf(Kuu)
    LinvKuf = trsm(L, Kuf)
    A = torch.eye(M, dtype=Z.dtype).unsqueeze(0) + syrk(LinvKuf) / noise_var_m
    LA = potrf(A)
    if mean is not None:
        Y = Y - mean
    LAInvLinvKufY = trsm(LA, gemm2(LinvKuf, Y))
    logL = -D * sumlogdiag(LA)
    logL = logL - ((Y ** 2) / noise_var_m + LOG2PI + torch.log(noise_var_m)).sum(-1).sum(-1) / 2
    logL = logL + ((LAInvLinvKufY ** 2) / (2 * noise_var_m ** 2)).sum(-1).sum(-1)
    logL = logL - D * (Kff_diag / (2 * noise_var)).sum(-1)
    logL = logL + D * ((LinvKuf ** 2) / (2. * noise_var_m)).sum(-1).sum(-1)
    if return_posterior:
        wv = trsm(L, trsm(LA, LAInvLinvKufY, transpose=True), transpose=True) / noise_var_m
        return logL, (wv[0].detach(), L[0].detach(), LA[0].detach())
    return logL


def sgp_predict(kern, Xt, Z, noise_var, L, LA, wv, kern_params, mean=None,
                noise_free=True, diagonal_variance=True):
    """sparsegp_regression.py:119-174."""
    N = Xt.shape[-2]
    Xt, Z, noise_var, L, LA, wv, kern_params = arrays_as_samples([Xt, Z, noise_var, L, LA, wv, kern_params])
    Kxt = kern.K(Z, Xt, **kern_params)
    mu = gemm2(Kxt, wv, True, False)
    if mean is not None:
        mu = mu + mean
    LinvKxt = trsm(L, Kxt)
    LAinvLinvKxt = trsm(LA, LinvKxt)
    if diagonal_variance:
        Ktt = kern.Kdiag(Xt, **kern_params)
        var = Ktt - (LinvKxt ** 2).sum(-2) + (LAinvLinvKxt ** 2).sum(-2)
        if not noise_free:
            var = var + noise_var
    else:
        Ktt = kern.K(Xt, **kern_params)
        var = Ktt - syrk(LinvKxt, True) + syrk(LAinvLinvKxt, True)
        if not noise_free:
            var = var + torch.eye(N, dtype=Xt.dtype).unsqueeze(0) * noise_var.unsqueeze(-2)
    return mu, var


def sgp_predict_sample(kern, Xt, Z, noise_var, L, LA, wv, kern_params, eps, mean=None, noise_free=True,
                       diagonal_variance=True, jitter=0.):
    """sparsegp_regression.py:188-255: mu + eps sqrt(var) (:228-234) or mu + chol(cov + jitter I) eps (:236-249); eps:(S,Nt,P)."""
    mu, var = sgp_predict(kern, Xt, Z, noise_var, L, LA, wv, kern_params, mean=mean, noise_free=noise_free,
                          diagonal_variance=diagonal_variance)
    if diagonal_variance:
        return mu + eps * torch.sqrt(var.unsqueeze(-1))
    N = Xt.shape[-2]
    if jitter > 0.:
        var = var + torch.eye(N, dtype=Xt.dtype).unsqueeze(0) * jitter
    Lc = potrf(var)
    return mu + trmm(Lc.expand(eps.shape[0], -1, -1), eps)


def sparse_gp_forward_sample(kern, X, Z, noise_var, kern_params, eps_u, eps_f, eps_y, mean=None):
    """The default draw_samples of SVGPRegression / SparseGPRegression: ForwardSamplingAlgorithm (forward_sampling.py:24-37 ->
    factor_graph.py:240-297) over the module graph of svgp_regression.py:349-374 / sparsegp_regression.py:323-347:
    U ~ GP(Z) (gp.py:124-153), F ~ GP(X | Z, U) (cond_gp.py:185-223), Y ~ N(F, noise_var) (normal.py:72-92).
    eps_u:(S,M,P), eps_f, eps_y:(S,N,P), consumed in that order."""
    U = gp_dist_draw(kern, Z, kern_params, eps_u)
    Fv = cond_gp_dist_draw(kern, X, Z, U, kern_params, eps_f, mean=mean)
    return normal_draw(Fv, noise_var.unsqueeze(-1) if noise_var.dim() == 2 else noise_var, eps_y), U, Fv


# ----------------------------------------------------------------------------
# GaussianProcess / ConditionalGaussianProcess distributions:
# mxfusion/components/distributions/gp/{gp,cond_gp}.py
# ----------------------------------------------------------------------------
def gp_dist_log_pdf(kern, X, rv, kern_params, mean=None, log_pdf_scaling=1.):
    """gp.py:95-122 -- no noise, no jitter."""
    D = rv.shape[-1]
    K = kern.K(X, **kern_params)
    L = potrf(K)
    if mean is not None:
        rv = rv - mean
    LinvY = trsm(L, rv)
    logdet_l = sumlogdiag(torch.abs(L))
    return (-logdet_l * D - (LinvY ** 2 + LOG2PI).sum(-1).sum(-1) / 2) * log_pdf_scaling


def gp_dist_draw(kern, X, kern_params, eps, mean=None):
    """gp.py:124-153 -- L eps (+mean)."""
    K = kern.K(X, **kern_params)
    L = potrf(K)
    y = trmm(L.expand(eps.shape[0], -1, -1), eps)
    return y + mean if mean is not None else y


def _cond_gp_moments(kern, X, X_cond, Y_cond, kern_params, mean_cond=None):
    """cond_gp.py:164-177 (shared by log_pdf_impl and draw_samples_impl)."""
    K = kern.K(X, **kern_params)
    Kc = kern.K(X_cond, X, **kern_params)
    Kcc = kern.K(X_cond, **kern_params)
    Lcc = potrf(Kcc)
    LccInvKc = trsm(Lcc, Kc)
    cov = K - syrk(LccInvKc, transpose=True)
    L = potrf(cov)
    if mean_cond is not None:
        Y_cond = Y_cond - mean_cond
    LccInvY = trsm(Lcc, Y_cond)
    rv_mean = gemm2(LccInvKc, LccInvY, True, False)
    return L, rv_mean


def cond_gp_dist_log_pdf(kern, X, X_cond, Y_cond, rv, kern_params, mean=None, mean_cond=None, log_pdf_scaling=1.):
    """cond_gp.py:124-183 -- including the reference's sum over the output axis BEFORE squaring (:179)."""
    D = rv.shape[-1]
    L, rv_mean = _cond_gp_moments(kern, X, X_cond, Y_cond, kern_params, mean_cond)
    if mean is not None:
        rv = rv - mean
    LinvY = trsm(L, rv - rv_mean).sum(-1)
    logdet_l = sumlogdiag(torch.abs(L))
    return (-logdet_l * D - (LinvY ** 2 + LOG2PI).sum(-1) / 2) * log_pdf_scaling


def cond_gp_dist_draw(kern, X, X_cond, Y_cond, kern_params, eps, mean=None, mean_cond=None):
    """cond_gp.py:185-223."""
    L, rv_mean = _cond_gp_moments(kern, X, X_cond, Y_cond, kern_params, mean_cond)
    rv = trmm(L.expand(eps.shape[0], -1, -1), eps) + rv_mean
    return rv + mean if mean is not None else rv


# ----------------------------------------------------------------------------
# Objective assembly: inference/{map,variational}.py + models/factor_graph.py:192-238
# ----------------------------------------------------------------------------
def factor_sum(logpdf):
    """factor_graph.py:223,233 -- F.sum(expectation(F, log_pdf))."""
    return expectation(logpdf).sum()


def map_gp_loss(kern, X, Y, raw, jitter=0.):
    """map.py:79-84 + inference_alg.py:75-83 for the GP-regression model of
    examples/notebooks/gp_regression.ipynb cell 10: all three positive parameters are
    optimised in softplus-raw space.  X:(N,Q) Y:(N,P); raw: dict of raw parameters."""
    params = {kern.name + '_lengthscale': add_sample_dimension(softplus(raw['lengthscale'])),
              kern.name + '_variance': add_sample_dimension(softplus(raw['variance']))}
    noise = add_sample_dimension(softplus(raw['noise_var']))
    logL = gp_log_pdf(kern, add_sample_dimension(X), add_sample_dimension(Y), noise, params, jitter=jitter)
    return -factor_sum(logL)


def svi_latent_svgp_loss(kern, Y, Z, raw, eps, jitter=0., log_pdf_scaling=1.):
    """variational.py:91-108 on the model of testing/modules/svgpregression_test.py:365-385:
    X ~ N(0,1) (prior), q(X)=N(qm, softplus(qv_raw)) mean-field, Y ~ SVGP(X).
    eps:(S,N,Q) is the injected reparameterisation noise (random_gen.py seam).
    raw: dict with qX_mean, qX_var(raw), noise_var(raw), lengthscale(raw), variance(raw),
    qU_mean, qU_cov_W, qU_cov_diag(raw); Z:(M,Q) (a plain parameter)."""
    qm = add_sample_dimension(raw['qX_mean'])
    qv = add_sample_dimension(softplus(raw['qX_var']))
    Xs = normal_draw(qm, qv, eps)                                  # posterior.draw_samples
    params = {kern.name + '_lengthscale': add_sample_dimension(softplus(raw['lengthscale'])),
              kern.name + '_variance': add_sample_dimension(softplus(raw['variance']))}
    noise = add_sample_dimension(softplus(raw['noise_var']))
    one = torch.ones(1, 1, 1, dtype=Xs.dtype)
    lp_prior = factor_sum(normal_log_pdf(0. * one, one, Xs))       # m.X ~ N(0,1)
    lp_svgp = factor_sum(svgp_log_pdf(
        kern, Xs, add_sample_dimension(Y), add_sample_dimension(Z), noise,
        add_sample_dimension(raw['qU_mean']), add_sample_dimension(raw['qU_cov_W']),
        add_sample_dimension(softplus(raw['qU_cov_diag'])), params, jitter=jitter,
        log_pdf_scaling=log_pdf_scaling))
    lq = factor_sum(normal_log_pdf(qm, qv, Xs))                    # posterior.log_pdf
    return -(lp_prior + lp_svgp - lq)


def map_svgp_loss(kern, X, Y, raw, jitter=0., log_pdf_scaling=1.):
    """map.py:79-84 + inference_alg.py:75-83 for the SVGP model of examples/notebooks/svgp_regression.ipynb cell 9 (X, Y observed; the
    module's hidden parameters qU_* and the inducing inputs are plain parameters, positive ones in softplus-raw space).
    raw: Z, noise_var(raw), lengthscale(raw), variance(raw), qU_mean, qU_cov_W, qU_cov_diag(raw)."""
    params = {kern.name + '_lengthscale': add_sample_dimension(softplus(raw['lengthscale'])),
              kern.name + '_variance': add_sample_dimension(softplus(raw['variance']))}
    logL = svgp_log_pdf(kern, add_sample_dimension(X), add_sample_dimension(Y), add_sample_dimension(raw['Z']),
                        add_sample_dimension(softplus(raw['noise_var'])), add_sample_dimension(raw['qU_mean']),
                        add_sample_dimension(raw['qU_cov_W']), add_sample_dimension(softplus(raw['qU_cov_diag'])), params, jitter=jitter,
                        log_pdf_scaling=log_pdf_scaling)
    return -factor_sum(logL)


def svi_uncertain_input_svgp_loss(kern, Xobs, Y, raw, eps, prior_var=1e-2, jitter=0., log_pdf_scaling=1.):
    """variational.py:91-108 on the minibatch-capable latent-input model (BASELINE.json configs[3]):
        X ~ N(Xobs, prior_var) row-wise,  Y ~ SVGP(X),  q(X) = N(Xobs, softplus(qx_var)) with ONE shared variance parameter,
    so that every factor is a sum over rows and a minibatch (rows of Xobs and Y) with rv_scaling = N/B on both X and Y is unbiased
    (minibatch_loop.py:36-40 -> inference_alg.py:183-187 -> normal.py:70, svgp_regression.py:108).  eps:(S,B,Q) injected noise.
    raw: qx_var(raw), noise_var(raw), lengthscale(raw), variance(raw), qU_mean, qU_cov_W, qU_cov_diag(raw), Z."""
    Xo = add_sample_dimension(Xobs)
    qv = softplus(raw['qx_var']).reshape(1, 1, 1)
    Xs = normal_draw(Xo, qv, eps)
    params = {kern.name + '_lengthscale': add_sample_dimension(softplus(raw['lengthscale'])),
              kern.name + '_variance': add_sample_dimension(softplus(raw['variance']))}
    pv = torch.full((1, 1, 1), float(prior_var), dtype=Xs.dtype)
    lp_prior = factor_sum(normal_log_pdf(Xo, pv, Xs, log_pdf_scaling=log_pdf_scaling))
    lp_svgp = factor_sum(svgp_log_pdf(
        kern, Xs, add_sample_dimension(Y), add_sample_dimension(raw['Z']), add_sample_dimension(softplus(raw['noise_var'])),
        add_sample_dimension(raw['qU_mean']), add_sample_dimension(raw['qU_cov_W']), add_sample_dimension(softplus(raw['qU_cov_diag'])),
        params, jitter=jitter, log_pdf_scaling=log_pdf_scaling))
    lq = factor_sum(normal_log_pdf(Xo, qv, Xs, log_pdf_scaling=log_pdf_scaling))
    return -(lp_prior + lp_svgp - lq)


# ----------------------------------------------------------------------------
# Optimiser: MXNet `adam` as driven by gluon.Trainer.step (batch_loop.py:46-60)
# ----------------------------------------------------------------------------
class MXNetAdam(object):
    """MXNet Adam (API knowledge): grad *= rescale_grad (=1/batch_size);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; lr_t = lr sqrt(1-b2^t)/(1-b1^t);
    w -= lr_t m / (sqrt(v)+eps).  b1=0.9 b2=0.999 eps=1e-8, wd=0."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
        self.t = 0
        self.m = {}
        self.v = {}

    def step(self, params, grads, batch_size=1):
        self.t += 1
        lr_t = self.lr * math.sqrt(1. - self.b2 ** self.t) / (1. - self.b1 ** self.t)
        for k in params:
            g = grads[k] * (1.0 / batch_size)
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k])
                self.v[k] = torch.zeros_like(params[k])
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            params[k] = params[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        return params


class MXNetRule(object):
    """MXNet 1.x 'rmsprop' (non-centred) / 'adagrad' / 'adadelta' / 'nag' update rules on one tensor (API knowledge of MXNet's python
    optimisers; no reference-held vector: parity unpinned).  Checker of mxf_opt_step."""

    def __init__(self, kind, lr, p1=None, epsilon=None, wd=0.0):
        d = {'rmsprop': (0.9, 1e-8), 'adagrad': (0.0, 1e-7), 'adadelta': (0.9, 1e-5), 'nag': (0.0, 0.0)}[kind]
        self.kind, self.lr, self.wd = kind, lr, wd
        self.p1 = d[0] if p1 is None else p1
        self.eps = d[1] if epsilon is None else epsilon
        self.s1 = self.s2 = None

    def step(self, w, grad, batch_size=1):
        g = grad * (1.0 / batch_size)
        if self.s1 is None:
            self.s1, self.s2 = torch.zeros_like(w), torch.zeros_like(w)
        if self.kind == 'rmsprop':
            g = g + self.wd * w
            self.s1 = (1 - self.p1) * g * g + self.p1 * self.s1
            return w - self.lr * g / torch.sqrt(self.s1 + self.eps)
        if self.kind == 'adagrad':
            self.s1 = self.s1 + g * g
            return w - self.lr * (g / torch.sqrt(self.s1 + self.eps) + self.wd * w)
        if self.kind == 'adadelta':
            self.s1 = self.p1 * self.s1 + (1 - self.p1) * g * g
            d = torch.sqrt(self.s2 + self.eps) / torch.sqrt(self.s1 + self.eps) * g
            self.s2 = self.p1 * self.s2 + (1 - self.p1) * d * d
            return w - (d + self.wd * w)
        g = g + self.wd * w
        self.s1 = self.p1 * self.s1 + g
        return w - self.lr * (g + self.p1 * self.s1)


def run_map_gp_notebook(X, Y, max_iter=100, lr=0.05, record=(10, 20, 30, 40, 50, 60, 70, 80, 90, 100)):
    """Replays examples/notebooks/gp_regression.ipynb cells 10-14 through the oracle:
    RBF(1), variance=1, lengthscale=1, noise=0.01, Adam lr 0.05, 100 iterations.
    Returns ({iteration: loss printed at that iteration}, final constrained params)."""
    kern = RBF(1, ARD=False)
    X = T(X)
    Y = T(Y)
    raw = {'lengthscale': inv_softplus(T([1.0])), 'variance': inv_softplus(T([1.0])),
           'noise_var': inv_softplus(T([0.01]))}
    opt = MXNetAdam(lr)
    losses = {}
    for i in range(max_iter):
        leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        loss = map_gp_loss(kern, X, Y, leaves)
        loss.backward()
        if (i + 1) in record:
            losses[i + 1] = float(loss.detach())
        raw = opt.step({k: v.detach() for k, v in leaves.items()},
                       {k: v.grad for k, v in leaves.items()}, batch_size=1)
    final = {k: float(softplus(v)[0]) for k, v in raw.items()}
    return losses, final


# ----------------------------------------------------------------------------
# PILCO rollout: mxfusion/inference/pilco_alg.py:55-90
# ----------------------------------------------------------------------------
def pilco_rollout(predict, policy, cost_function, s_0, n_time_steps):
    """pilco_alg.py:72-90 with `predict(x_t) -> (mean, variance)` standing for `model.Y.factor.predict(...)[0]` (:80).
    s_0: (S, state_dim); states and actions travel as (S, 1, dim) (the action is given that shape from the first step on).
    Differentiable through torch autograd w.r.t. whatever `policy` closes over."""
    S = s_0.shape[0]
    a_t = policy(s_0).reshape(S, 1, -1)
    x_t = torch.cat([s_0.reshape(S, 1, -1), a_t], dim=2)
    cost = 0
    for t in range(n_time_steps):
        res = predict(x_t)
        s_next = res[0]
        cost = cost + cost_function(s_next, a_t)
        a_t = policy(s_next).reshape(S, 1, -1)
        x_t = torch.cat([s_next, a_t], dim=2)
    return torch.sum(cost)


def run_svgp_notebook(X, Y, raw0, permutations, batch_size=10, phases=((50, 0.1), (50, 0.01)), jitter=1e-6):
    """examples/notebooks/svgp_regression.ipynb cell 11: MAP on the SVGP model through MinibatchInferenceLoop(batch_size=10,
    rv_scaling={Y: N/B}) (minibatch_loop.py:65-93: a fresh shuffled DataLoader and a fresh Adam per run() call, Trainer.step(batch_size=B),
    last_batch='rollover'), 50 epochs at lr 0.1 then 50 at lr 0.01.  `permutations` yields one index permutation per epoch (the
    DataLoader shuffle is MXNet-RNG dependent in the reference; injected here).  raw0: the initial raw parameters (see map_svgp_loss).
    Returns the final raw parameters and the per-epoch mean losses."""
    kern = RBF(X.shape[-1], ARD=False)          # RBF(input_dim=1, variance=1, lengthscale=1): ARD defaults to False (rbf.py:38)
    N = X.shape[0]
    raw = {k: v.clone() for k, v in raw0.items()}
    perms = iter(permutations)
    epoch_losses = []
    for n_epochs, lr in phases:
        opt = MXNetAdam(lr)
        carry = torch.empty(0, dtype=torch.long)
        for _ in range(n_epochs):
            idx = torch.cat([carry, torch.as_tensor(next(perms), dtype=torch.long)])
            n_full = idx.numel() // batch_size
            tot = 0.
            for i in range(n_full):
                sel = idx[i * batch_size:(i + 1) * batch_size]
                lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
                loss = map_svgp_loss(kern, X[sel], Y[sel], lv, jitter=jitter, log_pdf_scaling=N / batch_size)
                loss.backward()
                raw = opt.step({k: v.detach() for k, v in lv.items()}, {k: v.grad for k, v in lv.items()}, batch_size=batch_size)
                tot += float(loss)
            carry = idx[n_full * batch_size:]
            epoch_losses.append(tot / max(n_full, 1))
    return raw, epoch_losses
