"""SVGPRegression module and its algorithms (mxfusion/modules/gp_modules/svgp_regression.py:32-457).

  SVGPRegressionLogPdf.compute                 -> mxf_svgp_logpdf: streaming sufficient-statistics bound + reverse mode,
                                                  (M x M) factorisations once in float64 (composite.hip)
  SVGPRegressionMeanVariancePrediction.compute -> mxf_gram + mxf_potrf + mxf_trsm + mxf_gemm + mxf_coldot
"""
from types import SimpleNamespace

import numpy as np
import torch

from ... import ops
from ...components.variables.variable import Variable
from ...components.variables.var_trans import PositiveTransformation
from ...inference.inference_alg import SamplingAlgorithm
from ...inference.variational import VariationalInference
from ..module import Module, ModuleGraph
from ...inference.forward_sampling import ForwardSamplingAlgorithm
from ._fused import SVGPLogPdfFn, SVGPMatLogPdfFn, SVGPSampledLogPdfFn
from ._sampling_graph import build_sparse_gp_sampling_model
from ...components.distributions.gp import _linalg as lin
from .gp_regression import _grad_mode


def _S(t):
    return t.shape[0]


class SVGPRegressionLogPdf(VariationalInference):
    """svgp_regression.py:32-109."""

    def __init__(self, model, posterior, observed, jitter=0.):
        super(SVGPRegressionLogPdf, self).__init__(model=model, posterior=posterior, observed=observed)
        self.log_pdf_scaling = 1
        self.jitter = jitter

    def _f32_guard(self):
        """This algorithm object's float32 validity guard (one per module: its own condition slot and level, _fused.Float32Guard)."""
        g = getattr(self, '_guard', None)
        if g is None:
            from ._fused import Float32Guard
            g = self._guard = Float32Guard(getattr(getattr(self.model, 'F', None), 'name', None) or 'svgp')
        return g

    PMAX = 8     # output columns per fused call (register tile of the composites); wider Y is processed in column blocks

    kl_weight = 1.0   # weight of the row-independent terms (-KL(q(u) || p(u))): 1 / world size when a data-parallel loop shards the rows

    def compute(self, F, variables):
        """c * (data term over the rows handed in) + kl_weight * (-KL): with kl_weight = w != 1 evaluated as w * (c / w * data - KL), i.e. the
        same fused call with the scaling c / w (the bound is linear in c: svgp_regression.py:109)."""
        w = float(self.kl_weight)
        if w == 1.0:
            return self._compute(F, variables)
        c = self.log_pdf_scaling
        self.log_pdf_scaling = c / w
        try:
            return w * self._compute(F, variables)
        finally:
            self.log_pdf_scaling = c

    def _compute(self, F, variables):
        has_mean = self.model.F.factor.has_mean
        X = variables[self.model.X]
        Y = variables[self.model.Y]
        Z = variables[self.model.inducing_inputs]
        noise_var = variables[self.model.noise_var]
        mu = variables[self.posterior.qU_mean]
        S_W = variables[self.posterior.qU_cov_W]
        S_diag = variables[self.posterior.qU_cov_diag]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        if has_mean:
            Y = Y - variables[self.model.mean]
        P = Y.shape[-1]
        if P > self.PMAX:
            # The bound is a SUM over output columns for a shared q(u) covariance (svgp_regression.py:93-108: every term carries the
            # factor D or a sum over the D columns; the -KL part D times the covariance terms plus |L^-1 mu_d|^2 per column), so
            # log L(Y[:, :D]) = sum over column blocks of log L(block): one fused call per block of <= 8 columns, autograd adds the
            # gradients of the shared parameters.
            total, infos = None, []
            per_col = noise_var.shape[-1] == P and P > 1
            for p0 in range(0, P, self.PMAX):
                sl = slice(p0, min(p0 + self.PMAX, P))
                part = self._compute_columns(F, X, Y[..., sl], Z, noise_var[..., sl] if per_col else noise_var, mu[..., sl], S_W, S_diag,
                                             kern, kern_params)
                infos.append(self._last_info)
                total = part if total is None else total + part
            self._last_info = ops.merge_info(*infos)
            return total
        return self._compute_columns(F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params)

    # float32 calls on SMALL problems are evaluated in float64 inside (inputs widened, result narrowed; the exact GP and every prediction do the
    # same): below ~4 M covariances per call the float64 streaming kernels cost < 0.3 ms -- the call is launch-bound either way -- and the
    # float32 forms' condition limits are calibrated on BASELINE-sized problems: r04's randomised sweep found 3-5e-5 on the bound for
    # M = 7 ... 100 / B ~ 1 000 problems right below them (relative to a small |ELBO|).  Large problems keep the three-level guard.
    SMALL_F64_ELEMS = 1 << 22

    def _small_in_float64(self, X, Z):
        from ._fused import Float32Guard
        if not (X.is_cuda and X.dtype == torch.float32 and Float32Guard.enabled and Float32Guard.force is None):
            return False
        return X.shape[0] * X.shape[-2] * max(Z.shape[-2], 128) <= self.SMALL_F64_ELEMS

    def _compute_columns(self, F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params):
        if self._small_in_float64(X, Z):
            g = self._f32_guard()
            g.small_f64, g._widened_by_owner = True, True
            try:
                return self._compute_columns(F, *[t.double() for t in (X, Y, Z, noise_var, mu, S_W, S_diag)], kern,
                                             {k: v.double() for k, v in kern_params.items()}).float()
            finally:
                g._widened_by_owner = False
        if X.is_cuda and X.dtype == torch.float32 and getattr(self, '_guard', None) is not None:
            self._guard.small_f64 = False
        spec = kern.fused_spec()
        if spec is None:
            return self._compute_materialised(F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params)
        kind, ard = spec
        ls = kern_params[kern.name + '_lengthscale']
        var = kern_params[kern.name + '_variance']
        if kind == 'rbf' and X.is_cuda and X.dtype == torch.float32 and torch.is_grad_enabled():
            g_ = self._f32_guard()                      # the input-range check on the REAL rows and inducing inputs (the padding below is far away by design)
            g_.range_too_wide(X, Z, ls)
            g_._range_by_module = True
        X, Y, row_corr = self._pad_rows(X, Y, Z, noise_var, ls, var)                       # (rows first: the padded inducing points must clear them too)
        Z, mu, S_W, S_diag = self._pad_inducing(X, Y, Z, noise_var, mu, S_W, S_diag, ls, var)
        shared = (Z, noise_var, mu, S_W, S_diag, ls, var)
        scaling = float(self.log_pdf_scaling)
        if all(_S(t) == 1 for t in shared):      # (shared X with sampled Y runs natively too: the samples share the Kuf columns)
            logL, info = SVGPLogPdfFn.apply(self._f32_guard(), kind, ard, float(self.jitter), scaling, X, Y, *shared)
        elif noise_var.numel() == noise_var.shape[0] and Y.shape[-1] <= self.PMAX:
            # sampled hyper-parameters / inducing inputs / q(u) (runtime_variable.py:96-118 broadcast semantics): ONE fused call with a sample
            # stride per operand (mxf_svgp_logpdf_sampled; the reference's test_log_pdf_w_samples_* pattern, svgpregression_test.py:142-167)
            logL, info = SVGPSampledLogPdfFn.apply(self._f32_guard(), kind, ard, float(self.jitter), scaling, X, Y, *shared)
            info = ops.merge_info(*info.reshape(-1, 1))
        else:
            # sampled parameters together with per-point / per-output noise: one heteroscedastic call per sample
            S = max(_S(t) for t in (X, Y) + shared)
            pick = lambda t, s: t[s:s + 1] if _S(t) > 1 else t
            outs = [SVGPLogPdfFn.apply(self._f32_guard(), kind, ard, float(self.jitter), scaling, pick(X, s), pick(Y, s), *[pick(t, s) for t in shared])
                    for s in range(S)]
            logL = torch.cat([o[0] for o in outs])
            info = ops.merge_info(*[o[1] for o in outs])
        self._last_info = info
        if row_corr is not None:
            logL = logL - scaling * row_corr
        return logL


    PAD_MIN = 96        # inducing-point counts from here on are padded to the next multiple of 128 (float32 training calls)

    @staticmethod
    def _far_coordinates(X, Z, ls, n, sign):
        """n points along the first coordinate axis, beyond the data and the inducing inputs and 128 length-scales apart: every stationary
        covariance between them and anything real (and among themselves) is < exp(-128): exactly 0 in float32, 1e-56 in the float64 core.
        Their SCALED coordinates stay below 3e4 + range / lengthscale, inside the f16 range the reverse pass splits coordinates into
        (1e6-style offsets overflow it: inf * 0 there).  Built with device ops, no synchronisation; not differentiated."""
        with torch.no_grad():
            # (measured from the EDGE of the data on that side, not from the origin: inputs at an offset -- raw time stamps -- must not push the
            #  padded points out of the f16 range once the reverse pass has centred its operands)
            edge = torch.maximum(X[..., 0].amax(), Z[..., 0].amax()) if sign > 0 else torch.minimum(X[..., 0].amin(), Z[..., 0].amin())
            step = 128.0 * ls.reshape(ls.shape[0], -1)[:, 0].amax()
            far = X[..., :1, :].reshape(-1, X.shape[-1])[:1].expand(n, X.shape[-1]).clone()          # the other coordinates: those of a real point
            far[:, 0] = edge + sign * step * (1.0 + torch.arange(n, dtype=X.dtype, device=X.device))
        return far

    def _pad_inducing(self, X, Y, Z, noise_var, mu, S_W, S_diag, ls, var):
        """The split-GEMM float32 training path needs M % 16 == 0 (its 256-row tiles and the whitened form M % 256 / 128 == 0); a natural choice
        such as M = 1000 or 500 would fall to the generic float32 kernels (75 instead of 21 ms per step at M = 1000, and float64 above cond
        3e3).  Here M is padded to the next multiple of 128 with DECOUPLED inducing points (_far_coordinates: every covariance with the data
        and with the other inducing points is 0, so Kuu gains a diagonal block (variance + jitter) I) and q(u) on them equal to that prior
        (mean 0, no W, diagonal variance + jitter): the padded block adds 0 to the bound and to every gradient (its KL term is stationary
        at s = variance + jitter), autograd drops the padded slices.  Exact in the algebra; in float32 the padded diagonal differs from the
        core's float64 (variance + jitter) by 1e-8 relative, a second-order 1e-16 in the bound."""
        M = Z.shape[-2]
        if not (X.is_cuda and X.dtype == torch.float32 and torch.is_grad_enabled() and M >= self.PAD_MIN and M % 128 != 0):
            return Z, mu, S_W, S_diag
        if noise_var.numel() != noise_var.shape[0] or Y.shape[-1] > self.PMAX or X.shape[-1] > 16:     # paths that never take the split kernels
            return Z, mu, S_W, S_diag
        npad = (M + 127) // 128 * 128 - M
        Q, P = Z.shape[-1], mu.shape[-1]
        far = self._far_coordinates(X, Z, ls, npad, 1.0)
        Zp = torch.cat([Z, far.unsqueeze(0).expand(Z.shape[0], npad, Q)], -2)
        mup = torch.cat([mu, torch.zeros(mu.shape[0], npad, P, dtype=mu.dtype, device=mu.device)], -2)
        Wp = torch.nn.functional.pad(S_W, (0, npad, 0, npad))
        Sd = max(S_diag.shape[0], var.shape[0])
        sdp = torch.cat([S_diag.expand(Sd, M), (var.reshape(var.shape[0], 1) + float(self.jitter)).expand(Sd, npad)], -1)
        return Zp, mup, Wp, sdp

    def _pad_rows(self, X, Y, Z, noise_var, ls, var):
        """The same for the DATA rows: the split kernels need B % 16 == 0 (the whitened form S B % 256 == 0), and a data set of 65 000 or 20 011
        rows would fall to the generic float32 kernels (3x slower).  B is padded to the next multiple of 256 with rows whose inputs lie on the
        NEGATIVE side of the first coordinate (_far_coordinates: every covariance with the inducing points, padded ones included, is 0) and
        whose targets are 0: such a row contributes exactly -P/2 (log 2 pi + log noise) - P/2 variance / noise to the data term
        (svgp_regression.py:93-107 with k_n = 0, y_n = 0), which is subtracted again in closed form (autograd carries the subtraction into the
        noise and variance gradients)."""
        B, P, M = X.shape[-2], Y.shape[-1], Z.shape[-2]
        if not (X.is_cuda and X.dtype == torch.float32 and torch.is_grad_enabled() and B >= 256 and B % 256 != 0 and M >= self.PAD_MIN):      # (M itself is padded next)
            return X, Y, None
        if noise_var.numel() != noise_var.shape[0] or P > self.PMAX or X.shape[-1] > 16:
            return X, Y, None
        npad = (B + 255) // 256 * 256 - B
        far = self._far_coordinates(X, Z, ls, npad, -1.0)
        Xp = torch.cat([X, far.unsqueeze(0).expand(X.shape[0], npad, X.shape[-1])], -2)
        Yp = torch.cat([Y, torch.zeros(Y.shape[0], npad, P, dtype=Y.dtype, device=Y.device)], -2)
        nz, vk = noise_var.reshape(noise_var.shape[0]), var.reshape(var.shape[0])
        corr = npad * (-0.5 * P * (1.8378770664093453 + torch.log(nz)) - 0.5 * P * vk / nz)
        return Xp, Yp, corr

    def _compute_materialised(self, F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params):
        """Combination kernels (add_kernel.py:44-68, multiply_kernel.py:44-67): Kuu / Kuf / Kdiag come from kern.K (each sub-kernel one
        mxf_gram pass with its own reverse mode) and the bound from mxf_svgp_logpdf_mat, one call per sample.
        float32 above the explicit form's condition limit (this path has no whitened form): the WHOLE evaluation runs in float64, Gram
        matrices included -- a float32 Kuu / Kuf widened afterwards still carries its 1e-7 rounding, which cond(Kuu) amplifies (r04 sweep:
        3e-5 ... 3e-4 on the bound at cond 1e3 ... 1e6 with the widened-afterwards form)."""
        from ._fused import Float32Guard
        if X.is_cuda and X.dtype == torch.float32 and Float32Guard.enabled and Float32Guard.force is None:
            g = self._f32_guard()
            g.no_whitened_form = True            # (reports: this owner's calls run in float64 above the explicit limit)

            def wide():
                g._widened_by_owner = True       # (the float64 operands below belong to a float32 model: not "float64 inputs" in the report)
                try:
                    return self._compute_materialised(F, *[t.double() for t in (X, Y, Z, noise_var, mu, S_W, S_diag)], kern,
                                                      {k: v.double() for k, v in kern_params.items()}).float()
                finally:
                    g._widened_by_owner = False
            g.poll(X.device)
            if g.tier != Float32Guard.EXPLICIT:
                return wide()
            first = not g._checked_first
            g._owner_reruns = first              # an ill-conditioned first call is recomputed HERE, Gram matrices included: the bridge's own
            try:                                 # widened-afterwards rerun would be a third evaluation
                out = self._materialised_core(F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params)
            finally:
                g._owner_reruns = False
            return wide() if (first and g.tier != Float32Guard.EXPLICIT) else out        # (an owner's first call checks synchronously)
        return self._materialised_core(F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params)

    def _materialised_core(self, F, X, Y, Z, noise_var, mu, S_W, S_diag, kern, kern_params):
        Kuu = kern.K(F, Z, **kern_params)
        Kuf = kern.K(F, Z, X, **kern_params)
        Kdiag = kern.Kdiag(F, X, **kern_params)
        ops_in = (Kuu, Kuf, Kdiag, Y, noise_var, mu, S_W, S_diag)
        if all(_S(t) == 1 for t in (Kuu, Kuf, Kdiag, noise_var, mu, S_W, S_diag)):
            # shared inputs and parameters, Y possibly sampled (a hidden layer of a deep GP): ONE call for all samples of Y
            logL, info = SVGPMatLogPdfFn.apply(self._f32_guard(), float(self.jitter), float(self.log_pdf_scaling), *ops_in)
            self._last_info = info
            return logL * 1.0
        S = max(_S(t) for t in ops_in)
        pick = lambda t, s: t[s:s + 1] if _S(t) > 1 else t
        outs = [SVGPMatLogPdfFn.apply(self._f32_guard(), float(self.jitter), float(self.log_pdf_scaling), *[pick(t, s) for t in ops_in]) for s in range(S)]
        self._last_info = ops.merge_info(*[o[1] for o in outs])
        return torch.cat([o[0] for o in outs])


class SVGPRegressionMeanVariancePrediction(SamplingAlgorithm):
    """svgp_regression.py:112-189 (note: no arrays_as_samples; var gets a trailing unit axis, :170)."""

    def __init__(self, model, posterior, observed, noise_free=True, diagonal_variance=True, jitter=0.):
        super(SVGPRegressionMeanVariancePrediction, self).__init__(model=model, observed=observed, extra_graphs=[posterior])
        self.jitter = jitter
        self.noise_free = noise_free
        self.diagonal_variance = diagonal_variance

    def _moments(self, F, variables):
        X = variables[self.model.X]
        N = X.shape[-2]
        Z = variables[self.model.inducing_inputs]
        noise_var = variables[self.model.noise_var]
        mu = variables[self.graphs[1].qU_mean]
        S_W = variables[self.graphs[1].qU_cov_W]
        S_diag = variables[self.graphs[1].qU_cov_diag]
        M = Z.shape[-2]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        mean_fn = variables[self.model.mean] if self.model.F.factor.has_mean else None
        # r04: a float32 prediction that records no autograd graph is EVALUATED in float64 (inputs widened, moments narrowed).  The reference
        # factors Kuu in the model's dtype (svgp_regression.py:146-154): in float32 that costs cond(Kuu) 2^-24 of the posterior moments -- 1e-3 ..
        # 1e-1 at a trained model's condition numbers, where north_star asks for 1e-5.  (Differentiable rollouts keep their dtype.)
        wide = X.dtype == torch.float32 and X.is_cuda and not torch.is_grad_enabled()
        if wide:
            X, Z, noise_var, mu, S_W, S_diag = [t.double() for t in (X, Z, noise_var, mu, S_W, S_diag)]
            kern_params = {k: v.double() for k, v in kern_params.items()}
            mean_fn = None if mean_fn is None else mean_fn.double()
        fold = None                 # S samples of the test inputs against one posterior: fold them into columns (gp_regression.py here)
        if self.diagonal_variance and X.shape[0] > 1 and all(t.shape[0] == 1 for t in [Z, noise_var, mu, S_W, S_diag] + list(kern_params.values())):
            fold = tuple(X.shape[:2])
            X = X.reshape(1, fold[0] * fold[1], X.shape[-1])
        with torch.no_grad():       # everything that does not depend on the test inputs
            S = ops.gemm(S_W, S_W, transB=True) + ops.make_diagonal(S_diag)                 # :145 make_diagonal
            Kuu = kern.K(F, Z, **kern_params).contiguous().clone()
            if self.jitter > 0.:
                Kuu = Kuu + torch.eye(M, dtype=Z.dtype, device=Z.device) * self.jitter
            L, _ = ops.potrf_(Kuu)
            Ls, _ = ops.potrf_(S)
            LinvLs = ops.trsm_(L, Ls.clone())
            Linvmu = ops.trsm_(L, mu.contiguous().clone())
            LinvSLinvT = ops.gemm(LinvLs, LinvLs, transB=True)
            wv = ops.trsm_(L, Linvmu.clone(), transpose=True)
        Kxt = kern.K(F, Z, X, **kern_params)
        mu_t = lin.gemm(Kxt, wv, transA=True)
        if fold is not None:
            mu_t = mu_t.reshape(fold + (mu_t.shape[-1],))
        if mean_fn is not None:
            mu_t = mu_t + mean_fn
        if torch.is_grad_enabled() and L.shape[0] == 1:
            LinvKxt = lin.gemm(ops.trtri(L), Kxt)        # rollout: one GEMM (and one in the reverse pass) instead of a chain of panel solves
        else:
            LinvKxt = lin.trsm(L, Kxt)
        tmp = lin.gemm(LinvSLinvT, LinvKxt)
        if self.diagonal_variance:
            Ktt = kern.Kdiag(F, X, **kern_params)
            var = Ktt - lin.coldot(LinvKxt, LinvKxt) + lin.coldot(tmp, LinvKxt)
            if fold is not None:
                var = var.reshape(fold)
            var = var.unsqueeze(-1)
            if not self.noise_free:
                var = var + noise_var
        elif torch.is_grad_enabled():
            var = kern.K(F, X, **kern_params) - lin.gemm(LinvKxt, LinvKxt, transA=True) + lin.gemm(LinvKxt, tmp, transA=True)
            var = var.unsqueeze(-1)
            if not self.noise_free:
                var = var + torch.eye(N, dtype=X.dtype, device=X.device).reshape(1, N, N, 1) * noise_var.unsqueeze(-2)
        else:
            Ktt = kern.K(F, X, **kern_params).contiguous().clone()
            var = ops.gemm(LinvKxt, LinvKxt, transA=True, alpha=-1.0, beta=1.0, out=Ktt)
            var = ops.gemm(LinvKxt, tmp, transA=True, alpha=1.0, beta=1.0, out=var)
            var = var.unsqueeze(-1)
            if not self.noise_free:
                var = var + torch.eye(N, dtype=X.dtype, device=X.device).reshape(1, N, N, 1) * noise_var.unsqueeze(-2)
        if wide:
            mu_t, var = mu_t.float(), var.float()
        return mu_t, var

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):      # differentiable w.r.t. the test inputs only (PILCO rollouts); parameters are constants here
            mu, var = self._moments(F, variables)
        outcomes = {self.model.Y.uuid: (mu, var)}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes


class SVGPRegressionSamplingPrediction(SVGPRegressionMeanVariancePrediction):
    """svgp_regression.py:192-280."""

    def __init__(self, model, posterior, observed, rand_gen=None, noise_free=True, diagonal_variance=True, jitter=0.):
        super(SVGPRegressionSamplingPrediction, self).__init__(model, posterior, observed, noise_free, diagonal_variance, jitter)
        from ...components.distributions.random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):
            mu, var = self._moments(F, variables)      # `jitter` goes on Kuu (:236-238), as in the mean/variance algorithm
            out_shape = (self.num_samples,) + tuple(mu.shape[1:])
            die = self._rand_gen.sample_normal(shape=out_shape, dtype=mu.dtype, ctx=mu.device)
            if self.diagonal_variance:
                samples = mu + die * torch.sqrt(var)
            else:
                cov = var[..., 0]                       # (:262-267: the reference adds no jitter to the predictive covariance)
                Lc, info = lin.chol(cov)                # differentiable when the covariance is (reverse-mode Cholesky): the reference's autograd flows through potrf here
                self._last_info = info
                samples = mu + lin.gemm(Lc, die)
        outcomes = {self.model.Y.uuid: samples}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes


class SVGPRegression(Module):
    """svgp_regression.py:283-457."""

    row_additive = True      # the bound is a sum over data rows plus -KL(q(u) || p(u)) (svgp_regression.py:98-109): row shards add up

    def __init__(self, X, kernel, noise_var, inducing_inputs=None, num_inducing=10, mean=None, rand_gen=None, dtype=None, ctx=None):
        if not isinstance(X, Variable):
            X = Variable(value=X)
        if not isinstance(noise_var, Variable):
            noise_var = Variable(value=noise_var)
        if inducing_inputs is None:   # :317-320: global NumPy RNG at model-definition time
            inducing_inputs = Variable(shape=(num_inducing, kernel.input_dim), initial_value=np.random.randn(num_inducing, kernel.input_dim))
        inputs = [('X', X), ('inducing_inputs', inducing_inputs), ('noise_var', noise_var)]
        if mean is not None:
            inputs.append(('mean', mean))
        self._has_mean = mean is not None
        object.__setattr__(self, 'kernel', kernel)
        super(SVGPRegression, self).__init__(inputs=inputs, outputs=None, input_names=[k for k, _ in inputs],
                                             output_names=['random_variable'], rand_gen=rand_gen, dtype=dtype, ctx=ctx)

    def _generate_outputs(self, output_shapes=None):
        shape = output_shapes['random_variable']
        Y_shape = tuple(self.X.shape[:-1]) + (1,) if shape is None else shape
        self.set_outputs([Variable(shape=Y_shape)])

    def _build_module_graphs(self):
        Y = self.random_variable
        graph = ModuleGraph(name='sparsegp_regression')
        graph.X = self.X
        graph.inducing_inputs = self.inducing_inputs
        M = self.inducing_inputs.shape[0]
        graph.noise_var = self.noise_var
        if self._has_mean:
            graph.mean = self.mean
        graph.F = SimpleNamespace(factor=SimpleNamespace(has_mean=self._has_mean, dtype=self.dtype, kernel=self.kernel))
        graph.Y = Y
        graph.kernel = self.kernel
        for n, v in self.kernel.parameters.items():
            setattr(graph, n, v)
        post = ModuleGraph(name='svgp_posterior')      # :377-380
        post.qU_cov_diag = Variable(shape=(M,), transformation=PositiveTransformation())
        post.qU_cov_W = Variable(shape=(M, M))
        post.qU_mean = Variable(shape=(M, Y.shape[-1]))
        return graph, [post]

    def _attach_default_inference_algorithms(self):
        observed = [v for _, v in self.inputs] + [v for _, v in self.outputs]
        self.attach_log_pdf_algorithms(targets=self.output_names, conditionals=self.input_names,
                                       algorithm=SVGPRegressionLogPdf(self._module_graph, self._extra_graphs[0], observed),
                                       alg_name='svgp_log_pdf')
        observed = [v for _, v in self.inputs]
        # svgp_regression.py:399-403: draw_samples <- ForwardSamplingAlgorithm over the generative internal graph U -> F -> Y
        self._sampling_graph = build_sparse_gp_sampling_model(self, 'sparsegp_regression')
        self.attach_draw_samples_algorithms(targets=self.output_names, conditionals=self.input_names,
                                            algorithm=ForwardSamplingAlgorithm(self._sampling_graph, observed), alg_name='svgp_sampling')
        self.attach_prediction_algorithms(targets=self.output_names, conditionals=self.input_names,
                                          algorithm=SVGPRegressionMeanVariancePrediction(self._module_graph, self._extra_graphs[0], observed),
                                          alg_name='svgp_predict')

    @staticmethod
    def define_variable(X, kernel, noise_var, shape=None, inducing_inputs=None, num_inducing=10, mean=None, rand_gen=None,
                        dtype=None, ctx=None):
        gp = SVGPRegression(X=X, kernel=kernel, noise_var=noise_var, inducing_inputs=inducing_inputs, num_inducing=num_inducing,
                            mean=mean, rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        gp._generate_outputs({'random_variable': shape})
        return gp.random_variable

    @property
    def random_variable(self):
        return self._outputs[0][1]
