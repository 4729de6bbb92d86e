"""ConditionalGaussianProcess distribution (mxfusion/components/distributions/gp/cond_gp.py:25-235):
Y ~ N(K*c Kcc^-1 (Yc - g(Xc)) + g(X),  K** - K*c Kcc^-1 K*c^T).
The conditioning algebra is the reference's own (cond_gp.py:164-177): Lcc = chol(Kcc), V = Lcc^-1 Kc, cov = K - V^T V,
mean = V^T Lcc^-1 Yc -- error amplified by sqrt(cond(Kcc)), not cond(Kcc); reverse mode through _linalg.CholFn / TrsmFn / MatmulFn."""
import torch

from .... import ops
from ....common.exceptions import ModelSpecificationError
from ..distribution import Distribution
from ._linalg import CholLogPdfFn, chol, gemm, trsm


class ConditionalGaussianProcess(Distribution):
    def __init__(self, X, X_cond, Y_cond, kernel, mean=None, mean_cond=None, rand_gen=None, dtype=None, ctx=None):
        if (mean is None) and (mean_cond is not None):
            raise ModelSpecificationError("The argument mean and mean_cond need to be both specified.")
        inputs = [('X', X), ('X_cond', X_cond), ('Y_cond', Y_cond)] + [(k, v) for k, v in kernel.parameters.items()]
        self._has_mean = mean is not None
        self._has_mean_cond = mean_cond is not None
        if mean is not None:
            inputs.append(('mean', mean))
        if mean_cond is not None:
            inputs.append(('mean_cond', mean_cond))
        super(ConditionalGaussianProcess, self).__init__(inputs=inputs, outputs=None, input_names=[k for k, _ in inputs],
                                                         output_names=['random_variable'], rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        self.kernel = kernel

    @property
    def has_mean(self):
        return self._has_mean

    @staticmethod
    def define_variable(X, X_cond, Y_cond, kernel, shape=None, mean=None, mean_cond=None, rand_gen=None, dtype=None, ctx=None):
        gp = ConditionalGaussianProcess(X=X, X_cond=X_cond, Y_cond=Y_cond, kernel=kernel, mean=mean, mean_cond=mean_cond,
                                        rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        gp._generate_outputs(shape=tuple(X.shape[:-1]) + (1,) if shape is None else shape)
        return gp.random_variable

    def _moments(self, F, X, X_cond, Y_cond, kernel_params):
        mean = kernel_params.pop('mean', None) if self._has_mean else None
        mean_cond = kernel_params.pop('mean_cond', None) if self._has_mean_cond else None
        K = self.kernel.K(F, X, **kernel_params)
        Kc = self.kernel.K(F, X_cond, X, **kernel_params)
        Kcc = self.kernel.K(F, X_cond, **kernel_params)
        Lcc, info = chol(Kcc)                                     # cond_gp.py:167
        V = trsm(Lcc, Kc)                                         # Lcc^-1 K_c*  (:168)
        cov = K - gemm(V, V, transA=True)                         # :170 (K - syrk(Lcc^-1 Kc, transpose))
        if mean_cond is not None:
            Y_cond = Y_cond - mean_cond
        LccInvY = trsm(Lcc, Y_cond)                               # :176
        rv_mean = gemm(V, LccInvY, transA=True)                   # :177
        return cov, rv_mean, mean, info

    def log_pdf_impl(self, X, X_cond, Y_cond, random_variable, F=None, **kernel_params):
        """cond_gp.py:124-183.  NOTE the reference sums L^-1 (rv - mean) over the OUTPUT axis before squaring (:179) while the
        log-determinant is still multiplied by D (:182); reproduced as is (identical to the textbook density when D = 1)."""
        D = random_variable.shape[-1]
        cov, rv_mean, mean, info = self._moments(F, X, X_cond, Y_cond, kernel_params)
        if mean is not None:
            random_variable = random_variable - mean
        resid = (random_variable - rv_mean).sum(-1, keepdim=True)
        logL, _, _, info2 = CholLogPdfFn.apply(cov, resid, float(D))
        self._last_info = ops.merge_info(info, info2)
        return logL * self.log_pdf_scaling

    def draw_samples_impl(self, X, X_cond, Y_cond, rv_shape, num_samples=1, F=None, **kernel_params):
        """cond_gp.py:185-223."""
        cov, rv_mean, mean, info = self._moments(F, X, X_cond, Y_cond, dict(kernel_params))
        L, info2 = chol(cov)
        self._last_info = ops.merge_info(info, info2)
        out_shape = (num_samples,) + tuple(rv_shape)
        die = self._rand_gen.sample_normal(shape=out_shape, dtype=X.dtype, ctx=X.device)
        rv = gemm(L, die.reshape(out_shape).contiguous()) + rv_mean
        if mean is not None:
            rv = rv + mean
        return rv
