"""GaussianProcess distribution (mxfusion/components/distributions/gp/gp.py:24-163): Y ~ N(mean(X), K(X, X)), no noise term.
log-pdf: HIP Gram (kernel.K) + mxf_potrf / mxf_trsm with the closed-form reverse mode; samples: L eps through mxf_gemm."""
import torch

from .... import ops
from ..distribution import Distribution
from ...variables.variable import Variable
from ._linalg import CholLogPdfFn, chol, gemm


class GaussianProcess(Distribution):
    def __init__(self, X, kernel, mean=None, rand_gen=None, dtype=None, ctx=None):
        inputs = [('X', X)] + [(k, v) for k, v in kernel.parameters.items()]
        self._has_mean = mean is not None
        if mean is not None:
            inputs.append(('mean', mean))
        super(GaussianProcess, self).__init__(inputs=inputs, outputs=None, input_names=[k for k, _ in inputs],
                                              output_names=['random_variable'], rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        self.kernel = kernel

    @property
    def has_mean(self):
        return self._has_mean

    @staticmethod
    def define_variable(X, kernel, shape=None, mean=None, rand_gen=None, dtype=None, ctx=None):
        """gp.py:62-93: default shape = X.shape[:-1] + (1,)."""
        gp = GaussianProcess(X=X, kernel=kernel, mean=mean, rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        gp._generate_outputs(shape=tuple(X.shape[:-1]) + (1,) if shape is None else shape)
        return gp.random_variable

    def log_pdf_impl(self, X, random_variable, F=None, **kernel_params):
        """gp.py:95-122."""
        mean = kernel_params.pop('mean', None) if self._has_mean else None
        K = self.kernel.K(F, X, **kernel_params)
        if mean is not None:
            random_variable = random_variable - mean
        logL, _, _, info = CholLogPdfFn.apply(K, random_variable)
        self._last_info = info
        return logL * self.log_pdf_scaling

    def draw_samples_impl(self, X, rv_shape, num_samples=1, F=None, **kernel_params):
        """gp.py:124-153: L eps (+ mean); differentiable w.r.t. X and the kernel parameters (reparameterised draw) through the
        reverse-mode Cholesky of _linalg.CholFn, as the reference is through linalg.potrf."""
        mean = kernel_params.pop('mean', None) if self._has_mean else None
        K = self.kernel.K(F, X, **kernel_params)
        L, info = chol(K)
        self._last_info = info
        out_shape = (num_samples,) + tuple(rv_shape)
        die = self._rand_gen.sample_normal(shape=out_shape, dtype=X.dtype, ctx=X.device)
        rv = gemm(L, die.reshape(out_shape).contiguous())                        # linalg.trmm(L, die): L is lower with a zero upper part
        if mean is not None:
            rv = rv + mean
        return rv
