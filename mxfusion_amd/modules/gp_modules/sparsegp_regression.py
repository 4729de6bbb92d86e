"""SparseGPRegression (Titsias) module and its algorithms (mxfusion/modules/gp_modules/sparsegp_regression.py:32-430).

  SparseGPRegressionLogPdf.compute                 -> mxf_sgp_logpdf (streaming Psi2 / psi1 statistics, C = Kuu + Psi2/s2, closed-form reverse mode)
  SparseGPRegressionMeanVariancePrediction.compute -> mxf_gram + mxf_trsm + mxf_gemm + mxf_coldot
  SparseGPRegressionSamplingPrediction.compute     -> the same moments + mxf_potrf + mxf_gemm (trmm) on injected noise
"""
from types import SimpleNamespace

import numpy as np
import torch

from ... import ops
from ...components.variables.variable import Variable
from ...inference.inference_alg import SamplingAlgorithm
from ...inference.variational import VariationalInference
from ..module import Module, ModuleGraph
from ...inference.forward_sampling import ForwardSamplingAlgorithm
from ._fused import SGPLogPdfFn
from ._sampling_graph import build_sparse_gp_sampling_model
from ...components.distributions.gp import _linalg as lin
from .gp_regression import _grad_mode


class SparseGPRegressionLogPdf(VariationalInference):
    """sparsegp_regression.py:32-108."""

    def __init__(self, model, posterior, observed, jitter=0.):
        super(SparseGPRegressionLogPdf, self).__init__(model=model, posterior=posterior, observed=observed)
        self.log_pdf_scaling = 1          # set but never used by the reference either
        self.jitter = jitter

    def _f32_guard(self):
        """This algorithm object's float32 validity guard (its own condition slot; levels explicit float32 / float64: _fused.Float32Guard)."""
        g = getattr(self, '_guard', None)
        if g is None:
            from ._fused import Float32Guard
            g = self._guard = Float32Guard('sgp')
        return g

    def compute(self, F, variables):
        X = variables[self.model.X]
        Y = variables[self.model.Y]
        Z = variables[self.model.inducing_inputs]
        noise_var = variables[self.model.noise_var]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        spec = kern.fused_spec()
        if spec is None:
            return self._compute_materialised(F, variables, X, Y, Z, noise_var, kern, kern_params)
        kind, ard = spec
        ls = kern_params[kern.name + '_lengthscale']
        var = kern_params[kern.name + '_variance']
        if self.model.F.factor.has_mean:
            Y = Y - variables[self.model.mean]
        args = (X, Y, Z, noise_var, ls, var)
        S = max(t.shape[0] for t in args)
        pick = lambda t, s: t[s:s + 1] if t.shape[0] > 1 else t
        outs = [SGPLogPdfFn.apply(self._f32_guard(), kind, ard, float(self.jitter), *[pick(t, s) for t in args]) for s in range(S)]
        logL = torch.cat([o[0] for o in outs])
        with torch.no_grad():      # :99-106 persist sample 0 only
            self.set_parameter(variables, self.graphs[1].wv, outs[0][1])
            self.set_parameter(variables, self.graphs[1].L, outs[0][2])
            self.set_parameter(variables, self.graphs[1].LA, outs[0][3])
        return logL


    def _compute_materialised(self, F, variables, X, Y, Z, noise_var, kern, kern_params):
        """Combination kernels (Add / Multiply), Linear / Bias / White and kernels with active_dims have no fused description: the
        reference's own operator sequence (sparsegp_regression.py:67-106) on materialised Kuu / Kuf / Kdiag from kern.K (one mxf_gram pass
        per sub-kernel, each with its reverse mode), through the differentiable mxf_potrf / mxf_trsm / mxf_gemm bridges of _linalg."""
        import math
        if self.model.F.factor.has_mean:
            Y = Y - variables[self.model.mean]
        # r04: float32 on the GPU is evaluated in float64 inside (inputs widened, bound and the persisted posterior narrowed): two float32
        # Cholesky factorisations in a row (Kuu, then I + ...) lose cond(Kuu) 2^-24 of the bound, and this generic path is not the fast one anyway
        narrow = X.is_cuda and X.dtype == torch.float32
        if narrow:
            X, Y, Z, noise_var = [t.double() for t in (X, Y, Z, noise_var)]
            kern_params = {k: v.double() for k, v in kern_params.items()}
        D, M = Y.shape[-1], Z.shape[-2]
        eye = torch.eye(M, dtype=Z.dtype, device=Z.device).unsqueeze(0)
        nv = noise_var.unsqueeze(-2)                                   # (S, 1, 1)
        Kuu = kern.K(F, Z, **kern_params)
        if self.jitter > 0.:
            Kuu = Kuu + eye * self.jitter
        Kuf = kern.K(F, Z, X, **kern_params)
        Kff_diag = kern.Kdiag(F, X, **kern_params)
        L, info = lin.chol(Kuu)                                        # :83
        LinvKuf = lin.trsm(L, Kuf)                                     # :84
        A = eye + lin.gemm(LinvKuf, LinvKuf, transB=True) / nv         # :86
        LA, info2 = lin.chol(A)                                        # :87
        LAInvLinvKufY = lin.trsm(LA, lin.gemm(LinvKuf, Y))             # :92
        sumlog = torch.log(torch.diagonal(LA, dim1=-2, dim2=-1)).sum(-1)
        logL = -D * sumlog                                                                                        # :94
        logL = logL - ((Y ** 2) / nv + math.log(2 * math.pi) + torch.log(nv)).sum(-1).sum(-1) / 2               # :95
        logL = logL + ((LAInvLinvKufY ** 2) / (2 * nv ** 2)).sum(-1).sum(-1)                                      # :96
        logL = logL - D * (Kff_diag / (2 * noise_var)).sum(-1)                                                    # :97
        logL = logL + D * ((LinvKuf ** 2) / (2. * nv)).sum(-1).sum(-1)                                            # :98
        self._last_info = ops.merge_info(info, info2)
        with torch.no_grad():      # :99-106 persist sample 0 only
            wv = ops.trsm_(L[:1].contiguous(), ops.trsm_(LA[:1].contiguous(), LAInvLinvKufY[:1].detach().contiguous().clone(), transpose=True),
                           transpose=True) / nv[:1]
            nf = (lambda t: t.float()) if narrow else (lambda t: t)
            self.set_parameter(variables, self.graphs[1].wv, nf(wv[0]))
            self.set_parameter(variables, self.graphs[1].L, nf(L[0].detach()))
            self.set_parameter(variables, self.graphs[1].LA, nf(LA[0].detach()))
        return logL.float() if narrow else logL


class SparseGPRegressionMeanVariancePrediction(SamplingAlgorithm):
    """sparsegp_regression.py:111-174."""

    def __init__(self, model, posterior, observed, target_variables=None, noise_free=True, diagonal_variance=True):
        super(SparseGPRegressionMeanVariancePrediction, self).__init__(model=model, observed=observed, extra_graphs=[posterior],
                                                                       target_variables=target_variables)
        self.noise_free = noise_free
        self.diagonal_variance = diagonal_variance

    def _moments(self, F, variables, jitter=0.):
        if True:
            X = variables[self.model.X]
            N = X.shape[-2]
            Z = variables[self.model.inducing_inputs]
            noise_var = variables[self.model.noise_var]
            L = variables[self.graphs[1].L]
            LA = variables[self.graphs[1].LA]
            wv = variables[self.graphs[1].wv]
            kern = self.model.kernel
            kern_params = kern.fetch_parameters(variables)
            mean_fn = variables[self.model.mean] if self.model.F.factor.has_mean else None
            # r04: a float32 prediction that records no autograd graph is EVALUATED in float64 (the stored posterior widened, the moments
            # narrowed): the triangular solves against L / LA lose ~ sqrt(cond(Kuu)) 2^-24 in float32 (see svgp_regression.py here)
            wide = X.dtype == torch.float32 and X.is_cuda and not torch.is_grad_enabled()
            if wide:
                X, Z, noise_var, L, LA, wv = [t.double() for t in (X, Z, noise_var, L, LA, wv)]
                kern_params = {k: v.double() for k, v in kern_params.items()}
                mean_fn = None if mean_fn is None else mean_fn.double()
            fold = None             # S samples of the test inputs against one posterior: fold them into columns (gp_regression.py here)
            if self.diagonal_variance and X.shape[0] > 1 and all(t.shape[0] == 1 for t in [Z, noise_var, L, LA, wv] + list(kern_params.values())):
                fold = tuple(X.shape[:2])
                X = X.reshape(1, fold[0] * fold[1], X.shape[-1])
            Kxt = kern.K(F, Z, X, **kern_params)
            mu = lin.gemm(Kxt, wv, transA=True)
            if fold is not None:
                mu = mu.reshape(fold + (mu.shape[-1],))
            if mean_fn is not None:
                mu = mu + mean_fn
            LinvKxt = lin.trsm(L, Kxt)
            LAinvLinvKxt = lin.trsm(LA, LinvKxt)
            if self.diagonal_variance:
                var = kern.Kdiag(F, X, **kern_params) - lin.coldot(LinvKxt, LinvKxt) + lin.coldot(LAinvLinvKxt, LAinvLinvKxt)
                if fold is not None:
                    var = var.reshape(fold)
                if not self.noise_free:
                    var = var + noise_var
            elif torch.is_grad_enabled():
                var = kern.K(F, X, **kern_params) - lin.gemm(LinvKxt, LinvKxt, transA=True) + lin.gemm(LAinvLinvKxt, LAinvLinvKxt, transA=True)
                if not self.noise_free:
                    var = var + torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0) * noise_var.unsqueeze(-2)
            else:
                Ktt = kern.K(F, X, **kern_params).contiguous().clone()
                var = ops.gemm(LinvKxt, LinvKxt, transA=True, alpha=-1.0, beta=1.0, out=Ktt)
                var = ops.gemm(LAinvLinvKxt, LAinvLinvKxt, transA=True, alpha=1.0, beta=1.0, out=var)
                if not self.noise_free:
                    var = var + torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0) * noise_var.unsqueeze(-2)
            if wide:
                mu, var = mu.float(), var.float()
        return mu, var

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):       # differentiable w.r.t. the test inputs (PILCO rollouts)
            mu, var = self._moments(F, variables)
        outcomes = {self.model.Y.uuid: (mu, var)}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes


class SparseGPRegressionSamplingPrediction(SparseGPRegressionMeanVariancePrediction):
    """sparsegp_regression.py:177-255: draws from the predictive distribution -- mu + eps sqrt(var) (diagonal) or mu + chol(cov + jitter I) eps."""

    def __init__(self, model, posterior, observed, rand_gen=None, noise_free=True, diagonal_variance=True, jitter=0.):
        super(SparseGPRegressionSamplingPrediction, self).__init__(model, posterior, observed, noise_free=noise_free,
                                                                   diagonal_variance=diagonal_variance)
        from ...components.distributions.random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen
        self.jitter = jitter

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):
            mu, var = self._moments(F, variables)
            out_shape = (self.num_samples,) + tuple(mu.shape[1:])
            die = self._rand_gen.sample_normal(shape=out_shape, dtype=mu.dtype, ctx=mu.device)
            if self.diagonal_variance:
                samples = mu + die * torch.sqrt(var.unsqueeze(-1))                       # :228-234
            else:
                N = var.shape[-1]
                cov = var
                if self.jitter > 0.:                                                      # :241-242
                    cov = cov + torch.eye(N, dtype=cov.dtype, device=cov.device).unsqueeze(0) * self.jitter
                Lc, info = lin.chol(cov)                                                  # differentiable when the covariance is (reverse-mode Cholesky)
                self._last_info = info
                samples = mu + lin.gemm(Lc, die)                                          # trmm(L, die): L is lower with a zero upper part
        outcomes = {self.model.Y.uuid: samples}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes




class SparseGPRegression(Module):
    """sparsegp_regression.py:258-430."""

    def __init__(self, X, kernel, noise_var, inducing_inputs=None, num_inducing=10, mean=None, rand_gen=None, dtype=None, ctx=None):
        if not isinstance(X, Variable):
            X = Variable(value=X)
        if not isinstance(noise_var, Variable):
            noise_var = Variable(value=noise_var)
        if inducing_inputs is None:
            inducing_inputs = Variable(shape=(num_inducing, kernel.input_dim), initial_value=np.random.randn(num_inducing, kernel.input_dim))
        inputs = [('X', X), ('inducing_inputs', inducing_inputs), ('noise_var', noise_var)]
        if mean is not None:
            inputs.append(('mean', mean))
        self._has_mean = mean is not None
        object.__setattr__(self, 'kernel', kernel)
        super(SparseGPRegression, self).__init__(inputs=inputs, outputs=None, input_names=[k for k, _ in inputs],
                                                 output_names=['random_variable'], rand_gen=rand_gen, dtype=dtype, ctx=ctx)

    def _generate_outputs(self, output_shapes=None):
        shape = output_shapes['random_variable']
        self.set_outputs([Variable(shape=tuple(self.X.shape[:-1]) + (1,) if shape is None else shape)])

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
        post = ModuleGraph(name='sparsegp_posterior')       # what prediction needs (:353-357)
        post.L = Variable(shape=(M, M))
        post.LA = Variable(shape=(M, M))
        post.wv = Variable(shape=(M, Y.shape[-1]))
        for v in (post.L, post.LA, post.wv):
            v.is_posterior_cache = True
        return graph, [post]

    def _attach_default_inference_algorithms(self):
        observed = [v for _, v in self.inputs] + [v for _, v in self.outputs]
        self.attach_log_pdf_algorithms(targets=self.output_names, conditionals=self.input_names,
                                       algorithm=SparseGPRegressionLogPdf(self._module_graph, self._extra_graphs[0], observed),
                                       alg_name='sgp_log_pdf')
        observed = [v for _, v in self.inputs]
        # sparsegp_regression.py:374-378: draw_samples <- ForwardSamplingAlgorithm over the generative internal graph U -> F -> Y
        self._sampling_graph = build_sparse_gp_sampling_model(self, 'sparsegp_regression')
        self.attach_draw_samples_algorithms(targets=self.output_names, conditionals=self.input_names,
                                            algorithm=ForwardSamplingAlgorithm(self._sampling_graph, observed), alg_name='sgp_sampling')
        self.attach_prediction_algorithms(targets=self.output_names, conditionals=self.input_names,
                                          algorithm=SparseGPRegressionMeanVariancePrediction(self._module_graph, self._extra_graphs[0], observed),
                                          alg_name='sgp_predict')

    @staticmethod
    def define_variable(X, kernel, noise_var, shape=None, inducing_inputs=None, num_inducing=10, mean=None, rand_gen=None,
                        dtype=None, ctx=None):
        gp = SparseGPRegression(X=X, kernel=kernel, noise_var=noise_var, inducing_inputs=inducing_inputs, num_inducing=num_inducing,
                                mean=mean, rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        gp._generate_outputs({'random_variable': shape})
        return gp.random_variable

    @property
    def random_variable(self):
        return self._outputs[0][1]
