"""GPRegression module and its algorithms (mxfusion/modules/gp_modules/gp_regression.py:31-428).

Same classes, constructor signatures, attributes (`jitter`, `noise_free`, `diagonal_variance`) and
`compute(F, variables)` contract as the reference; the bodies call the MI355X kernels through the C ABI:
  GPRegressionLogPdf.compute                 -> mxf_gp_logpdf (fused Gram + potrf + trsm + log-det + reverse mode)
  GPRegressionMeanVariancePrediction.compute -> mxf_gram + mxf_trsm + mxf_gemm
  GPRegressionSampling / SamplingPrediction  -> mxf_gram + mxf_potrf + mxf_gemm (trmm)
"""
from types import SimpleNamespace

import torch

from ... import ops
from ...common import config
from ...components.variables.variable import Variable
from ...components.variables.runtime_variable import arrays_as_samples
from ...inference.inference_alg import SamplingAlgorithm
from ...inference.variational import VariationalInference
from ..module import Module, ModuleGraph
from ...components.distributions.gp import _linalg as lin
from ...components.distributions.gp._linalg import CholLogPdfFn
from ._fused import GPLogPdfFn


def _chol_logpdf_generic(F, K, Y):
    """log N(Y | 0, K) per sample from an explicit K (any kernel with an autograd-capable K()): value via
    mxf_potrf / mxf_trsm, reverse mode via dK = 1/2 (alpha alpha^T - P K^-1)."""
    return CholLogPdfFn.apply(K, Y)


class GPRegressionLogPdf(VariationalInference):
    """gp_regression.py:31-76."""

    def __init__(self, model, posterior, observed, jitter=0.):
        super(GPRegressionLogPdf, self).__init__(model=model, posterior=posterior, observed=observed)
        self.log_pdf_scaling = 1     # set but unused by the reference as well
        self.jitter = jitter

    def compute(self, F, variables):
        has_mean = self.model.F.factor.has_mean
        X = variables[self.model.X]
        Y = variables[self.model.Y]
        noise_var = variables[self.model.noise_var]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        if has_mean:
            Y = Y - variables[self.model.mean]
        spec = kern.fused_spec()
        if spec is not None:
            kind, ard = spec
            ls = kern_params[kern.name + '_lengthscale']
            var = kern_params[kern.name + '_variance']
            logL, L, LinvY, info = GPLogPdfFn.apply(kind, ard, float(self.jitter), X, Y, noise_var, ls, var)
            Xs = X
        else:   # combination kernels: K through the kernel's own (HIP, autograd-capable) K(), then Cholesky
            narrow = X.is_cuda and X.dtype == torch.float32        # (r04: float32 evaluated in float64 inside, as the fused call: _fused.GPLogPdfFn)
            if narrow:
                X, Y, noise_var = X.double(), Y.double(), noise_var.double()
                kern_params = {k: v.double() for k, v in kern_params.items()}
            X, Y, noise_var, kern_params = arrays_as_samples(F, [X, Y, noise_var, kern_params])
            N = X.shape[-2]
            eye = torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0)
            K = kern.K(F, X, **kern_params) + eye * (noise_var.unsqueeze(-2) + self.jitter)
            logL, L, LinvY, info = _chol_logpdf_generic(F, K, Y)
            Xs = X
            if narrow:
                logL, L, LinvY, Xs = logL.float(), L.float(), LinvY.float(), X.float()
        self._last_info = info
        with torch.no_grad():      # gp_regression.py:72-75: only sample 0 is persisted
            self.set_parameter(variables, self.posterior.X, Xs[0].detach())
            self.set_parameter(variables, self.posterior.L, L[0])
            self.set_parameter(variables, self.posterior.LinvY, LinvY[0])
        return logL


class GPRegressionSampling(SamplingAlgorithm):
    """gp_regression.py:79-135: prior draws Y = L eps (+ mean)."""

    def __init__(self, model, observed, num_samples=1, target_variables=None, rand_gen=None):
        super(GPRegressionSampling, self).__init__(model=model, observed=observed, num_samples=num_samples, target_variables=target_variables)
        from ...components.distributions.random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen

    def compute(self, F, variables):
        has_mean = self.model.F.factor.has_mean
        X = variables[self.model.X]
        noise_var = variables[self.model.noise_var]
        N = X.shape[-2]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        X, noise_var, kern_params = arrays_as_samples(F, [X, noise_var, kern_params])
        # differentiable w.r.t. X, the noise and the kernel parameters (reverse-mode Cholesky: _linalg.CholFn), as the reference's
        # linalg.potrf is -- a GP prior draw inside a differentiated objective keeps its gradients
        K = kern.K(F, X, **kern_params) + torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0) * noise_var.unsqueeze(-2)
        L, _ = lin.chol(K)
        Y_shape = (N, int(self.model.Y.shape[-1]))
        out_shape = (self.num_samples,) + Y_shape
        die = self._rand_gen.sample_normal(shape=out_shape, dtype=X.dtype, ctx=X.device)
        y_samples = lin.gemm(L, die)          # trmm: L is lower with a clean upper triangle
        if has_mean:
            y_samples = y_samples + variables[self.model.mean]
        samples = {self.model.Y.uuid: y_samples}
        if self.target_variables:
            return tuple(samples[v] for v in self.target_variables)
        return samples


def _grad_mode(Xt):
    """Prediction records an autograd graph only when the test inputs are themselves differentiable (the PILCO rollout feeds the
    policy's actions back in as inputs, pilco_alg.py:76-88); plain prediction from data runs without one."""
    return torch.enable_grad() if (torch.is_grad_enabled() and Xt.requires_grad) else torch.no_grad()


class GPRegressionMeanVariancePrediction(SamplingAlgorithm):
    """gp_regression.py:138-196."""

    def __init__(self, model, posterior, observed, noise_free=True, diagonal_variance=True):
        super(GPRegressionMeanVariancePrediction, self).__init__(model=model, observed=observed, extra_graphs=[posterior])
        self.noise_free = noise_free
        self.diagonal_variance = diagonal_variance

    def _mean_and_V(self, F, variables):
        X = variables[self.model.X]
        noise_var = variables[self.model.noise_var]
        X_cond = variables[self.graphs[1].X]
        L = variables[self.graphs[1].L]
        LinvY = variables[self.graphs[1].LinvY]
        kern = self.model.kernel
        kern_params = kern.fetch_parameters(variables)
        mean_fn = variables[self.model.mean] if self.model.F.factor.has_mean else None
        # r04: a float32 prediction that records no autograd graph is EVALUATED in float64 (the stored posterior L, L^-1 Y widened, the results
        # narrowed by compute()): the solve against L loses ~ sqrt(cond(K + noise I)) 2^-24 in float32.  Differentiable rollouts keep their dtype.
        self._wide = X.dtype == torch.float32 and X.is_cuda and not torch.is_grad_enabled()
        if self._wide:
            X, noise_var, X_cond, L, LinvY = [t.double() for t in (X, noise_var, X_cond, L, LinvY)]
            kern_params = {k: v.double() for k, v in kern_params.items()}
            mean_fn = None if mean_fn is None else mean_fn.double()
        # S samples of the test inputs against ONE posterior (the rollout's trajectories; any sampled-input prediction): the sample axis is
        # folded into the column axis, so the cross Gram is one (N x S*Nt) matrix and every product below one full-width GEMM instead of
        # S skinny ones (at N=1000, S=64, Nt=1: 0.34 ms -> ~0.03 ms per product)
        fold = None
        if self.diagonal_variance and X.shape[0] > 1 and all(t.shape[0] == 1 for t in [X_cond, L, LinvY, noise_var] + list(kern_params.values())):
            fold = tuple(X.shape[:2])
            X = X.reshape(1, fold[0] * fold[1], X.shape[-1])
        Kxt = kern.K(F, X_cond, X, **kern_params)                    # (S, N, Nt)
        if torch.is_grad_enabled() and L.shape[0] == 1 and not L.requires_grad:
            if getattr(self, '_linv_cache', None) is None:
                self._linv_cache = lin._InverseCache()
            LinvKxt = lin.solve_shared(L, Kxt, self._linv_cache)     # rollout: V = (L^-1) Kxt with the cached inverse, one GEMM
        else:
            LinvKxt = lin.trsm(L, Kxt)                               # V = L^-1 Kxt
        mu = lin.gemm(LinvKxt, LinvY, transA=True)                   # V^T LinvY (LinvY broadcast over S by stride 0)
        if fold is not None:
            mu = mu.reshape(fold + (mu.shape[-1],))
        if mean_fn is not None:
            mu = mu + mean_fn
        return X, noise_var, kern, kern_params, LinvKxt, mu, fold

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):
            X, noise_var, kern, kern_params, LinvKxt, mu, fold = self._mean_and_V(F, variables)
            N = X.shape[-2]
            if self.diagonal_variance:
                Ktt = kern.Kdiag(F, X, **kern_params)
                var = Ktt - lin.coldot(LinvKxt, LinvKxt)
                if fold is not None:
                    var = var.reshape(fold)
                if not self.noise_free:
                    var = var + noise_var
            elif torch.is_grad_enabled():
                var = kern.K(F, X, **kern_params) - lin.gemm(LinvKxt, LinvKxt, transA=True)
                if not self.noise_free:
                    var = var + torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0) * noise_var.unsqueeze(-2)
            else:
                Ktt = kern.K(F, X, **kern_params)
                var = ops.gemm(LinvKxt, LinvKxt, transA=True, alpha=-1.0, beta=1.0, out=Ktt.contiguous().clone())
                if not self.noise_free:
                    var = var + torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0) * noise_var.unsqueeze(-2)
            if self._wide:
                mu, var = mu.float(), var.float()
        outcomes = {self.model.Y.uuid: (mu, var)}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes


class GPRegressionSamplingPrediction(GPRegressionMeanVariancePrediction):
    """gp_regression.py:199-275."""

    def __init__(self, model, posterior, observed, rand_gen=None, noise_free=True, diagonal_variance=True, jitter=0.):
        super(GPRegressionSamplingPrediction, self).__init__(model, posterior, observed, noise_free, diagonal_variance)
        from ...components.distributions.random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen
        self.jitter = jitter

    def compute(self, F, variables):
        with _grad_mode(variables[self.model.X]):
            X, noise_var, kern, kern_params, LinvKxt, mu, fold = self._mean_and_V(F, variables)
            N = X.shape[-2]
            out_shape = (self.num_samples,) + tuple(mu.shape[1:])
            die = self._rand_gen.sample_normal(shape=out_shape, dtype=X.dtype, ctx=X.device)
            if self.diagonal_variance:
                var = kern.Kdiag(F, X, **kern_params) - lin.coldot(LinvKxt, LinvKxt)
                if fold is not None:
                    var = var.reshape(fold)
                if not self.noise_free:
                    var = var + noise_var
                samples = mu + die * torch.sqrt(var.unsqueeze(-1))
            elif torch.is_grad_enabled():
                # differentiable w.r.t. the test inputs (the reference's autograd flows through potrf, gp_regression.py:251-268): the covariance
                # through the autograd bridges of mxf_gram / mxf_gemm, its factor through the reverse-mode Cholesky (lin.chol, Murray 2016)
                cov = kern.K(F, X, **kern_params) - lin.gemm(LinvKxt, LinvKxt, transA=True)
                eye = torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0)
                if not self.noise_free:
                    cov = cov + eye * noise_var.unsqueeze(-2)
                if self.jitter > 0.:
                    cov = cov + eye * self.jitter
                Lc, info = lin.chol(cov)
                self._last_info = info
                samples = mu + lin.gemm(Lc, die)
            else:
                cov = ops.gemm(LinvKxt, LinvKxt, transA=True, alpha=-1.0, beta=1.0, out=kern.K(F, X, **kern_params).contiguous().clone())
                eye = torch.eye(N, dtype=X.dtype, device=X.device).unsqueeze(0)
                if not self.noise_free:
                    cov = cov + eye * noise_var.unsqueeze(-2)
                if self.jitter > 0.:
                    cov = cov + eye * self.jitter
                Lc, _ = ops.potrf_(cov.contiguous())
                samples = mu + ops.gemm(Lc, die)
            if self._wide:
                samples = samples.float()
        outcomes = {self.model.Y.uuid: samples}
        if self.target_variables:
            return tuple(outcomes[v] for v in self.target_variables)
        return outcomes


class GPRegression(Module):
    """gp_regression.py:278-428."""

    def __init__(self, X, kernel, noise_var, mean=None, rand_gen=None, dtype=None, ctx=None):
        if not isinstance(X, Variable):
            X = Variable(value=X)
        if not isinstance(noise_var, Variable):
            noise_var = Variable(value=noise_var)
        inputs = [('X', X), ('noise_var', noise_var)]
        if mean is not None:
            inputs.append(('mean', mean))
        self._has_mean = mean is not None
        object.__setattr__(self, 'kernel', kernel)
        super(GPRegression, self).__init__(inputs=inputs, outputs=None, input_names=[k for k, _ in inputs],
                                           output_names=['random_variable'], rand_gen=rand_gen, dtype=dtype, ctx=ctx)

    def _generate_outputs(self, output_shapes):
        shape = output_shapes['random_variable']
        Y_shape = tuple(self.X.shape[:-1]) + (1,) if shape is None else shape
        self.set_outputs([Variable(shape=Y_shape)])

    def _build_module_graphs(self):
        Y = self.random_variable
        graph = ModuleGraph(name='gp_regression')
        graph.X = self.X
        graph.noise_var = self.noise_var
        if self._has_mean:
            graph.mean = self.mean
        graph.F = SimpleNamespace(factor=SimpleNamespace(has_mean=self._has_mean, dtype=self.dtype, kernel=self.kernel))
        graph.Y = Y
        graph.kernel = self.kernel
        for n, v in self.kernel.parameters.items():
            setattr(graph, n, v)
        post = ModuleGraph(name='gp_regression_posterior')      # stores what prediction needs (:355-359)
        post.L = Variable(shape=tuple(self.X.shape[:-1]) + tuple(self.X.shape[-2:-1]))
        post.LinvY = Variable(shape=tuple(self.X.shape[:-1]) + tuple(Y.shape[-1:]))
        post.X = Variable(shape=self.X.shape)
        for v in (post.L, post.LinvY, post.X):
            v.is_posterior_cache = True
        return graph, [post]

    def _attach_default_inference_algorithms(self):
        observed = [v for _, v in self.inputs] + [v for _, v in self.outputs]
        self.attach_log_pdf_algorithms(targets=self.output_names, conditionals=self.input_names,
                                       algorithm=GPRegressionLogPdf(self._module_graph, self._extra_graphs[0], observed),
                                       alg_name='gp_log_pdf')
        observed = [v for _, v in self.inputs]
        self.attach_draw_samples_algorithms(targets=self.output_names, conditionals=self.input_names,
                                            algorithm=GPRegressionSampling(self._module_graph, observed, rand_gen=self._rand_gen),
                                            alg_name='gp_sampling')
        self.attach_prediction_algorithms(targets=self.output_names, conditionals=self.input_names,
                                          algorithm=GPRegressionMeanVariancePrediction(self._module_graph, self._extra_graphs[0], observed),
                                          alg_name='gp_predict')

    @staticmethod
    def define_variable(X, kernel, noise_var, shape=None, mean=None, rand_gen=None, dtype=None, ctx=None):
        gp = GPRegression(X=X, kernel=kernel, noise_var=noise_var, mean=mean, rand_gen=rand_gen, dtype=dtype, ctx=ctx)
        gp._generate_outputs({'random_variable': shape})
        return gp.random_variable

    @property
    def random_variable(self):
        return self._outputs[0][1]
