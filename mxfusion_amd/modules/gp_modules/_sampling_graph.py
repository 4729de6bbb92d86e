"""The generative internal graph of the two sparse GP modules, for forward sampling:
    U ~ GP(Z; kernel),  F ~ GP(X | Z, U; kernel) (+ mean),  Y ~ N(F, noise_var)
(svgp_regression.py:349-374 and sparsegp_regression.py:323-347 build exactly this as the module graph; the default `svgp_sampling` /
`sgp_sampling` algorithm is a ForwardSamplingAlgorithm over it, :399-403 / :374-378).  The log-pdf and prediction algorithms of this
build read the module's variables through the ModuleGraph namespace; this Model holds REPLICAS of them (same UUIDs, so the runtime
`variables` dict is shared -- SURVEY A.9) wired to the three factors."""
from ...components.distributions.gp.cond_gp import ConditionalGaussianProcess
from ...components.distributions.gp.gp import GaussianProcess
from ...components.distributions.normal import Normal
from ...models.model import Model


def build_sparse_gp_sampling_model(module, name):
    Y = module.random_variable
    g = Model(name=name)
    g.X = module.X.replicate_self()
    g.inducing_inputs = module.inducing_inputs.replicate_self()
    g.noise_var = module.noise_var.replicate_self()
    M = module.inducing_inputs.shape[0]
    kw = dict(rand_gen=module._rand_gen, dtype=module.dtype, ctx=module.ctx)
    g.U = GaussianProcess.define_variable(X=g.inducing_inputs, kernel=module.kernel, shape=(M, Y.shape[-1]), **kw)
    mean = None
    if module._has_mean:
        mean = module.mean.replicate_self()
        g.mean = mean
    g.F = ConditionalGaussianProcess.define_variable(X=g.X, X_cond=g.inducing_inputs, Y_cond=g.U, kernel=module.kernel, shape=Y.shape,
                                                     mean=mean, **kw)
    y = Y.replicate_self()
    Normal(mean=g.F, variance=g.noise_var, **kw).set_single_output(y)      # variance broadcasts to Y.shape (broadcast_to, :370)
    g.Y = y
    g.kernel = module.kernel
    return g
