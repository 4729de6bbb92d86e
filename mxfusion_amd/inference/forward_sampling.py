"""ForwardSamplingAlgorithm (mxfusion/inference/forward_sampling.py)."""
from .inference_alg import SamplingAlgorithm


class ForwardSamplingAlgorithm(SamplingAlgorithm):
    def compute(self, F, variables):
        samples = self.model.draw_samples(F=F, variables=variables, num_samples=self.num_samples,
                                          targets=[t.uuid if hasattr(t, 'uuid') else t for t in self.target_variables]
                                          if self.target_variables else None)
        if self.target_variables:
            tv = [t.uuid if hasattr(t, 'uuid') else t for t in self.target_variables]
            return tuple(samples[u] for u in tv)
        return samples


class ForwardSampling(object):
    """forward_sampling.py:40-79: TransferInference over a ForwardSamplingAlgorithm (parameters inherited from a finished inference)."""

    def __new__(cls, num_samples, model, observed, var_tie, infr_params, target_variables=None, hybridize=False, constants=None,
                dtype=None, context=None):
        from ..components.variables.variable import Variable
        from .inference import TransferInference
        if target_variables is not None:
            target_variables = [v.uuid for v in target_variables if isinstance(v, Variable)]
        infr = ForwardSamplingAlgorithm(num_samples=num_samples, model=model, observed=observed, target_variables=target_variables)
        return TransferInference(inference_algorithm=infr, var_tie=var_tie, infr_params=infr_params, constants=constants,
                                 hybridize=hybridize, dtype=dtype, context=context)


class _PosteriorForwardSamplingAlgorithm(SamplingAlgorithm):
    """Forward sampling of the model with the prior of every latent variable replaced by its variational posterior
    (forward_sampling.py:100-115 merge_posterior_into_model + :24-56): the latent variables are drawn from q first and handed to the
    model's ancestral pass as given values -- the same joint as sampling the merged graph, without cloning and re-wiring graphs."""

    def __init__(self, model, posterior, observed, num_samples=1, target_variables=None):
        super(_PosteriorForwardSamplingAlgorithm, self).__init__(model=model, observed=observed, num_samples=num_samples,
                                                                 target_variables=target_variables, extra_graphs=[posterior])
        self._posterior = posterior

    def compute(self, F, variables):
        q_samples = self._posterior.draw_samples(F=F, variables=variables, num_samples=self.num_samples)
        for u, v in q_samples.items():
            variables[u] = v
        samples = dict(q_samples)
        samples.update(self.model.draw_samples(F=F, variables=variables, num_samples=self.num_samples))
        if self.target_variables:
            tv = [t.uuid if hasattr(t, 'uuid') else t for t in self.target_variables]
            return tuple(samples[u] for u in tv)
        return samples


class VariationalPosteriorForwardSampling(object):
    """forward_sampling.py:118-160: forward sampling for a finished variational (or MAP) inference -- latent variables from the learned
    posterior, everything downstream from the model."""

    def __new__(cls, num_samples, observed, inherited_inference, target_variables=None, hybridize=False, constants=None, dtype=None,
                context=None):
        from ..common.exceptions import InferenceError
        from ..components.variables.variable import Variable
        from .inference import TransferInference
        from .map import MAP
        from .variational import StochasticVariationalInference
        alg = inherited_inference.inference_algorithm
        if not isinstance(alg, (StochasticVariationalInference, MAP)):
            raise InferenceError('inherited_inference needs to be a subclass of SVIInference or SVIMiniBatchInference.')
        if target_variables is not None:
            target_variables = [v.uuid for v in target_variables if isinstance(v, Variable)]
        infr = _PosteriorForwardSamplingAlgorithm(model=alg.model, posterior=alg.posterior, observed=observed, num_samples=num_samples,
                                                  target_variables=target_variables)
        return TransferInference(inference_algorithm=infr, var_tie={}, infr_params=inherited_inference.params, constants=constants,
                                 hybridize=hybridize, dtype=dtype, context=context)
