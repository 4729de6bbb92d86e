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
