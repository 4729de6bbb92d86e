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
