"""ExpectationAlgorithm (mxfusion/inference/expectation.py:24-60): forward samples of every variable, averaged over the sample axis
(components/variables/runtime_variable.py:101-118 `expectation`)."""
from ..components.variables.runtime_variable import expectation
from .inference_alg import SamplingAlgorithm


class ExpectationAlgorithm(SamplingAlgorithm):
    def compute(self, F, variables):
        samples = self.model.draw_samples(F=F, variables=variables, num_samples=self.num_samples)
        samples = {k: expectation(F, v) for k, v in samples.items()}
        if self.target_variables:
            return tuple(samples[getattr(v, 'uuid', v)] for v in self.target_variables)
        return samples
