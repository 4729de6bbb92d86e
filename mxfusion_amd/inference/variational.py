"""VariationalInference / StochasticVariationalInference (mxfusion/inference/variational.py:23-108)."""
from .inference_alg import InferenceAlgorithm


class VariationalInference(InferenceAlgorithm):
    def __init__(self, model, posterior, observed):
        super(VariationalInference, self).__init__(model=model, observed=observed, extra_graphs=[posterior])

    @property
    def posterior(self):
        return self._extra_graphs[0]


class StochasticVariationalInference(VariationalInference):
    """variational.py:60-108: ELBO by Monte-Carlo with reparameterised samples of the posterior."""

    def __init__(self, num_samples, model, posterior, observed):
        super(StochasticVariationalInference, self).__init__(model=model, posterior=posterior, observed=observed)
        self.num_samples = num_samples

    def compute(self, F, variables):
        samples = self.posterior.draw_samples(F=F, variables=variables, num_samples=self.num_samples)
        variables.update(samples)
        logL = self.model.log_pdf(F=F, variables=variables)
        logL = logL - self.posterior.log_pdf(F=F, variables=variables)
        return -logL, -logL
