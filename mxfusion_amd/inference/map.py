"""MAP (mxfusion/inference/map.py:24-84)."""
from .variational import VariationalInference
from ..components.variables.variable import Variable, VariableType
from ..components.distributions.pointmass import PointMass
from ..models.posterior import Posterior


class MAP(VariationalInference):
    def __init__(self, model, observed):
        posterior = MAP.create_posterior(model, observed)
        super(MAP, self).__init__(model=model, posterior=posterior, observed=observed)

    @staticmethod
    def create_posterior(model, observed):
        q = Posterior(model)
        for v in model.get_latent_variables(observed):
            q[v].assign_factor(PointMass(location=Variable(shape=v.shape)))
        return q

    def compute(self, F, variables):
        for v in self.model.variables.values():
            if v.type == VariableType.RANDVAR and v not in self._observed:
                variables[v.uuid] = variables[self.posterior[v].factor.location.uuid]
        logL = self.model.log_pdf(F=F, variables=variables)
        return -logL, -logL
