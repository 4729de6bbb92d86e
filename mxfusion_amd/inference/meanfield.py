"""create_Gaussian_meanfield (mxfusion/inference/meanfield.py:24-44)."""
from ..components.variables.variable import Variable, VariableType
from ..components.variables.var_trans import PositiveTransformation
from ..components.distributions.normal import Normal
from ..models.posterior import Posterior
from ..util.inference import variables_to_UUID


def create_Gaussian_meanfield(model, observed, dtype=None):
    observed = variables_to_UUID(observed)
    q = Posterior(model)
    for v in list(model.variables.values()):
        if v.type == VariableType.RANDVAR and v.uuid not in observed:
            mean = Variable(shape=v.shape)
            variance = Variable(shape=v.shape, transformation=PositiveTransformation())
            q[v].set_prior(Normal(mean=mean, variance=variance, dtype=dtype))
    return q
