"""Posterior (mxfusion/models/posterior.py): a FactorGraph over replicas of the model's variables (same
UUIDs, no factors) to which variational factors are assigned: q[v].assign_factor(...) / q[v].set_prior(...)."""
from .factor_graph import FactorGraph
from ..components.variables.variable import Variable


class _PosteriorVariable(Variable):
    """Replica of a model variable: keeps the UUID (variable.py replicate_self), drops the factor."""

    def __init__(self, src, graph):
        super(_PosteriorVariable, self).__init__(shape=src.shape, transformation=src.transformation)
        self.uuid = src.uuid
        self.name = src.name
        self.graph = graph
        self.isConstant = src.isConstant
        self._value = src._value
        self.isInherited = True
        self._src = src
        self._has_q = False

    def assign_factor(self, factor):
        factor.set_single_output(self)
        self._has_q = True
        self.graph._register_factor(factor)

    def set_prior(self, factor):
        from ..components.distributions.distribution import Distribution
        if isinstance(factor, Distribution):
            self.assign_factor(factor)
        else:
            super(_PosteriorVariable, self).set_prior(factor)

    @property
    def type(self):
        if self.factor is None and not self.isConstant:
            return self._src.type     # untouched replica: same role as in the model
        return super(_PosteriorVariable, self).type


class Posterior(FactorGraph):
    def __init__(self, model, name='posterior', verbose=False):
        super(Posterior, self).__init__(name=name, verbose=verbose)
        object.__setattr__(self, '_model', model)
        for uuid, v in model.variables.items():
            self._variables[uuid] = _PosteriorVariable(v, self)

    def get_parameters(self, excluded=None, include_inherited=False):
        """Only the variational parameters introduced by the posterior factors (replicas belong to the model)."""
        excluded = excluded or set()
        out = []
        for f in self._factors:
            for _, v in f.inputs:
                if v.factor is None and not v.isConstant and v.uuid not in excluded and not isinstance(v, _PosteriorVariable):
                    if all(v is not o for o in out):
                        out.append(v)
        return out

    def get_latent_variables(self, observed):
        return [v for v in self._variables.values() if isinstance(v, _PosteriorVariable) and v._has_q]
