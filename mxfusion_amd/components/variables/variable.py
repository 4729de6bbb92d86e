"""Variable (mxfusion/components/variables/variable.py:24-245): a UUID'd graph node that is a CONSTANT,
PARAMETER, RANDVAR (output of a Distribution / Module) or FUNCVAR (output of a function)."""
import uuid as _uuid
from enum import Enum

import numpy as np
import torch


class VariableType(Enum):
    CONSTANT = 0
    PARAMETER = 1
    RANDVAR = 2
    FUNCVAR = 3


class Variable(object):
    def __init__(self, value=None, shape=None, transformation=None, isInherited=False, initial_value=None):
        self.uuid = str(_uuid.uuid4()).replace('-', '_')
        self.name = None
        self.graph = None
        self.factor = None            # the factor whose output this variable is
        self.isInherited = isInherited
        self._transformation = transformation
        self._value = None
        self.isConstant = False
        self._initial_value = initial_value
        self.shape = None
        if isinstance(initial_value, (int, float)):
            self._initial_value = np.array([initial_value], dtype=np.float64) if shape in (None, (1,)) else initial_value
        if value is not None:
            self.set_prior(value)     # value may be a Distribution-produced Variable replacement or a constant
        if shape is None and isinstance(self._initial_value, (np.ndarray, torch.Tensor)):
            shape = tuple(self._initial_value.shape)
        self.shape = tuple(shape) if shape is not None else (1,)

    def set_prior(self, value):
        """variable.py:191-198: set the distribution this variable is drawn from (m.x.set_prior(Normal(...))); the factor and its input
        variables join the variable's graph.  (A plain number / array makes the variable a constant -- the constructor's `value=` path.)"""
        from ..factor import Factor
        if isinstance(value, Factor):
            self.assign_factor(value)
            if self.graph is not None:
                self.graph._register_factor(value)
        elif isinstance(value, (int, float, np.ndarray, torch.Tensor)):
            self.isConstant = True
            self._value = value
        else:
            raise TypeError('set_prior: a distribution (or a constant value) is expected, not %s' % type(value).__name__)

    @property
    def constant(self):
        return self._value

    @property
    def transformation(self):
        return self._transformation

    @property
    def initial_value(self):
        return self._initial_value

    @property
    def type(self):
        if self.isConstant:
            return VariableType.CONSTANT
        if self.factor is None:
            return VariableType.PARAMETER
        from ..distributions.distribution import Distribution
        from ...modules.module import Module
        if isinstance(self.factor, (Distribution, Module)):
            return VariableType.RANDVAR
        return VariableType.FUNCVAR

    def replicate_self(self):
        """variable.py:100-115: a copy that KEEPS the UUID (so the runtime `variables` dict entry is shared) but has no factor and no
        graph -- how a module's internal graph refers to the outer model's variables."""
        rep = Variable(shape=self.shape, transformation=self._transformation, isInherited=True, initial_value=self._initial_value)
        rep.uuid = self.uuid
        rep.name = self.name
        rep.isConstant = self.isConstant
        rep._value = self._value
        return rep

    def assign_factor(self, factor):
        """q[v].assign_factor(PointMass / Normal ...) (map.py:56-59, meanfield.py:40-43)."""
        factor.set_single_output(self)

    def __deepcopy__(self, memo):
        """(FactorGraph.clone) the copy is hashable -- it has its UUID -- before anything that may use it as a dictionary key is copied."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        object.__setattr__(new, 'uuid', self.uuid)
        for k, v in self.__dict__.items():
            if k != 'uuid':
                object.__setattr__(new, k, copy.deepcopy(v, memo))
        return new

    def __hash__(self):
        return hash(self.uuid)

    def __eq__(self, other):
        return isinstance(other, Variable) and other.uuid == self.uuid

    def __repr__(self):
        return 'Variable(%s, %s)' % (self.name, self.uuid[:8])
