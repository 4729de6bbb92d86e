"""Inference / TransferInference (mxfusion/inference/inference.py:31-365)."""
import warnings

import numpy as np
import torch

from ..common import config
from ..common.exceptions import InferenceError
from ..components.variables.variable import Variable
from .inference_parameters import InferenceParameters


def discover_shape_constants(data_shapes, graphs):
    """util/inference.py:62-87: bind symbolic dimensions (m.N = Variable()) from the observed shapes."""
    consts = {}
    for g in graphs:
        for uuid, shape in data_shapes.items():
            if uuid not in g.variables:
                continue
            vshape = g.variables[uuid].shape
            for s, d in zip(vshape, shape):
                if isinstance(s, Variable):
                    consts[s.uuid] = int(d)
    return consts


class Inference(object):
    def __init__(self, inference_algorithm, constants=None, hybridize=False, dtype=None, context=None):
        self.dtype = dtype if dtype is not None else config.DEFAULT_DTYPE
        self.mxnet_context = context if context is not None else config.get_default_device()
        self._graphs = inference_algorithm.graphs
        self._inference_algorithm = inference_algorithm
        self.params = InferenceParameters(constants=constants, dtype=self.dtype, context=self.mxnet_context)
        self._initialized = False

    @property
    def observed_variables(self):
        return self._inference_algorithm.observed_variables

    @property
    def observed_variable_UUIDs(self):
        return self._inference_algorithm.observed_variable_UUIDs

    @property
    def observed_variable_names(self):
        return self._inference_algorithm.observed_variable_names

    @property
    def graphs(self):
        return self._graphs

    @property
    def inference_algorithm(self):
        return self._inference_algorithm

    def create_executor(self):
        return self._inference_algorithm.create_executor(data_def=self.observed_variable_UUIDs, params=self.params,
                                                         var_ties=self.params.var_ties)

    def _initialize_params(self):
        self.params.initialize_params(self._graphs, self.observed_variable_UUIDs)

    def initialize(self, **kw):
        """inference.py:126-156."""
        if self._initialized:
            return
        data = [kw[v] for v in self.observed_variable_names]
        if len(data) > 0:
            if isinstance(data[0], (tuple, list)):
                data_shapes = {i: tuple(d) for i, d in zip(self.observed_variable_UUIDs, data)}
            elif isinstance(data[0], (torch.Tensor, np.ndarray)):
                data_shapes = {i: tuple(d.shape) for i, d in zip(self.observed_variable_UUIDs, data)}
            else:
                raise InferenceError('Keywords must be device tensors or shapes')
            self.params.update_constants(discover_shape_constants(data_shapes, self._graphs))
        self._initialize_params()
        self._initialized = True

    def _to_device(self, d):
        if isinstance(d, np.ndarray):
            d = torch.as_tensor(d)
        return d.to(device=self.mxnet_context, dtype=config.torch_dtype(self.dtype))

    def run(self, **kwargs):
        """inference.py:158-171."""
        data = [self._to_device(kwargs[v]) for v in self.observed_variable_names]
        self.initialize(**kwargs)
        executor = self.create_executor()
        with torch.no_grad():
            return executor(*data)

    def print_params(self):
        for u, v in self.params._vars.items():
            if u in self.params:
                print(v.name, u, self.params[v])


class TransferInference(Inference):
    """inference.py:312-365: a new inference whose parameters are inherited (by UUID) from a finished one."""

    def __init__(self, inference_algorithm, infr_params, var_tie=None, constants=None, hybridize=False, dtype=None, context=None):
        self._var_tie = var_tie if var_tie is not None else {}
        self._inherited_params = infr_params
        if dtype is None:
            dtype = infr_params.dtype
        super(TransferInference, self).__init__(inference_algorithm=inference_algorithm, constants=constants, hybridize=hybridize,
                                                dtype=dtype, context=context)

    def _initialize_params(self):
        carry = self._inherited_params.export_raw()
        self.params.initialize_params(self._graphs, self.observed_variable_UUIDs, carry=carry)
