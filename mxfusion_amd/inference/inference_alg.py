"""InferenceAlgorithm + the executor that pre/post-processes one evaluation
(mxfusion/inference/inference_alg.py:25-293).  `compute(F, variables)` is the plug-in point; `F` is kept for
signature compatibility (the reference passes mx.nd) and carries the op namespace."""
import torch

from .. import ops as _F
from ..common.constants import SET_PARAMETER_PREFIX
from ..components.variables.runtime_variable import add_sample_dimension_to_arrays
from ..components.variables.variable import Variable, VariableType
from ..util.inference import variables_to_UUID, VariablesDict


class ObjectiveBlock(object):
    """inference_alg.py:25-90 (a Gluon HybridBlock there): var-ties, sample axis, positive transforms,
    compute(), then the SET_<uuid> parameter writes."""

    def __init__(self, infr_method, constants, data_def, var_trans, var_ties, infr_params):
        self._infr_method, self._constants, self._data_def = infr_method, constants, data_def
        self._var_trans, self._var_ties, self._infr_params = var_trans, var_ties, infr_params

    def __call__(self, *args):
        kw = dict(self._infr_params.tensors())            # uuid -> parameter view (autograd-connected to the flat buffer)
        for to_uuid, from_uuid in self._var_ties.items():
            kw[to_uuid] = kw[from_uuid]
        variables = VariablesDict()
        add_sample_dimension_to_arrays(_F, dict(zip(self._data_def, args)), out=variables)
        keys = [k for k in self._var_trans if k in kw]
        if keys:          # (positive parameters: one batched softplus call instead of one per parameter, var_trans.transform_many)
            from ..components.variables.var_trans import transform_many
            for k, y in zip(keys, transform_many([self._var_trans[k] for k in keys], [kw[k] for k in keys])):
                kw[k] = y
        add_sample_dimension_to_arrays(_F, kw, out=variables)
        add_sample_dimension_to_arrays(_F, self._constants, out=variables)
        obj = self._infr_method.compute(F=_F, variables=variables)
        with torch.no_grad():
            for k, v in list(variables.items()):
                if isinstance(k, str) and k.startswith(SET_PARAMETER_PREFIX):
                    self._infr_params[v[0]] = v[1]
        return obj


class InferenceAlgorithm(object):
    def __init__(self, model, observed, extra_graphs=None):
        self._model_graph = model
        self._extra_graphs = list(extra_graphs) if extra_graphs is not None else []
        self._graphs = [model] + self._extra_graphs
        self._observed = set(observed)
        self._observed_uuid = variables_to_UUID(observed)
        self._observed_names = [v.name for v in observed]

    @property
    def observed_variables(self):
        return self._observed

    @property
    def observed_variable_UUIDs(self):
        return self._observed_uuid

    @property
    def observed_variable_names(self):
        return self._observed_names

    @property
    def model(self):
        return self._model_graph

    @property
    def graphs(self):
        return self._graphs

    def prepare_executor(self, rv_scaling=None, global_weight=None):
        """inference_alg.py:165-190: collect the variable transformations; push rv_scaling into the factors.
        `global_weight` (row-sharded data-parallel loops; no reference counterpart): every factor whose variable is NOT in rv_scaling -- priors
        of global parameters, global latent variables, q(u) -- is evaluated in full by every rank and carries this weight (1 / world size), the
        factors in rv_scaling see only this rank's rows: the ranks' objectives then add up to the single-process objective."""
        from ..modules.module import Module
        var_trans = {}
        excluded = set()
        for g in self._graphs:
            for v in g.variables.values():
                if v.type == VariableType.PARAMETER and v.transformation is not None:
                    var_trans[v.uuid] = v.transformation
                if v.type == VariableType.RANDVAR and v.factor is not None and not isinstance(v.factor, Module):
                    f = v.factor
                    if rv_scaling is not None and v.uuid in rv_scaling:
                        f.log_pdf_scaling = rv_scaling[v.uuid]
                        f._mxf_global_weighted = False
                    elif global_weight is not None:
                        if not getattr(f, '_mxf_global_weighted', False):
                            f._mxf_unweighted_scaling = f.log_pdf_scaling
                        f.log_pdf_scaling = f._mxf_unweighted_scaling * global_weight
                        f._mxf_global_weighted = True
                    elif getattr(f, '_mxf_global_weighted', False):      # an earlier row-sharded run left its weight here
                        f.log_pdf_scaling = f._mxf_unweighted_scaling
                        f._mxf_global_weighted = False
            for f in getattr(g, '_factors', []):
                if isinstance(f, Module):
                    var_trans.update(f.prepare_executor(rv_scaling=rv_scaling, global_weight=global_weight))
        return var_trans, excluded

    def create_executor(self, data_def, params, var_ties, rv_scaling=None, global_weight=None):
        var_trans, _ = self.prepare_executor(rv_scaling=rv_scaling, global_weight=global_weight)
        return ObjectiveBlock(infr_method=self, constants=params.constants, data_def=data_def, var_trans=var_trans,
                              var_ties=var_ties, infr_params=params)

    def compute(self, F, variables):
        raise NotImplementedError

    def set_parameter(self, variables, target_variable, target_value):
        """inference_alg.py:236-251: request a direct parameter write after compute()."""
        variables[SET_PARAMETER_PREFIX + target_variable.uuid] = (target_variable, target_value)


class SamplingAlgorithm(InferenceAlgorithm):
    """inference_alg.py:254-293."""

    def __init__(self, model, observed, num_samples=1, target_variables=None, extra_graphs=None):
        super(SamplingAlgorithm, self).__init__(model=model, observed=observed, extra_graphs=extra_graphs)
        self.num_samples = num_samples
        self.target_variables = target_variables
