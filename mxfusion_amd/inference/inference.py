"""Inference / TransferInference (mxfusion/inference/inference.py:31-365)."""
import warnings

import numpy as np
import torch

from ..util import serialization as ser
from ..common.exceptions import SerializationError

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
        """inference.py:62-83: the inference parameters as a string, one block per parameter --
        `<variable> in <Model | Posterior | FactorGraph>(<graph id>) : <value>` -- naming the first graph that holds the variable."""
        from ..models.model import Model
        from ..models.posterior import Posterior

        def kind(graph):
            return 'Model' if isinstance(graph, Model) else ('Posterior' if isinstance(graph, Posterior) else 'FactorGraph')
        out = ''
        for u, v in self.params._vars.items():
            if u not in self.params:
                continue
            owner = next((g for g in self._graphs if u in g), None)
            var = owner[u] if owner is not None else v
            tag = '%s(%s)' % (kind(owner), ('%x' % id(owner))[-5:]) if owner is not None else 'FactorGraph(?)'
            out += '{} in {} : {} \n\n'.format(var, tag, self.params[v])
        return out

    # ---- checkpoint (inference.py:179-310): the reference's zip layout, parameters keyed by UUID -------------------
    def _graph_listing(self):
        """graphs.json exactly as the reference lays it out (inference.py:283 -> FactorGraph.as_json, factor_graph.py:619-628): one
        networkx node-link dict per graph, every component encoded as serialization.py:42-53 does -- util/graph_json.py."""
        from ..util import graph_json
        return [graph_json.graph_as_json(g) for g in self._graphs]

    def get_serializable(self):
        return {'observed': self.observed_variable_UUIDs}

    def save(self, zip_filename=ser.DEFAULT_ZIP):
        """inference.py:255-310: version.json, graphs.json, mxnet_parameters.npz (one array per parameter UUID, the stored
        = unconstrained value, as gluon keeps it), mxnet_constants.npz, variable_constants.json, configuration.json."""
        params = self.params.export_raw()
        arr_consts, var_consts = {}, {}
        for u, c in self.params.constants.items():
            if isinstance(c, (int, float, np.integer, np.floating)):
                var_consts[u] = c.item() if hasattr(c, 'item') else c
            else:
                arr_consts[u] = c
        ser.write_zip(zip_filename,
                      {ser.FILENAMES['graphs']: self._graph_listing(), ser.FILENAMES['variable_constants']: var_consts,
                       ser.FILENAMES['configuration']: self.get_serializable(),
                       ser.FILENAMES['version_file']: {'serialization_version': ser.SERIALIZATION_VERSION}},
                      {ser.FILENAMES['mxnet_params']: params, ser.FILENAMES['mxnet_constants']: arr_consts})

    def _reconcile_saved_graphs(self, saved_graphs):
        """{saved uuid: current uuid}.  graphs.json in the reference's node-link layout (written by the reference itself or by save()
        above) goes through the reference's reconciliation (FactorGraph.reconcile_graphs, factor_graph.py:479-588, restated in
        util/graph_json.py); the positional listing round 1-2 checkpoints of this package carry is still understood."""
        from ..util import graph_json
        if graph_json.is_reference_graphs_json(saved_graphs):
            if len(saved_graphs) != len(self._graphs):
                raise SerializationError('saved inference has %d graphs, the current one %d' % (len(saved_graphs), len(self._graphs)))
            prev = graph_json.load_graphs(saved_graphs)
            return graph_json.reconcile_graphs(self._graphs, prev[0], prev[1:])
        current = [{'name': g.name, 'variables': [{'uuid': u, 'name': v.name, 'type': type(v).__name__,
                                                    'has_factor': v.factor is not None and type(v.factor).__name__}
                                                   for u, v in g.variables.items()]} for g in self._graphs]
        if len(saved_graphs) != len(current):
            raise SerializationError('saved inference has %d graphs, the current one %d' % (len(saved_graphs), len(current)))
        uuid_map = {}
        for sg, cg in zip(saved_graphs, current):
            if len(sg['variables']) != len(cg['variables']):
                raise SerializationError('graph %s: %d saved variables vs %d current' % (cg['name'], len(sg['variables']), len(cg['variables'])))
            for sv, cv in zip(sg['variables'], cg['variables']):
                if sv['type'] != cv['type'] or sv['has_factor'] != cv['has_factor'] or \
                        (sv['name'] != cv['name'] and sv['name'] != sv['uuid'] and cv['name'] != cv['uuid']):
                    raise SerializationError('graph %s: saved component %s (%s) does not match current %s (%s)'
                                             % (cg['name'], sv['name'], sv['type'], cv['name'], cv['type']))
                uuid_map[sv['uuid']] = cv['uuid']
        return uuid_map

    def load(self, zip_filename=ser.DEFAULT_ZIP):
        """inference.py:179-228: reconcile the saved graphs with the current ones ({saved uuid: current uuid} in
        self._uuid_map) and load the parameters / constants through that map.  Call after initialize().  Everything is validated BEFORE
        the first write: a checkpoint that cannot be restored completely leaves the inference untouched."""
        import zipfile
        ver = ser.load_json_from_zip(zip_filename, ser.FILENAMES['version_file'])
        if ver.get('serialization_version') != ser.SERIALIZATION_VERSION:
            raise SerializationError('Serialization version of saved inference and running code are not the same.')
        with zipfile.ZipFile(zip_filename, 'r') as zf:
            saved_params = ser.load_parameters(ser.FILENAMES['mxnet_params'], zf)
            saved_consts = ser.load_parameters(ser.FILENAMES['mxnet_constants'], zf)
        var_consts = ser.load_json_from_zip(zip_filename, ser.FILENAMES['variable_constants'])
        saved_graphs = ser.load_json_from_zip(zip_filename, ser.FILENAMES['graphs'])
        if not self._initialized:
            raise SerializationError('load() needs an initialised inference (call initialize(...) first, as the reference tests do)')
        uuid_map = self._reconcile_saved_graphs(saved_graphs)
        # ---- validate: every saved parameter has a counterpart of the same size, every trainable parameter is covered
        plan = []
        for su, arr in saved_params.items():
            cu = uuid_map.get(su)
            if cu is None:
                raise SerializationError('saved parameter %s has no counterpart in the current graphs' % su)
            raw = self.params._to_tensor(arr)
            if cu in self.params._slices:
                o, n, shape = self.params._slices[cu]
                if raw.numel() != n:
                    raise SerializationError('saved parameter %s has %d elements, the current one %d' % (su, raw.numel(), n))
            plan.append((cu, raw))
        # a checkpoint that silently leaves some trainable parameters at their random initial values is worse than an error
        restored = {cu for cu, _ in plan}
        missing = [u for u in self.params._slices if u not in restored]
        if missing:
            names = [getattr(self.params._vars.get(u), 'name', None) or u for u in missing]
            raise SerializationError('the checkpoint holds no value for %d trainable parameter(s): %s' % (len(missing), ', '.join(map(str, names[:8]))))
        # ---- write
        self._uuid_map = uuid_map
        for cu, raw in plan:
            if cu in self.params._slices:
                o, n, shape = self.params._slices[cu]
                with torch.no_grad():
                    self.params._flat[o:o + n] = raw.reshape(-1)
            else:
                self.params._fixed[cu] = raw
                if cu not in self.params._vars:
                    for g in self._graphs:
                        if cu in g.variables:
                            self.params._vars[cu] = g.variables[cu]
        consts = {}
        for su, arr in saved_consts.items():
            if su in uuid_map:
                consts[uuid_map[su]] = self.params._to_tensor(arr)
        for su, c in var_consts.items():
            if su in uuid_map:
                consts[uuid_map[su]] = c
        self.params.update_constants(consts)
        return self


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
