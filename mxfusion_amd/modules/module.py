"""Module(Factor) (mxfusion/modules/module.py:27-479): a probabilistic module owns an internal model graph,
extra (posterior) graphs with hidden parameters, and a registry
    {conditionals -> [(targets, algorithm, alg_name)]}
for log_pdf / draw_samples / predict.  This registry is the reference's plug-in point for the hot path
(module.py:193-237): the MI355X algorithms are attached through it exactly like the reference's own."""
from types import SimpleNamespace

from ..common.exceptions import ModelSpecificationError
from ..components.factor import Factor
from ..components.variables.variable import Variable, VariableType


class ModuleGraph(SimpleNamespace):
    """Internal graph of a module: a namespace of the module's variables (same objects / UUIDs as the outer
    model -- the reference replicates them keeping the UUID)."""

    def __getitem__(self, key):
        for v in vars(self).values():
            if isinstance(v, Variable) and v.uuid == (key.uuid if isinstance(key, Variable) else key):
                return v
        raise KeyError(key)

    @property
    def variables(self):
        return {v.uuid: v for v in vars(self).values() if isinstance(v, Variable)}

    def get_parameters(self):
        return [v for v in self.variables.values() if v.type == VariableType.PARAMETER]


class Module(Factor):
    def __init__(self, inputs, outputs, input_names, output_names, rand_gen=None, dtype=None, ctx=None):
        super(Module, self).__init__(inputs, outputs, input_names, output_names)
        from ..components.distributions.random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen
        self.dtype = dtype
        self.ctx = ctx
        self._module_graph = None
        self._extra_graphs = []
        self._log_pdf_algorithms = {}
        self._draw_samples_algorithms = {}
        self._prediction_algorithms = {}
        self.log_pdf_scaling = 1

    # ---- construction ------------------------------------------------------------------------------
    def set_outputs(self, variables):
        variables = [variables] if not isinstance(variables, (list, tuple)) else variables
        self._outputs = list(zip(self._output_names, variables))
        for v in variables:
            v.factor = self
        self._module_graph, self._extra_graphs = self._build_module_graphs()
        self._attach_default_inference_algorithms()

    def _build_module_graphs(self):
        raise NotImplementedError

    def _attach_default_inference_algorithms(self):
        raise NotImplementedError

    def extra_parameters(self):
        """Hidden parameters (module.py:136-147): parameters of the internal graphs that are not module inputs."""
        ins = {v.uuid for _, v in self.inputs} | {v.uuid for _, v in self.outputs}
        out = []
        for g in [self._module_graph] + list(self._extra_graphs):
            if g is None:
                continue
            for v in g.get_parameters():
                if v.uuid not in ins and all(v is not o for o in out):
                    out.append(v)
        return out

    @property
    def hidden_parameters(self):
        return [v.uuid for v in self.extra_parameters()]

    # True for modules whose log-pdf is a SUM over data rows plus row-independent terms (SVGP: svgp_regression.py:98-109): only those can be
    # evaluated on a shard of the rows by each rank of a data-parallel loop (second axis)
    row_additive = False

    def prepare_executor(self, rv_scaling=None, global_weight=None):
        """module.py:393-418: transformations of the hidden parameters; rv_scaling -> log_pdf_scaling.
        `global_weight` (row-sharded data-parallel loops, no reference counterpart): the weight of the terms that do NOT depend on the data rows
        (the module's KL term; the whole module when its outputs are not row-sharded), so that the ranks' objectives add up to the
        single-process objective."""
        var_trans = {v.uuid: v.transformation for v in self.extra_parameters() if v.transformation is not None}
        sharded = False
        if rv_scaling is not None:
            for _, v in self.outputs:
                if v.uuid in rv_scaling:
                    self.log_pdf_scaling = rv_scaling[v.uuid]
                    sharded = True
        if global_weight is not None and sharded and not self.row_additive:
            from ..common.exceptions import InferenceError
            raise InferenceError('%s: the log-pdf of this module is not a sum over data rows -- it cannot be sharded by rows '
                                 "(use shard='samples', or replicas)" % type(self).__name__)
        self.global_weight = 1.0 if global_weight is None else float(global_weight)
        self._rows_sharded = sharded
        return var_trans

    # ---- registry -----------------------------------------------------------------------------------
    def get_names_from_uuid(self, uuids):
        u2n = {v.uuid: k for k, v in self.inputs}
        u2n.update({v.uuid: k for k, v in self.outputs})
        return tuple(sorted(u2n[u] for u in uuids if u in u2n))

    def _attach_algorithm(self, registry, targets, conditionals, algorithm, alg_name):
        """module.py:239-302: re-attaching under the same (targets, conditionals) replaces the entry."""
        if targets is not None:
            targets = tuple(sorted(targets))
        conditionals = tuple(sorted(conditionals)) if conditionals is not None else ()
        lst = registry.setdefault(conditionals, [])
        for i, (t, _, n) in enumerate(lst):
            if t == targets:
                if n is not None and hasattr(self, n):
                    try:
                        object.__delattr__(self, n)
                    except AttributeError:
                        pass
                lst[i] = (targets, algorithm, alg_name)
                break
        else:
            lst.append((targets, algorithm, alg_name))
        if alg_name is not None:
            object.__setattr__(self, alg_name, algorithm)

    def attach_log_pdf_algorithms(self, targets, conditionals, algorithm, alg_name=None):
        self._attach_algorithm(self._log_pdf_algorithms, targets, conditionals, algorithm, alg_name)

    def attach_draw_samples_algorithms(self, targets, conditionals, algorithm, alg_name=None):
        self._attach_algorithm(self._draw_samples_algorithms, targets, conditionals, algorithm, alg_name)

    def attach_prediction_algorithms(self, targets, conditionals, algorithm, alg_name=None):
        self._attach_algorithm(self._prediction_algorithms, targets, conditionals, algorithm, alg_name)

    def _get_algorithm_for_target_conditional_pair(self, algorithms, targets, conditionals, exact_match=False):
        """module.py:366-391."""
        if conditionals not in algorithms:
            raise ModelSpecificationError('The module %s has no algorithm for conditionals %s' % (type(self).__name__, conditionals))
        alg = None
        for t, a, _ in algorithms[conditionals]:
            if (exact_match and t == targets) or (not exact_match and (t is None or set(targets) <= set(t))):
                alg = a
                break
        if alg is None:
            raise ModelSpecificationError('The module %s has no algorithm for targets %s given %s' % (type(self).__name__, targets, conditionals))
        return alg

    # ---- runtime ----------------------------------------------------------------------------------------
    @staticmethod
    def _uuids(targets):
        """Targets may be given as Variables (pilco_alg.py:80 passes [model.Y]; a reference Variable hashes and compares as its UUID)."""
        return None if targets is None else [getattr(t, 'uuid', t) for t in targets]

    def _names(self, variables, targets):
        targets = self._uuids(targets)
        if targets is None:
            target_names = tuple(sorted(self.output_names))
        else:
            target_names = self.get_names_from_uuid(targets)
        conditionals_names = self.get_names_from_uuid([v.uuid for _, v in self.inputs if v.uuid in variables])
        return target_names, conditionals_names

    def log_pdf(self, F, variables, targets=None):
        """module.py:304-322."""
        target_names, conditionals_names = self._names(variables, targets)
        alg = self._get_algorithm_for_target_conditional_pair(self._log_pdf_algorithms, target_names, conditionals_names, exact_match=True)
        alg.log_pdf_scaling = self.log_pdf_scaling
        gw = getattr(self, 'global_weight', 1.0)
        if gw != 1.0:
            if getattr(self, '_rows_sharded', False):
                alg.kl_weight = gw           # row-sharded: the data term is this rank's share already, the row-independent terms carry 1 / world
                try:
                    return alg.compute(F, variables)
                finally:
                    alg.kl_weight = 1.0
            return gw * alg.compute(F, variables)       # a module every rank evaluates in full
        return alg.compute(F, variables)

    def draw_samples(self, F, variables, num_samples=1, targets=None):
        """module.py:324-344."""
        target_names, conditionals_names = self._names(variables, targets)
        alg = self._get_algorithm_for_target_conditional_pair(self._draw_samples_algorithms, target_names, conditionals_names)
        alg.num_samples = num_samples
        alg.target_variables = self._uuids(targets)
        return alg.compute(F, variables)

    def predict(self, F, variables, num_samples=1, targets=None):
        """module.py:346-364."""
        target_names, conditionals_names = self._names(variables, targets)
        alg = self._get_algorithm_for_target_conditional_pair(self._prediction_algorithms, target_names, conditionals_names, exact_match=True)
        alg.num_samples = num_samples
        alg.target_variables = self._uuids(targets)
        return alg.compute(F, variables)
