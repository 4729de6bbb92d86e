"""FactorGraph runtime (mxfusion/models/factor_graph.py:28-297): registration of named variables and the
topological walks log_pdf (:192-238) / draw_samples (:240-297); clone (:415-477) as a structural copy.  Graph reconciliation / JSON
(:479-643) lives in util/graph_json.py."""
from ..common.exceptions import ModelSpecificationError
from ..components.factor import Factor
from ..components.variables.variable import Variable, VariableType
from ..components.variables.runtime_variable import expectation


class FactorGraph(object):
    def __init__(self, name='graph', verbose=False):
        object.__setattr__(self, '_names', {})        # attribute name -> component
        object.__setattr__(self, '_variables', {})    # uuid -> Variable
        object.__setattr__(self, '_factors', [])      # registration order
        object.__setattr__(self, 'name', name)
        object.__setattr__(self, '_verbose', verbose)

    # ---- registration -------------------------------------------------------------------------
    def __setattr__(self, name, value):
        if isinstance(value, Variable):
            value.name = name if value.name is None else value.name
            self._names[name] = value
            self._register_variable(value)
            if self._verbose:
                print('Variable %s (%s) registered' % (name, value.uuid))
        elif isinstance(value, Factor):
            self._names[name] = value
            self._register_factor(value)
        object.__setattr__(self, name, value)

    def _register_variable(self, v):
        if v.uuid in self._variables and self._variables[v.uuid] is v:
            return
        self._variables[v.uuid] = v
        if v.graph is None:
            v.graph = self
        for s in (v.shape or ()):
            if isinstance(s, Variable):
                self._register_variable(s)
        if v.factor is not None:
            self._register_factor(v.factor)

    def _register_factor(self, f):
        if any(f is g for g in self._factors):
            return
        self._factors.append(f)
        f.graph = self if f.graph is None else f.graph
        for _, v in f.inputs:
            self._register_variable(v)
        for _, v in f.outputs:
            self._register_variable(v)
        for v in getattr(f, 'extra_parameters', lambda: [])():
            self._register_variable(v)

    def __repr__(self):
        """factor_graph.py:49-59: one line per factor in evaluation order -- `outputs = Function(...)` / `outputs ~ Distribution(...)`."""
        from ..components.functions.function_evaluation import FunctionEvaluation
        lines = ['%s (%s)' % (type(self).__name__, ('%x' % id(self))[-5:])]
        for f in self.ordered_factors:
            outs = ', '.join(str(v) for _, v in f.outputs)
            lines.append(outs + (' = ' if isinstance(f, FunctionEvaluation) else ' ~ ') + str(f))
        return '\n'.join(lines)

    # ---- the reference's serialisation / reconciliation entry points (factor_graph.py:479-643), implemented in util/graph_json.py ------
    def as_json(self):
        """factor_graph.py:619-628: the networkx node-link dictionary of this graph, laid out as the reference writes it."""
        from ..util import graph_json
        return graph_json.graph_as_json(self)

    @staticmethod
    def save(graph_file, json_graphs):
        """factor_graph.py:630-643."""
        import json
        with open(graph_file, 'w') as f:
            json.dump(json_graphs, f, ensure_ascii=False)

    @staticmethod
    def load_graphs(graphs_list, existing_graphs=None):
        """factor_graph.py:604-617: the saved graphs as read-only SavedGraph objects (uuid / name / type / edges -- what reconciliation
        needs; rebuilding live components from JSON is not supported: a model is re-created from its script and reconciled)."""
        from ..util import graph_json
        return graph_json.load_graphs(graphs_list if isinstance(graphs_list, list) else [graphs_list])

    @staticmethod
    def reconcile_graphs(current_graphs, primary_previous_graph, secondary_previous_graphs=None, primary_current_graph=None):
        """factor_graph.py:479-524: {uuid in the previous graphs: uuid in the current graphs}; the previous graphs may be live FactorGraphs
        or SavedGraph objects from load_graphs."""
        from ..util import graph_json
        as_saved = lambda g: graph_json.load_graphs([graph_json.graph_as_json(g)])[0] if isinstance(g, FactorGraph) else g
        cur = ([primary_current_graph] if primary_current_graph is not None else []) + list(current_graphs)
        return graph_json.reconcile_graphs(cur, as_saved(primary_previous_graph), [as_saved(g) for g in (secondary_previous_graphs or [])])

    def clone(self, leaves=None):
        """factor_graph.py:415-477: an independent copy of the graph -- same topology, same UUIDs and names, new component objects (modules
        with their internal graphs and attached algorithms, kernels with their parameter Variables included); array values (constants,
        initial values) are SHARED with the original, nothing is copied on the device.  `leaves`: accepted for signature compatibility --
        the whole graph is cloned."""
        import copy
        import torch
        memo = {}

        def share(o):          # arrays keep their identity in the copy
            memo[id(o)] = o
        for v in self._variables.values():
            for a in (getattr(v, '_value', None), getattr(v, '_initial_value', None)):
                if isinstance(a, torch.Tensor):
                    share(a)
        return copy.deepcopy(self, memo)

    def __getitem__(self, key):
        uuid = key.uuid if isinstance(key, Variable) else key
        return self._variables[uuid]

    def __contains__(self, key):
        uuid = key.uuid if isinstance(key, Variable) else key
        return uuid in self._variables

    @property
    def variables(self):
        return self._variables

    @property
    def components(self):
        return dict(self._names)

    # ---- structure ------------------------------------------------------------------------------
    @property
    def ordered_factors(self):
        """Topological order: a factor comes after the factors producing its inputs."""
        order, done = [], set()

        def visit(f):
            if id(f) in done:
                return
            done.add(id(f))
            for _, v in f.inputs:
                g = self._variables.get(v.uuid, v).factor
                if g is not None and any(g is h for h in self._factors):
                    visit(g)
            order.append(f)
        for f in self._factors:
            visit(f)
        return order

    def get_latent_variables(self, observed):
        obs = {(o.uuid if isinstance(o, Variable) else o) for o in observed}
        return [v for v in self._variables.values() if v.type == VariableType.RANDVAR and v.uuid not in obs]

    def get_parameters(self, excluded=None, include_inherited=True):
        excluded = excluded or set()
        out = []
        for v in self._variables.values():
            if v.type == VariableType.PARAMETER and v.uuid not in excluded and (include_inherited or not v.isInherited):
                out.append(v)
        return out

    def get_constants(self):
        return [v for v in self._variables.values() if v.type == VariableType.CONSTANT]

    # ---- runtime ---------------------------------------------------------------------------------
    def log_pdf(self, F, variables, targets=None):
        """factor_graph.py:192-238: logL = sum over factors of sum(mean_S(log_pdf))."""
        from ..components.distributions.distribution import Distribution
        from ..components.functions.function_evaluation import FunctionEvaluation
        from ..modules.module import Module
        if targets is not None:
            targets = {t.uuid if isinstance(t, Variable) else t for t in targets}
        logL = 0.
        for f in self.ordered_factors:
            if isinstance(f, FunctionEvaluation):
                outcome = f.eval(F=F, variables=variables, always_return_tuple=True)
                for v, (_, ov) in zip(outcome, f.outputs):
                    variables[ov.uuid] = v
            elif isinstance(f, Distribution):
                if targets is None or f.random_variable.uuid in targets:
                    if hasattr(f, 'log_pdf_sum'):
                        logL = logL + f.log_pdf_sum(F, variables)          # fused sum(mean_S(.)) kernel
                    else:
                        logL = logL + expectation(F, f.log_pdf(F=F, variables=variables)).sum()
            elif isinstance(f, Module):
                if targets is None:
                    module_targets = [v.uuid for _, v in f.outputs if v.uuid in variables]
                else:
                    module_targets = [v.uuid for _, v in f.outputs if v.uuid in targets]
                if len(module_targets) > 0:
                    logL = logL + expectation(F, f.log_pdf(F=F, variables=variables, targets=module_targets)).sum()
            else:
                raise ModelSpecificationError("There is an object in the factor graph that isn't a factor.")
        return logL

    def draw_samples(self, F, variables, num_samples=1, targets=None):
        """factor_graph.py:240-297."""
        from ..components.distributions.distribution import Distribution
        from ..components.functions.function_evaluation import FunctionEvaluation
        from ..modules.module import Module
        samples = {}
        for f in self.ordered_factors:
            if isinstance(f, FunctionEvaluation):
                outcome = f.eval(F=F, variables=variables, always_return_tuple=True)
                for v, (_, ov) in zip(outcome, f.outputs):
                    variables[ov.uuid] = v
                    samples[ov.uuid] = v
            elif isinstance(f, Distribution):
                known = [v.uuid in variables for _, v in f.outputs]
                if all(known):
                    continue
                outcome = f.draw_samples(F=F, variables=variables, num_samples=num_samples, always_return_tuple=True)
                for v, (_, ov) in zip(outcome, f.outputs):
                    variables[ov.uuid] = v
                    samples[ov.uuid] = v
            elif isinstance(f, Module):
                outcome_uuid = [v.uuid for _, v in f.outputs]
                if all(u in variables for u in outcome_uuid):
                    continue
                outcome = f.draw_samples(F=F, variables=variables, num_samples=num_samples, targets=outcome_uuid)
                for v, u in zip(outcome, outcome_uuid):
                    variables[u] = v
                    samples[u] = v
            else:
                raise ModelSpecificationError("There is an object in the factor graph that isn't a factor.")
        if targets:
            targets = [t.uuid if isinstance(t, Variable) else t for t in targets]
            return {u: samples[u] for u in targets if u in samples}
        return samples
