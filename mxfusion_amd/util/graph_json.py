"""graphs.json of a checkpoint in the REFERENCE's format, both directions, and the UUID reconciliation of a saved set of graphs with the
current ones.

What the reference writes (mxfusion/inference/inference.py:255-310 -> FactorGraph.as_json, models/factor_graph.py:619-628): a list with
one networkx `node_link_data` dict per graph of the inference (the model first, then the posterior / extra graphs),

    {"directed": true, "multigraph": true, "graph": {}, "name": <graph name>,
     "nodes": [{"id": <component>}, ...],
     "links": [{"name": <edge name>, "key": <edge name>, "source": <component>, "target": <component>}, ...]}

with every ModelComponent encoded by util/serialization.py:42-53 (ModelComponentEncoder) from its `as_json()`:

    {"uuid": ..., "name": <attribute name in its graph | null>, "attributes": [uuid of attribute variables, e.g. shape variables],
     "version": "1.0", "type": <class name>}                                   components/model_component.py:62-65
    + "inherited_name": <name | null>                     for Variables        components/variables/variable.py:99-102
    + "graphs": [<module graph>, <extra graphs> ...]      for Modules          modules/module.py:475-479

Edges run predecessor -> successor and carry the name the successor knows the predecessor by (a factor's input name, e.g. "mean",
"rbf_lengthscale"; for a factor -> variable edge the factor's output name, "random_variable"): components/model_component.py:130-199.

Reading it back the reference rebuilds bare ModelComponents (serialization.py:56-83) and matches them to the graphs of the running script
(FactorGraph.reconcile_graphs, models/factor_graph.py:479-588): components with the same NAME first, then breadth-first over the
predecessors of matched components, pairing neighbours by EDGE NAME; a Module pairs its internal graphs recursively
(modules/module.py:435-444).  `reconcile_graphs` below restates that algorithm over plain dictionaries, so that a checkpoint written by
the reference loads here, and `graph_as_json` emits the same layout for the graphs of this package, so that the reference's loader finds
what it expects.  The modules of this package keep their internal graphs as flat namespaces of the variables they share with the outer
model (same UUIDs); their JSON therefore lists those variables as nodes without the internal factors -- enough for the
name-based step of the reference's reconciliation, which is all a module's internal graphs need (every parameter-carrying variable
inside a module is either named there or an input of the module itself)."""
from ..common.exceptions import SerializationError

GRAPH_JSON_VERSION = '1.0'


# ------------------------------------------------------------------------------------------------------------ current graphs -> views
class _View(object):
    """One graph of the running script as the reconciliation sees it: components by uuid, named components, predecessors with edge names."""

    def __init__(self, graph):
        from ..components.variables.variable import Variable
        from ..components.factor import Factor
        self.name = getattr(graph, 'name', None) or type(graph).__name__
        self.components, self.named, self._pred = {}, {}, {}
        variables = dict(getattr(graph, 'variables', {}))
        factors = list(getattr(graph, '_factors', []))
        for u, v in variables.items():
            self.components[u] = v
        for f in factors:
            self.components[f.uuid] = f
        names = getattr(graph, 'components', None)
        if isinstance(names, dict):
            self.named = {n: c for n, c in names.items() if isinstance(c, (Variable, Factor))}
        else:                                              # a module's flat namespace: attribute name = component name
            self.named = {n: c for n, c in vars(graph).items() if isinstance(c, (Variable, Factor))}
        for n, c in self.named.items():
            self.components.setdefault(c.uuid, c)
        in_graph = {id(f) for f in factors}
        for u, v in variables.items():
            f = getattr(v, 'factor', None)
            if f is not None and id(f) in in_graph:
                self._pred[u] = [(n, f) for n, o in f.outputs if o.uuid == u]
        for f in factors:
            self._pred[f.uuid] = factor_predecessors(f)

    def predecessors(self, uuid):
        return self._pred.get(uuid, [])


def factor_predecessors(f):
    """[(edge name, variable)] of a factor as the reference wires it: its inputs, and for the GP modules / distributions the kernel's
    parameters under their prefixed names (`<kernel>_<parameter>`, kernels/kernel.py:232-245; the reference passes them to the factor as
    inputs, modules/gp_modules/gp_regression.py:317-331), plus a mean function's parameters."""
    out = list(f.inputs)
    have = {n for n, _ in out}
    kern = f.__dict__.get('kernel', None)
    if kern is not None and hasattr(kern, 'parameters'):
        for n, v in kern.parameters.items():
            if n not in have:
                out.append((n, v))
                have.add(n)
    return out


def _module_graphs(c):
    mg = c.__dict__.get('_module_graph', None) if hasattr(c, '__dict__') else None
    if mg is None:
        return None
    return [mg] + list(c.__dict__.get('_extra_graphs', []))


# ------------------------------------------------------------------------------------------------------------ writing
def component_json(c, name=None):
    """`name`: the attribute name of the component in the graph being written (FactorGraph.__setattr__ gives a component its name,
    factor_graph.py:71-87; the flat module graphs of this package do not rename the shared variables)."""
    from ..components.variables.variable import Variable
    d = {'uuid': c.uuid, 'name': name if name is not None else getattr(c, 'name', None),
         'attributes': [a.uuid for a in (getattr(c, 'shape', None) or ()) if isinstance(a, Variable)] if isinstance(c, Variable) else []}
    if isinstance(c, Variable):
        d['inherited_name'] = getattr(c, 'inherited_name', None) if getattr(c, 'isInherited', False) else None
    graphs = _module_graphs(c)
    if graphs is not None:
        d['graphs'] = [graph_as_json(g, module=c if i == 0 else None) for i, g in enumerate(graphs)]
    d['version'] = GRAPH_JSON_VERSION
    d['type'] = type(c).__name__
    return d


def graph_as_json(graph, module=None):
    """FactorGraph.as_json (factor_graph.py:619-628) for a graph of this package.  `module`: the graph is an internal graph of that module.
    The reference keeps the kernel's parameters there as UNNAMED variables feeding its internal GaussianProcess factor under their prefixed
    names, with the latent function variable `F` as that factor's output (gp_regression.py:346-349); this package's module graphs are flat
    namespaces, so that little structure is written out explicitly (the factor and F get uuids derived from the module's) -- the
    reference's loader pairs the kernel parameters by walking F <- GaussianProcess <- parameters."""
    view = _View(graph)
    names = {c.uuid: n for n, c in view.named.items()}
    kpar = {}
    if module is not None:
        kern = module.__dict__.get('kernel', None)
        kpar = dict(kern.parameters) if kern is not None and hasattr(kern, 'parameters') else {}
        for n, v in kpar.items():
            if names.get(v.uuid) == n:
                names.pop(v.uuid)                                   # unnamed in the reference's module graph
    enc = {u: component_json(c, names.get(u)) for u, c in view.components.items()}
    links = []
    for u in view.components:
        for name, p in view.predecessors(u):
            if p.uuid not in enc:
                enc[p.uuid] = component_json(p, names.get(p.uuid))
            links.append({'name': name, 'key': name, 'source': enc[p.uuid], 'target': enc[u]})
    if kpar and all(v.uuid in enc for v in kpar.values()):
        gp = {'uuid': module.uuid + '_gp', 'name': None, 'attributes': [], 'version': GRAPH_JSON_VERSION, 'type': 'GaussianProcess'}
        fv = {'uuid': module.uuid + '_F', 'name': 'F', 'attributes': [], 'inherited_name': None, 'version': GRAPH_JSON_VERSION, 'type': 'Variable'}
        enc[gp['uuid']], enc[fv['uuid']] = gp, fv
        for n, v in kpar.items():
            links.append({'name': n, 'key': n, 'source': enc[v.uuid], 'target': gp})
        links.append({'name': 'random_variable', 'key': 'random_variable', 'source': gp, 'target': fv})
    return {'directed': True, 'multigraph': True, 'graph': {}, 'nodes': [{'id': e} for e in enc.values()], 'links': links,
            'name': view.name}


# ------------------------------------------------------------------------------------------------------------ reading
class SavedComponent(object):
    """serialization.py:62-83: a bare component rebuilt from its JSON."""

    def __init__(self, obj):
        if obj.get('version') != GRAPH_JSON_VERSION:
            raise SerializationError('The format of the stored model component %s is from version %s; the current version is %s.'
                                     % (obj.get('name'), obj.get('version'), GRAPH_JSON_VERSION))
        self.uuid, self.name, self.type = obj['uuid'], obj.get('name'), obj.get('type')
        self.attributes = list(obj.get('attributes', []))
        self.inherited_name = obj.get('inherited_name')
        self.graphs = [SavedGraph(g) for g in obj['graphs']] if 'graphs' in obj else None      # a Module (serialization.py:71-73)


class SavedGraph(object):
    """FactorGraph.load_from_json (factor_graph.py:590-602) without networkx: node-link dict -> components, named components, predecessors."""

    def __init__(self, js):
        if not isinstance(js, dict) or 'nodes' not in js:
            raise SerializationError('graphs.json: not a node-link graph')
        self.name = js.get('name')
        self.components = {}
        for n in js['nodes']:
            c = SavedComponent(n['id'])
            self.components[c.uuid] = c
        self._pred = {}
        for e in js.get('links', js.get('edges', [])):
            s, t = e['source'], e['target']
            su = s['uuid'] if isinstance(s, dict) else s
            tu = t['uuid'] if isinstance(t, dict) else t
            for side in (s, t):
                if isinstance(side, dict) and side['uuid'] not in self.components:
                    self.components[side['uuid']] = SavedComponent(side)
            if su not in self.components or tu not in self.components:
                raise SerializationError('graphs.json: an edge of graph %s names an unknown component' % self.name)
            self._pred.setdefault(tu, []).append((e.get('name', e.get('key')), self.components[su]))
        self.named = {c.name: c for c in self.components.values() if c.name}

    def predecessors(self, uuid):
        return self._pred.get(uuid, [])


def is_reference_graphs_json(obj):
    return isinstance(obj, list) and len(obj) > 0 and all(isinstance(g, dict) and 'nodes' in g for g in obj)


def load_graphs(graphs_list):
    """FactorGraph.load_graphs (factor_graph.py:604-617)."""
    return [SavedGraph(g) for g in graphs_list]


# ------------------------------------------------------------------------------------------------------------ reconciliation
def _reconcile_level(traverse, cmap, cur, prev, strict=True):
    """FactorGraph._reconcile_graph (factor_graph.py:526-588): breadth-first over the predecessors of already matched components.
    `strict` = False (secondary graphs): the reference's posterior graphs are CLONES of the model graph (factor_graph.py:325-391: variables
    keep their uuid, factors are replicated), this package's posteriors hold only what they add -- a saved predecessor edge without a
    current counterpart there belongs to the cloned part, whose variables are paired through the model graph already."""
    while traverse:
        new_level = {}
        for pu, cu in traverse.items():
            if pu not in prev.components or cu not in cur.components:
                continue
            prev_n, cur_n = prev.predecessors(pu), cur.predecessors(cu)
            names = [n for n, _ in prev_n]
            dup = {n for n in names if names.count(n) > 1}
            for edge_name, node in prev_n:
                if node.uuid in cmap:
                    continue
                if edge_name in dup:
                    raise SerializationError("Multiple edges connecting unnamed nodes have the same name (%s), this isn't supported." % edge_name)
                match = [c for n, c in cur_n if n == edge_name]
                if not match and not strict:
                    continue
                if not match:
                    raise SerializationError('the saved graph %s has an edge "%s" into %s that the current graph does not have'
                                             % (prev.name, edge_name, prev.components[pu].name or pu))
                cmap[node.uuid] = match[0].uuid
                new_level[node.uuid] = match[0].uuid
                if node.graphs is not None:                      # a Module: pair its internal graphs (module.py:435-444)
                    cur_graphs = _module_graphs(match[0])
                    if cur_graphs is None:
                        raise SerializationError('saved component %s is a module, the current one (%s) is not' % (node.type, type(match[0]).__name__))
                    cmap.update(reconcile_graphs(cur_graphs, node.graphs[0], node.graphs[1:], module=match[0]))
        traverse = new_level


def reconcile_graphs(current_graphs, primary_previous, secondary_previous=None, module=None):
    """FactorGraph.reconcile_graphs (factor_graph.py:479-524): {saved uuid: current uuid}.  `module` (the current Module object when the
    graphs are a module's internal ones): the internal graphs of a module of this package are flat (no internal factors), so saved
    components that exist only inside the reference's module graphs (the latent function variable F, the internal Normal /
    GaussianProcess factors) have no counterpart and are skipped -- they carry no parameters.  The kernel's parameters DO live there in the
    reference (inputs of the internal GaussianProcess factors under their prefixed names, gp_regression.py:346-349, kernel.py:232-245):
    they are paired through those edge names."""
    secondary_previous = list(secondary_previous or [])
    views = [_View(g) for g in current_graphs]
    if len(views) - 1 != len(secondary_previous):
        raise SerializationError('Different number of secondary graphs: current %d, saved %d' % (len(views) - 1, len(secondary_previous)))
    cmap, traverse = {}, {}

    def named(prev, cur, trav):
        for name, pc in prev.named.items():
            cc = cur.named.get(name)
            if cc is None:
                if module is not None or pc.uuid in cmap:       # (a posterior shares the model's components: matched through the model already)
                    continue
                raise SerializationError('the saved graph %s names a component "%s" that the current graph %s does not have'
                                         % (prev.name, name, cur.name))
            cmap[pc.uuid] = cc.uuid
            trav[pc.uuid] = cc.uuid
            if pc.graphs is not None and _module_graphs(cc) is not None:
                cmap.update(reconcile_graphs(_module_graphs(cc), pc.graphs[0], pc.graphs[1:], module=cc))
    named(primary_previous, views[0], traverse)
    if module is None:
        _reconcile_level(traverse, cmap, views[0], primary_previous)
    for cg, pg in zip(views[1:], secondary_previous):
        trav = {pc: cc for pc, cc in cmap.items() if pc in pg.components}
        named(pg, cg, trav)
        if module is None:
            _reconcile_level(trav, cmap, cg, pg, strict=False)
    if module is not None:
        kern = module.__dict__.get('kernel', None)
        kpar = dict(kern.parameters) if kern is not None and hasattr(kern, 'parameters') else {}
        for pg in [primary_previous] + secondary_previous:
            for tu, preds in pg._pred.items():
                for edge_name, src in preds:
                    if edge_name in kpar and src.uuid not in cmap:
                        cmap[src.uuid] = kpar[edge_name].uuid
    return cmap
