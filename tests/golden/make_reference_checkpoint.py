"""Builds tests/golden/reference_format_gp.zip: a checkpoint laid out as the REFERENCE (amzn/MXFusion v0.3.1) writes one with
Inference.save (mxfusion/inference/inference.py:255-310), for the model of testing/inference/inference_serialization_test.py:175-220

    m.N = Variable(); m.X = Variable(shape=(m.N, 3)); m.noise_var = Variable(transformation=PositiveTransformation(), ...)
    kernel = RBF(input_dim=3, ARD=True, ...); m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, 1))
    Inference(MAP(model=m, observed=[m.X, m.Y]))

MXNet is not installable here, so the file is assembled BY HAND from the reference's source, independently of this package's writer:
  * zip members and encodings: inference.py:281-310, util/serialization.py:28-35;
  * graphs.json = [FactorGraph.as_json() for the inference's graphs] (inference.py:283; MAP: model + posterior) = networkx node_link_data
    of the MultiDiGraph of components (factor_graph.py:619-628), nodes {"id": component}, links {"name", "key", "source", "target"};
  * a component = {"uuid", "name", "attributes"} (+ "inherited_name" for a Variable, + "graphs" for a Module) + {"version": "1.0", "type":
    class name}: model_component.py:62-65, variable.py:99-102, module.py:475-479, serialization.py:42-53;
  * edges: predecessor -> successor, named as the successor's input / the factor's output (model_component.py:130-199);
  * the module's internal graphs: gp_regression.py:333-360 -- graph 'gp_regression' with X, noise_var (replicas: same uuid, same name),
    F ~ GaussianProcess(X, kernel) whose factor takes the kernel parameters as inputs 'rbf_lengthscale' / 'rbf_variance' (unnamed
    variables), Y ~ Normal(mean=F, variance=broadcast_to(noise_var)); and the posterior graph: a clone of it (variables keep their uuid
    and name, factors get new uuids, factor_graph.py:325-391) plus L, LinvY and a fresh X;
  * mxnet_parameters.npz: one array per PARAMETER variable, keyed by uuid, holding the UNCONSTRAINED value (softplus^-1 for a
    PositiveTransformation, var_trans.py:91); variable_constants.json: {uuid of N: 10}; configuration.json: {"observed": [X, Y]}.
Every uuid is written with a 'ref_' prefix: none of them can coincide with a uuid of the running script.

usage: python tests/golden/make_reference_checkpoint.py   (writes next to itself; deterministic)"""
import io
import json
import os
import zipfile

import numpy as np

VER = '1.0'


def comp(uuid, name, typ, attributes=(), variable=True, graphs=None):
    d = {'uuid': uuid, 'name': name, 'attributes': list(attributes)}
    if variable:
        d['inherited_name'] = None
    if graphs is not None:
        d['graphs'] = graphs
    d['version'] = VER
    d['type'] = typ
    return d


def node_link(name, nodes, edges):
    return {'directed': True, 'multigraph': True, 'graph': {}, 'nodes': [{'id': n} for n in nodes],
            'links': [{'name': e, 'key': e, 'source': s, 'target': t} for s, e, t in edges], 'name': name}


def inv_softplus(x):
    return np.log(np.expm1(x))


def main():
    u = lambda s: 'ref_' + s
    N = comp(u('N'), 'N', 'Variable')
    X = comp(u('X'), 'X', 'Variable', [u('N')])
    noise = comp(u('noise_var'), 'noise_var', 'Variable')
    Y = comp(u('Y'), 'Y', 'Variable', [u('N')])
    ls = comp(u('lengthscale'), None, 'Variable')
    var = comp(u('variance'), None, 'Variable')
    # ---- module graph 'gp_regression' (gp_regression.py:333-352)
    F = comp(u('F'), 'F', 'Variable', [u('N')])
    gpf = comp(u('GaussianProcess_factor'), None, 'GaussianProcess', variable=False)
    bcast = comp(u('broadcast_to_factor'), None, 'BroadcastToOperator', variable=False)
    nv_b = comp(u('noise_var_broadcast'), None, 'Variable')
    normal = comp(u('Normal_factor'), None, 'Normal', variable=False)
    inner_nodes = [X, noise, F, gpf, ls, var, Y, normal, nv_b, bcast, N]
    inner_edges = [(X, 'X', gpf), (ls, 'rbf_lengthscale', gpf), (var, 'rbf_variance', gpf), (gpf, 'random_variable', F),
                   (noise, 'data', bcast), (bcast, 'output', nv_b), (F, 'mean', normal), (nv_b, 'variance', normal), (normal, 'random_variable', Y)]
    g_inner = node_link('gp_regression', inner_nodes, inner_edges)
    # ---- its posterior: clone (variables keep uuid + name, factors new uuids) + L, LinvY, X (gp_regression.py:354-359)
    gpf2 = comp(u('GaussianProcess_factor_post'), None, 'GaussianProcess', variable=False)
    bcast2 = comp(u('broadcast_to_factor_post'), None, 'BroadcastToOperator', variable=False)
    normal2 = comp(u('Normal_factor_post'), None, 'Normal', variable=False)
    L = comp(u('post_L'), 'L', 'Variable', [u('N'), u('N')])
    LinvY = comp(u('post_LinvY'), 'LinvY', 'Variable', [u('N')])
    Xp = comp(u('post_X'), 'X', 'Variable', [u('N')])
    post_nodes = [X, noise, F, gpf2, ls, var, Y, normal2, nv_b, bcast2, N, L, LinvY, Xp]
    post_edges = [(X, 'X', gpf2), (ls, 'rbf_lengthscale', gpf2), (var, 'rbf_variance', gpf2), (gpf2, 'random_variable', F),
                  (noise, 'data', bcast2), (bcast2, 'output', nv_b), (F, 'mean', normal2), (nv_b, 'variance', normal2), (normal2, 'random_variable', Y)]
    g_post = node_link('posterior', post_nodes, post_edges)
    module = comp(u('GPRegression_module'), None, 'GPRegression', variable=False, graphs=[g_inner, g_post])
    # ---- the two graphs of a MAP inference (map.py:44-59: the model and Posterior(model) -- a clone of the model graph in which every latent
    # variable gets a PointMass; here nothing is latent).  The clone's module is a replica: new uuid, internal graphs cloned (module.py:446-465)
    g_model = node_link('model', [N, X, noise, module, Y], [(X, 'X', module), (noise, 'noise_var', module), (module, 'random_variable', Y)])
    module_q = comp(u('GPRegression_module_in_posterior'), None, 'GPRegression', variable=False, graphs=[g_inner, g_post])
    g_q = node_link('posterior', [N, X, noise, module_q, Y], [(X, 'X', module_q), (noise, 'noise_var', module_q), (module_q, 'random_variable', Y)])
    rng = np.random.RandomState(42)
    values = {'noise_var': np.array([0.37]), 'lengthscale': np.array([0.5, 0.6, 0.7]), 'variance': np.array([1.3])}
    Xc = rng.rand(10, 3)
    K = values['variance'] * np.exp(-0.5 * ((Xc[:, None, :] - Xc[None, :, :]) ** 2 / values['lengthscale'] ** 2).sum(-1)) + 0.37 * np.eye(10)
    Lc = np.linalg.cholesky(K)
    Yc = rng.rand(10, 1)
    params = {u('noise_var'): inv_softplus(values['noise_var']), u('lengthscale'): inv_softplus(values['lengthscale']),
              u('variance'): inv_softplus(values['variance']), u('post_L'): Lc, u('post_LinvY'): np.linalg.solve(Lc, Yc), u('post_X'): Xc}
    here = os.path.dirname(os.path.abspath(__file__))
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, 'a', zipfile.ZIP_DEFLATED, False) as zf:
        zf.writestr('graphs.json', json.dumps([g_model, g_q], ensure_ascii=False))
        b = io.BytesIO(); np.savez(b, **params); zf.writestr('mxnet_parameters.npz', b.getvalue())
        b = io.BytesIO(); np.savez(b); zf.writestr('mxnet_constants.npz', b.getvalue())
        zf.writestr('variable_constants.json', json.dumps({u('N'): 10}))
        zf.writestr('configuration.json', json.dumps({'observed': [u('X'), u('Y')]}))
        zf.writestr('version.json', json.dumps({'serialization_version': '2.0'}))
    with open(os.path.join(here, 'reference_format_gp.zip'), 'wb') as f:
        f.write(buf.getvalue())
    np.savez(os.path.join(here, 'reference_format_gp_expected.npz'), X=Xc, Y=Yc, L=Lc, **values)
    print('wrote reference_format_gp.zip (%d bytes)' % len(buf.getvalue()))


if __name__ == '__main__':
    main()
