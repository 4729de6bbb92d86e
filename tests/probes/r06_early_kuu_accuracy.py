"""r06: float64 gradient error of the uncertain-input SVGP toy step (tests/test_gpu_config4.py, M = 4 inducing points in [0, 1]^2: cond(Kuu) ~ 1e6)
against the oracle's autograd, for the core reverse mode's two formulations (probe build: MXF_SVGP_EARLY_KUU=0 / 1), over a few seeds."""
import os, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from oracle import gp_oracle as O
from test_gpu_config4 import build_uncertain_input_svgp, _t
from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator

worst = {}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rng = np.random.RandomState(seed)
    N, Q, M, B, S = 24, 2, 4, 8, 4
    X, Y, Z = rng.rand(N, Q), rng.rand(N, 1), rng.rand(M, Q)
    eps = rng.randn(S, B, Q)
    m, q, infr, loop, kernel = build_uncertain_input_svgp(N, Q, M, B, S, 'float64', _t(Z))
    post = m.Y.factor._extra_graphs[0]
    qm, qW, qd = rng.randn(M, 1) * 0.1, rng.randn(M, M) * 0.05, rng.rand(M) + 0.5
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = _t(qm), _t(qW), _t(qd)
    q[m.X].factor._rand_gen = MockRandomGenerator(_t(eps.reshape(-1)))
    ex = infr.create_executor()
    sel = rng.permutation(N)[:B]
    loss = loop.step(ex, [_t(X[sel]), _t(Y[sel])], infr.params)
    sp, isp = O.softplus, O.inv_softplus
    raw = {'qx_var': isp(O.T([1e-2])), 'noise_var': isp(O.T([0.01])), 'lengthscale': isp(O.T(np.ones(Q))), 'variance': isp(O.T([1.0])),
           'qU_mean': O.T(qm), 'qU_cov_W': O.T(qW), 'qU_cov_diag': isp(O.T(qd)), 'Z': O.T(Z)}
    lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    ref = O.svi_uncertain_input_svgp_loss(O.RBF(Q, ARD=True), O.T(X[sel]), O.T(Y[sel]), lv, O.T(eps), prior_var=1e-2, jitter=1e-6, log_pdf_scaling=N / B)
    ref.backward()
    P = infr.params
    g = P.flat.grad
    for var, name in ((kernel.lengthscale, 'lengthscale'), (kernel.variance, 'variance'), (m.Z, 'Z'), (post.qU_mean, 'qU_mean'), (post.qU_cov_W, 'qU_cov_W')):
        o, n, _ = P._slices[var.uuid]
        a, b = g[o:o + n].cpu().numpy(), lv[name].grad.numpy().ravel()
        e = float(np.abs(a - b).max() / np.abs(b).max())
        worst[name] = max(worst.get(name, 0.0), e)
    Kuu = O.RBF(Q, ARD=True)
print('EARLY_KUU=%s' % os.environ.get('MXF_SVGP_EARLY_KUU', 'default'), {k: '%.2e' % v for k, v in worst.items()})
