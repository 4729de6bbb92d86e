"""r03: gradient error of the float32 SVGP step against float64 when the T product (gradients only) takes its big operand Kfu through the
HIGH f16 plane alone (two MFMA products, MXF_SPLIT_BHI=1 in the probe build) instead of hi + lo (three).  Prints, per length-scale, the
relative l2 error of every gradient and the largest element error of dX / dZ relative to the largest element.
usage: t_terms.py [B] [S]    (re-runs itself per setting: the knob is read once per process)"""
import os
import subprocess
import sys

if len(sys.argv) < 4:
    here = os.path.abspath(__file__)
    root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
    for bhi in (0, 1):
        env = dict(os.environ, MXF_SPLIT_BHI=str(bhi), MXF_GP_LIB=os.path.join(root, 'mxfusion_amd', 'libmxf_gp_probe.so'))
        a = sys.argv[1:3] + ['65536', '2'][len(sys.argv) - 1:]
        print('--- Kfu planes in T: %s' % ('hi only' if bhi else 'hi + lo'), flush=True)
        subprocess.run([sys.executable, here] + a + ['child'], env=env)
    sys.exit(0)

import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
M, Q, P = 1024, 8, 1
rng = np.random.default_rng(0)
X0 = rng.uniform(-3., 3., (B, Q))
w = rng.standard_normal(Q)
Y = np.sin(X0 @ w)[:, None] + 0.05 * rng.standard_normal((B, 1))
Z = X0[rng.permutation(B)[:M]].copy()
X = X0[None] + 0.1 * rng.standard_normal((S, B, Q))
qm = 0.3 * rng.standard_normal((M, P))
qW = 0.05 * rng.standard_normal((M, M)) / np.sqrt(M) * 8
qd = rng.uniform(0.05, 0.5, M)
noise = np.array([0.02]); var = np.array([1.0])


def run(dt, ls):
    d = lambda a: torch.as_tensor(np.asarray(a), dtype=dt).cuda()
    r = ops.svgp_logpdf('rbf', d(X), d(Y[None]), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, gscale=1.0 / S, want_grad=True)
    torch.cuda.synchronize()
    return {k: v.double().cpu().numpy() for k, v in r.items()}


for l in (1.0, 1.5, 2.2):
    ls = np.full(Q, l)
    r64, r32 = run(torch.float64, ls), run(torch.float32, ls)
    line = 'l=%.1f ELBO rel %.1e |' % (l, np.abs(r32['logL'] - r64['logL']).max() / np.abs(r64['logL']).max())
    for k in ('dX', 'dZ', 'dls', 'dvar', 'dmu', 'dW', 'dSdiag', 'dnoise'):
        a, b = r32[k].ravel(), r64[k].ravel()
        line += ' %s %.1e' % (k, np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    for k in ('dX', 'dZ'):
        line += ' | max|%s err|/max|%s| %.1e' % (k, k, np.abs(r32[k] - r64[k]).max() / np.abs(r64[k]).max())
    print(line, flush=True)
