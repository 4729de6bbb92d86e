"""r05: the reverse pass fused into the T product's epilogue (probe build, MXF_SVGP_FUSE=1) against the separate pass (MXF_SVGP_FUSE=0) and
against float64: every output of the float32 training call.  usage: fuse_check.py  (spawns itself once per setting)"""
import os, sys, subprocess, json
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
CASES = [(256, 256, 1, 8), (512, 1024, 2, 8), (256, 512, 1, 3), (1024, 2048, 2, 5)]


def run():
    import torch
    from mxfusion_amd import ops
    out = {}
    for (M, B, S, Q) in CASES:
        rng = np.random.default_rng(M + B + S + Q)
        X = rng.uniform(-3, 3, (S, B, Q)); Y = np.sin(X[0] @ rng.standard_normal((Q, 1))) + 0.05 * rng.standard_normal((B, 1))
        Z = rng.uniform(-3, 3, (M, Q))
        qm, qW, qd = 0.3 * rng.standard_normal((M, 1)), 0.3 * rng.standard_normal((M, M)) / np.sqrt(M), rng.uniform(0.05, 0.5, M)
        ls, var, noise = np.full(Q, 1.3), np.array([1.2]), np.array([0.03])
        for dt in (torch.float32, torch.float64):
            d = lambda a: torch.as_tensor(np.asarray(a), dtype=dt).cuda()
            r = ops.svgp_logpdf('rbf', d(X), d(Y[None]), d(Z), d(noise), d(qm), d(qW), d(qd), d(ls), d(var), True, jitter=1e-6, gscale=1.0 / S, want_grad=True)
            torch.cuda.synchronize()
            out['%d_%d_%d_%d_%s' % (M, B, S, Q, 'f32' if dt == torch.float32 else 'f64')] = {k: v.double().cpu().numpy().ravel().tolist() for k, v in r.items() if k != 'info'}
    json.dump(out, open(sys.argv[2], 'w'))


if len(sys.argv) > 1 and sys.argv[1] == 'child':
    run()
    sys.exit(0)
res = {}
for fuse in ('0', '1'):
    f = '/tmp/fuse_%s.json' % fuse
    env = dict(os.environ, MXF_SVGP_FUSE=fuse, MXF_GP_LIB=os.path.join(root, 'mxfusion_amd', 'libmxf_gp_probe.so'))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', f], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        print('child failed (fuse=%s):' % fuse, p.stderr[-3000:])
        sys.exit(1)
    res[fuse] = json.load(open(f))
nrm = lambda a, b: float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))
for (M, B, S, Q) in CASES:
    k32, k64 = '%d_%d_%d_%d_f32' % (M, B, S, Q), '%d_%d_%d_%d_f64' % (M, B, S, Q)
    print('M=%d B=%d S=%d Q=%d' % (M, B, S, Q))
    for key in sorted(res['0'][k32]):
        print('   %-8s separate vs f64 %.2e   fused vs f64 %.2e   fused vs separate %.2e' % (
            key, nrm(res['0'][k32][key], res['0'][k64][key]), nrm(res['1'][k32][key], res['1'][k64][key]), nrm(res['1'][k32][key], res['0'][k32][key])))
