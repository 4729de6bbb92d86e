"""Longer optimisation of the bench model (latent-input SVGP, SVI) in float32 (split-GEMM training path) and float64: the loss
trajectories must stay finite and track each other.  usage: train_probe.py [N] [steps] [lr]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-2
Q, M, S = 8, 1024, 8
X, Y, Z = bench.synth(N, Q, M)
from mxfusion_amd.inference.batch_loop import _Adam
out = {}
for dtype in ('float32', 'float64'):
    torch.manual_seed(0)
    m, q, infr, loop, qX = bench.build(N, Q, M, S, dtype, X, Y, Z, False)
    td = torch.float32 if dtype == 'float32' else torch.float64
    Yd = torch.as_tensor(Y, dtype=td).cuda()
    ex = infr.create_executor()
    opt = _Adam(infr.params, lr)
    tr = []
    for it in range(steps):
        loss = loop.step(ex, [Yd], infr.params)
        opt.step()
        if it % 25 == 0 or it == steps - 1:
            tr.append((it, float(loss.detach())))
    info = int(m.Y.factor.svgp_log_pdf._last_info.abs().sum())
    ls = infr.params[m.Y.factor.kernel.lengthscale].double().cpu().numpy()
    from mxfusion_amd.modules.gp_modules._fused import Float32Guard
    torch.cuda.synchronize()
    g = m.Y.factor.svgp_log_pdf._f32_guard()
    print(dtype, 'guard: level', Float32Guard.NAMES[g.tier], 'switches', g.switches, 'cond max %.3e last %.3e' % (g.cond_max, g.cond_last), flush=True)
    out[dtype] = (tr, info, ls, float(infr.params[m.noise_var]))
    del infr, m, q, ex
    torch.cuda.empty_cache()
for (i, a), (_, b) in zip(out['float32'][0], out['float64'][0]):
    print('iter %4d  loss f32 %14.2f   f64 %14.2f   rel diff %.2e' % (i, a, b, abs(a - b) / abs(b)))
print('potrf info f32 / f64:', out['float32'][1], out['float64'][1])
print('lengthscale f32', np.round(out['float32'][2], 4), 'noise', out['float32'][3])
print('lengthscale f64', np.round(out['float64'][2], 4), 'noise', out['float64'][3])
