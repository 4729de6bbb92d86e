"""Gram kernel time at N=65536, Q=8 per kind and dtype (HIP events).  usage: gram_time.py [f32|f64] [sweep]
`sweep` re-runs itself per (MXF_GRAM_LEAN, MXF_GRAM_NW, MXF_GRAM_TR) setting (the knobs are read once per process)."""
import os
import subprocess
import sys

if len(sys.argv) > 2 and sys.argv[2] == 'sweep':
    for lean, nw, tr in ((1, 1, 16), (0, 1, 16), (1, 1, 32), (1, 1, 64), (0, 1, 64), (0, 4, 64), (1, 1, 8)):
        env = dict(os.environ, MXF_GRAM_NW=str(nw), MXF_GRAM_TR=str(tr), MXF_GRAM_LEAN=str(lean))
        out = subprocess.run([sys.executable, __file__, sys.argv[1]], env=env, capture_output=True, text=True).stdout
        print('LEAN=%d NW=%d TR=%2d | %s' % (lean, nw, tr, ' | '.join(l.strip() for l in out.strip().splitlines())), flush=True)
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
N, Q = 65536, 8
dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.float64
X = torch.rand(1, N, Q, device='cuda', dtype=dt) * 6 - 3
ls = torch.ones(1, Q, device='cuda', dtype=dt); var = torch.ones(1, 1, device='cuda', dtype=dt)
out = torch.empty(1, N, N, device='cuda', dtype=dt)
for kind in ('rbf', 'matern12', 'matern32', 'matern52'):
    for _ in range(2):
        ops.gram(kind, X, None, ls, var, True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ops.gram(kind, X, None, ls, var, True, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print('%s %s: %.3f ms %.2f TB/s' % (kind, sys.argv[1] if len(sys.argv) > 1 else 'f32', ms, N * N * out.element_size() / ms / 1e9))
