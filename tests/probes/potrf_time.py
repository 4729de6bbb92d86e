"""potrf time (float64) at n = 512 / 1024 / 2048 with and without the persistent tile-dataflow kernel (MXF_POTRF_TILES, read once per process).
usage: potrf_time.py [sweep]"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == 'sweep':
    for v in ('1', '2', '0'):
        out = subprocess.run([sys.executable, __file__], env=dict(os.environ, MXF_POTRF_TILES=v), capture_output=True, text=True).stdout
        print('MXF_POTRF_TILES=%s | %s' % (v, ' | '.join(l.strip() for l in out.strip().splitlines())), flush=True)
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
torch.manual_seed(0)
for n in (512, 1024, 2048, 4096, 8192):
    X = torch.randn(n, 8, device='cuda', dtype=torch.float64)
    K = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(n, device='cuda', dtype=torch.float64)
    A = K[None].clone()
    ref = torch.linalg.cholesky(K)
    L, info = ops.potrf_(A.clone())
    err = float((L[0] - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bufs = [A.clone() for _ in range(10)]
    torch.cuda.synchronize()
    e0.record()
    for bb in bufs:
        ops.potrf_(bb)
    e1.record(); torch.cuda.synchronize()
    print('n=%d: %.3f ms, max err %.2e, info %d' % (n, e0.elapsed_time(e1) / 10, err, int(info[0])))
