import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from mxfusion_amd.inference.batch_loop import _Adam
N, Q, M, Dh, S = int(sys.argv[1]), 16, int(sys.argv[2]), 2, int(sys.argv[3])
X, Y, _ = bench.synth(N, Q, M)
for dtype in sys.argv[4:]:
    torch.manual_seed(0)
    infr, loop = bench.build_deepgp(N, Q, M, Dh, S, dtype, X, Y, False)
    td = torch.float64 if dtype == 'float64' else torch.float32
    data = [torch.as_tensor(X, dtype=td).cuda(), torch.as_tensor(Y, dtype=td).cuda()]
    ex = infr.create_executor(); tr = _Adam(infr.params, 1e-3)
    for i in range(4):
        loss = loop.step(ex, data, infr.params)
        g = infr.params.flat.grad
        print(dtype, i, float(loss), float(g.abs().max()), bool(torch.isfinite(g).all()), [int(a.svgp_log_pdf._last_info.sum()) for a in (infr._graphs[0].H.factor, infr._graphs[0].Y.factor)], flush=True)
        tr.step(1)
