"""Is the per-rank step (few MC samples) paced by the host?  Enqueue time of K steps (host clock, no synchronisation inside) against their
wall time incl. the final synchronise, for the bench model at --samples S.  usage: host_bound.py [S] [K]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import bench
from mxfusion_amd.inference.batch_loop import _Adam
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, Q, M = 65536, 8, 1024
X, Y, Z = bench.synth(N, Q, M)
m, q, infr, loop, qX = bench.build(N, Q, M, S, 'float32', X, Y, Z, False)
Yd = torch.as_tensor(Y, dtype=torch.float32).cuda()
ex = infr.create_executor()
tr = _Adam(infr.params, 1e-3)
for _ in range(5):
    loop.step(ex, [Yd], infr.params); tr.step(batch_size=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    loop.step(ex, [Yd], infr.params); tr.step(batch_size=1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('S=%d: host enqueue %.3f ms/step, wall %.3f ms/step (device still busy for %.3f ms after the last enqueue)' %
      (S, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
# phases of the host time of one step
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    loop.step(ex, [Yd], infr.params); tr.step(batch_size=1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
