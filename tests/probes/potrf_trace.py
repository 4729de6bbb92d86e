"""Stage timestamps of the persistent Cholesky kernel (probe build: hipcc ... -DMXF_POTRF_TRACE, see potrf_trace.sh): the critical path of
potrf(1024) block column by block column, in microseconds.  Stages per tile (i, j < i): 0 start, 1 left-looking product done, 2 L[j][j]
arrived, 3 operands in LDS, 4 solved, 5 published, 6 diagonal share accumulated; diagonal tile: 7 start, 8 in LDS, 9 factored, 10 published."""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
X = torch.randn(n, 8, device='cuda', dtype=torch.float64)
K = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(n, device='cuda', dtype=torch.float64)
for _ in range(3):
    ops.potrf_(K[None].clone())
torch.cuda.synchronize()
buf = np.zeros(16 * 17 * 16, dtype=np.int64)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mxf_debug_potrf_trace.argtypes = [ctypes.c_void_p]
assert lib.mxf_debug_potrf_trace(buf.ctypes.data) == 0
t = buf.reshape(16, 17, 16).astype(np.float64) / 100.0       # us
t0 = t[0, 0, 7]
nb = min(n // 64, 16)
print('diagonal tiles (us from the start of row 0): start, in LDS, factored, published | factor, publish')
for i in range(nb):
    d = t[i, i]
    print('  %2d: %7.1f %7.1f %7.1f %7.1f | %5.1f %5.1f' % (i, d[7] - t0, d[8] - t0, d[9] - t0, d[10] - t0, d[9] - d[8], d[10] - d[9]))
print('first 16-column round of the diagonal factor of row 3: 16x16 factor %.1f, rows below %.1f, trailing tiles %.1f us' % (t[3, 3, 11] - t[3, 3, 8], t[3, 3, 12] - t[3, 3, 11], t[3, 3, 13] - t[3, 3, 12]))
print('sub-diagonal tile (i, i-1), relative to the publication of L[i-1][i-1]: arrived, in LDS, solved, published, diag share; then diagonal start/in LDS/factored/published')
for i in range(1, nb):
    e, d, pub = t[i, i - 1], t[i, i], t[i - 1, i - 1, 10]
    print('  %2d: prod done %6.1f | arr %5.1f lds %5.1f solved %5.1f pub %5.1f cd %5.1f | dstart %5.1f dlds %5.1f fact %5.1f pub %5.1f' % (
        i, e[1] - pub, e[2] - pub, e[3] - pub, e[4] - pub, e[5] - pub, e[6] - pub, d[7] - pub, d[8] - pub, d[9] - pub, d[10] - pub))
