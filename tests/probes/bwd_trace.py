"""Shader-clock stamps inside the SVGP reverse pass (probe build: tests/probes/build_bwd_trace_lib.sh, -DMXF_BWD_TRACE): where one wave's
cycles per 16 x 16 tile go.  Stages per row tile: 0 top, 1 T tile in registers, 2 weights computed (exp, multiplies), 3 weights split
(hi / lo f16), 4 matrix-pipe block issued and drained, 5 tile transposed into LDS / end.
usage: MXF_GP_LIB=mxfusion_amd/libmxf_gp_bwdtrace.so python tests/probes/bwd_trace.py"""
import ctypes
import os
import subprocess
import sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from mxfusion_amd import _lib
import bench  # noqa: F401  (the bench step is the workload)
sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras']
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
buf = np.zeros(8 * 8 * 8, dtype=np.uint32)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.mxf_debug_bwd_trace.argtypes = [ctypes.c_void_p]
assert lib.mxf_debug_bwd_trace(buf.ctypes.data) == 0
t = buf.reshape(8, 8, 8).astype(np.int64)
d = lambda a, b: (a - b) % (1 << 32)
print('column tile, row tile: T wait | weights | split | MFMA block | transpose+tail || tile total (cycles)')
for it in range(1, 6):
    for mt in range(8):
        s = t[it, mt]
        nxt = t[it, mt + 1, 0] if mt < 7 else t[it + 1, 0, 0]
        print('  %d %d: %5d %5d %5d %5d %5d || %5d' % (it, mt, d(s[1], s[0]), d(s[2], s[1]), d(s[3], s[2]), d(s[4], s[3]), d(s[5], s[4]), d(nxt, s[0])))
    a, b = t[it, 7], t[it + 1, 0]
    print('    between column tiles: last transposed product %5d | column flush (shuffles, dX atomics) %5d | next tile: loads requested %5d | '
          'residuals, splits %5d | first dots %5d' % (d(a[6], a[5]), d(a[7], a[6]), d(b[6], a[7]), d(b[7], b[6]), d(b[0], b[7])))
