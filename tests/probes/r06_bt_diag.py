"""T-shaped K-major product alone (gemm_bt.hip), with the probe build's timing diagnostics: MXF_BT_DIAG=1 reads the A fragments, =2 the Bt
fragments from LDS for an item's first k block only (results are WRONG; the time shows what those LDS reads cost).  usage: r06_bt_diag.py"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = """
import torch, sys
sys.path.insert(0, %r)
from mxfusion_amd import ops
M, SB = 1024, 2097152
pa = ops.f16x2_split(torch.randn(M, M, device='cuda')); Bt = torch.rand(M, SB, device='cuda'); pb = ops.f16x2_split(Bt); del Bt
out = torch.empty(M, SB, device='cuda'); w = torch.randn(M, device='cuda')
run = lambda: ops.gemm_f16x2_planes_kmajor(pa, pb, M, SB, M, out=out, blocked=True, w=w)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('%%.3f ms' %% (e0.elapsed_time(e1) / 10))
""" % root
for v in ('0', '1', '2', '0', '1', '2'):
    env = dict(os.environ, MXF_BT_DIAG=v, MXF_GP_LIB=os.path.join(root, 'mxfusion_amd', 'libmxf_gp_probe.so'))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print('MXF_BT_DIAG=%s: %s' % (v, (r.stdout.strip() or r.stderr[-300:])))
