"""A/B of the two f16x2 split-GEMM kernels (MXF_SPLIT_WIDE=0/1) at the T shape (1024 x SB x 1024) and the Psi2 shape (1024 x 1024 x SB,
lower blocks only).  usage: split_ab.py [SB]   -- re-runs itself once per setting (the knob is read once per process)."""
import os
import subprocess
import sys

if len(sys.argv) < 3:
    for wide in ('0', '1'):
        out = subprocess.run([sys.executable, __file__, sys.argv[1] if len(sys.argv) > 1 else '2097152', 'run'],
                             env=dict(os.environ, MXF_SPLIT_WIDE=wide), capture_output=True, text=True)
        print('WIDE=%s | %s' % (wide, ' | '.join(out.stdout.strip().splitlines())), out.stderr[-300:] if out.returncode else '', flush=True)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
M, SB = 1024, int(sys.argv[1])
A = torch.randn(M, M, device='cuda')
B = torch.rand(SB, M, device='cuda')
pa, pb = ops.f16x2_split(A), ops.f16x2_split(B)
ref_cols = (A.double() @ B[:512].double().T)
del B
out = torch.empty(M, SB, device='cuda')


def timeit(f, reps=3):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timeit(lambda: ops.gemm_f16x2_planes(pa, pb, M, SB, M, out=out))
err = float((out[:, :512].double() - ref_cols).abs().max() / ref_cols.abs().max())
print('T %d x %d x %d: %.2f ms %.0f TF err %.1e' % (M, SB, M, ms, 2.0 * M * M * SB / ms / 1e9, err))
del out, pb
torch.cuda.empty_cache()
C = torch.rand(M, SB, device='cuda')
pc = ops.f16x2_split(C)
psi = torch.zeros(M, M, device='cuda')
ms = timeit(lambda: ops.gemm_f16x2_planes(pc, pc, M, M, SB, out=psi, lower_only=True))
ref = (C[:, :].double() @ C.double().T)
e = float(((psi.double() - ref).tril().abs().max()) / ref.abs().max())
print('Psi2 %d x %d x %d lower: %.2f ms %.0f TF (of the lower half) err %.1e' % (M, M, SB, ms, 1.0 * M * M * SB / ms / 1e9, e))
