import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxfusion_amd import ops
M, K = 1024, 2097152
A = torch.randn(1, K, M, device='cuda', dtype=torch.float32)
out = torch.empty(1, M, M, device='cuda', dtype=torch.float32)
ops.gemm(A, A, True, False, out=out); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): ops.gemm(A, A, True, False, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"lower={os.environ.get('MXF_GEMM_LOWER','0')} splitk={os.environ.get('MXF_GEMM_SPLITK','auto')}: {ms:.2f} ms", flush=True)
