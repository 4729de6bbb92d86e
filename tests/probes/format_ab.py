"""f16 vs bf16 on the SAME kernel: the T-shaped product (1024 x SB x 1024, blocked output) of the wide split GEMM with its
operand planes filled directly with uniform random values of the format -- f16 planes through v_mfma_f32_32x32x16_f16 (the production
instruction), bf16 planes through v_mfma_f32_32x32x16_bf16 (probe build, MXF_SPLIT_BF16MFMA=1: same schedule, same loads, same bytes) --
and with all-zero planes.  What differs between the two runs is the multiplier's operand format only.
usage (probe build):  MXF_GP_LIB=mxfusion_amd/libmxf_gp_probe.so [MXF_SPLIT_BF16MFMA=1] python tests/probes/format_ab.py [SB]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops, _lib
M, SB = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
bf = os.environ.get('MXF_SPLIT_BF16MFMA', '0') == '1'
dt = torch.bfloat16 if bf else torch.float16
n_a, n_b = _lib.load().mxf_f32x3_plane_elems(M, M), _lib.load().mxf_f32x3_plane_elems(SB, M)
word = torch.full((1,), 8192.0, device='cuda').view(torch.int32)        # scale 1
out = torch.empty(M, SB, device='cuda')


def planes(n, kind):
    if kind == 'zero':
        return torch.zeros(2 * n, dtype=torch.int16, device='cuda')
    return (torch.rand(2 * n, device='cuda') * 2 - 1).to(dt).view(torch.int16)


def timed(pa, pb, reps=8):
    fn = lambda: ops.gemm_f16x2_planes((pa, word), (pb, word), M, SB, M, out=out, blocked=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for kind in ('random', 'zero', 'random'):
    ms = timed(planes(n_a, kind), planes(n_b, kind))
    print('%s %-6s planes: %.3f ms = %.0f TF of products (3 x 2 M N K)' % ('bf16' if bf else 'f16 ', kind, ms, 3 * 2.0 * M * M * SB / ms / 1e9), flush=True)
