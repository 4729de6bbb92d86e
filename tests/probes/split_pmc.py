"""One-kernel workloads for the PMC passes of the split GEMMs.  usage: split_pmc.py t|tbt|tbt0|tzero|psi2|v|vp|tr [SB]
tbt (r06): the T shape through the K-MAJOR second operand (gemm_bt.hip: the Kuf planes Psi2 reads, LDS-DMA + transposing LDS reads) with the fused U row; tbt0: without U;
tzero: the T shape with an ALL-ZERO B operand (no operand toggling: the schedule's own ceiling, clock from GRBM_GUI_ACTIVE -- VERDICT r03 2a);
v: V = L^-1 Kuf of the whitened tier as the step runs it (triangular A; planes of V and of V^T and the partial sums of U from one launch);
vp: the same product writing the planes of V only; tr: the stand-alone planes transposition + U pass (mxf_f16x2_planes_transpose).
t: T = H0 Kuf shape (1024 x SB x 1024) with the output in 16-column blocks as the training step writes it (persistent 128 x 256 kernel); psi2: Kuf Kuf^T, lower blocks (1024 x 1024 x SB, 128 x 256 kernel).  3 launches."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import ops
which = sys.argv[1]
M, SB = 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 2097152
if which in ('v', 'vp', 'tr'):
    pa = ops.f16x2_split(torch.tril(torch.randn(M, M, device='cuda')))
    B = torch.rand(SB, M, device='cuda')
    pb = ops.f16x2_split(B)
    del B
    a = torch.randn(M, device='cuda')
    for _ in range(3):
        if which == 'v':
            pl, plT, U = ops.gemm_f16x2_planes_out(pa, pb, M, SB, M, alpha=64.0, a_lower=True, a=a)
        else:
            pl = ops.gemm_f16x2_planes_out(pa, pb, M, SB, M, alpha=64.0, a_lower=True)
    if which == 'tr':
        sc = torch.ones(1, device='cuda')
        for _ in range(3):
            ops.f16x2_planes_transpose(pl, M, SB, a=a, scale=sc)
elif which in ('tbt', 'tbt0'):
    pa = ops.f16x2_split(torch.randn(M, M, device='cuda'))
    Bt = torch.rand(M, SB, device='cuda')
    pb = ops.f16x2_split(Bt)
    del Bt
    out = torch.empty(M, SB, device='cuda')
    w = torch.randn(M, device='cuda') if which == 'tbt' else None
    for _ in range(3):
        ops.gemm_f16x2_planes_kmajor(pa, pb, M, SB, M, out=out, blocked=True, w=w)
elif which in ('t', 'tzero'):
    pa = ops.f16x2_split(torch.randn(M, M, device='cuda'))
    B = torch.zeros(SB, M, device='cuda') if which == 'tzero' else torch.rand(SB, M, device='cuda')
    pb = ops.f16x2_split(B)
    del B
    out = torch.empty(M, SB, device='cuda')
    for _ in range(3):
        ops.gemm_f16x2_planes(pa, pb, M, SB, M, out=out, blocked=True)
else:
    C = torch.rand(M, SB, device='cuda')
    pc = ops.f16x2_split(C)
    del C
    psi = torch.zeros(M, M, device='cuda')
    for _ in range(3):
        ops.gemm_f16x2_planes(pc, pc, M, M, SB, out=psi, lower_only=True)
torch.cuda.synchronize()
