"""mxf_comm_* / mxf_allreduce_sum / mxf_bcast (include/mxf_gp.h): the RCCL exchange behind the C ABI.  One GPU is available to the test
run, so the communicator has ONE rank: the calls must go through RCCL (library found, communicator created on the handle's device,
collectives enqueued on the caller's stream) and a one-rank sum / broadcast is the identity.  The N > 1 semantics of the loops that use
the exchange are covered by the gloo tests (tests/test_distributed_gloo.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_allreduce_and_broadcast():
    from mxfusion_amd import ops, _lib
    uid = ops.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(_lib.MXFError):                 # no communicator yet
        ops.allreduce_sum_(torch.ones(4, device='cuda'))
    ops.comm_init(1, 0, uid)
    try:
        with pytest.raises(_lib.MXFError):             # a second communicator on the same handle is refused
            ops.comm_init(1, 0, uid)
        for dt in (torch.float32, torch.float64):
            g = torch.randn(1_060_000, device='cuda', dtype=dt)        # the size of the bench model's flat gradient
            ref = g.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                 # ordered on the caller's stream
                g.mul_(2.0)
                ops.allreduce_sum_(g)
                ops.bcast_(g, 0)
                g.mul_(0.5)
            s.synchronize()
            assert torch.equal(g, ref)
        with pytest.raises(_lib.MXFError):
            ops.bcast_(torch.ones(4, device='cuda'), root=3)
    finally:
        ops.comm_destroy()
    with pytest.raises(_lib.MXFError):
        ops.allreduce_sum_(torch.ones(4, device='cuda'))
