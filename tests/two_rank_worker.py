"""Worker of tests/test_gpu_two_ranks.py (one process per GPU, launched through torch.distributed.run): SURVEY 8(e)'s parity check on real
RCCL.  The latent-input SVGP model of tests/test_gpu_api.py in float64 with INJECTED noise; rank r evaluates samples [r S/W, (r+1) S/W)
through the product's DistributedBatchInferenceLoop (torch.distributed 'nccl' all-reduce of the flat gradient), every rank also evaluates
ALL S samples with the single-process loop; the two flat gradients must agree to 1e-10.  The same local gradients are then summed a second
time through the C ABI's exchange (mxf_comm_init / mxf_allreduce_sum), which must give the same numbers."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(eps, S):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import StochasticVariationalInference, create_Gaussian_meanfield
    rng = np.random.RandomState(4)
    N, Q, M = 96, 3, 16
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=t(rng.randn(M, Q)))
    m.noise_var = Variable(transformation=PositiveTransformation(), initial_value=t([0.05]))
    kernel = RBF(input_dim=Q, ARD=True, variance=t([1.3]), lengthscale=t(rng.rand(Q) + 0.7), dtype='float64')
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype='float64')
    m.Y.factor.svgp_log_pdf.jitter = 1e-6
    q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype='float64')
    q[m.X].factor._rand_gen = MockRandomGenerator(t(eps))
    alg = StochasticVariationalInference(model=m, posterior=q, num_samples=S, observed=[m.Y])
    Y = t(np.sin(rng.randn(N, 1)))
    init = {'qm': t(0.2 * rng.randn(M, 1)), 'qW': t(0.1 * rng.randn(M, M)), 'qd': t(rng.rand(M) + 0.3), 'xm': t(rng.randn(N, Q)),
            'xv': t(rng.rand(N, Q) * 0.1 + 0.01)}
    return m, q, alg, Y, init


def flat_gradient(loop_cls, eps, S, weight_note):
    from mxfusion_amd.inference import GradBasedInference
    m, q, alg, Y, init = build(eps, S)
    loop = loop_cls()
    infr = GradBasedInference(alg, grad_loop=loop, dtype='float64')
    infr.initialize(Y=tuple(Y.shape))
    post = m.Y.factor._extra_graphs[0]
    qX = q[m.X].factor
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = init['qm'], init['qW'], init['qd']
    infr.params[qX.mean], infr.params[qX.variance] = init['xm'], init['xv']
    ex = infr.create_executor()
    loss = loop.step(ex, [Y], infr.params)
    return infr.params.flat.grad.detach().clone(), float(loss.detach()), loop


def rows_gradients():
    """(flat gradient, loss) of one MAP step on the SVGP notebook model through DistributedBatchInferenceLoop(shard='rows') and through the
    single-process BatchInferenceLoop on all rows."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import MAP, GradBasedInference, BatchInferenceLoop, DistributedBatchInferenceLoop
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()
    out = []
    for dist_loop in (True, False):
        rng = np.random.RandomState(12)
        N, Q, M = 128, 3, 16
        X = rng.uniform(-2, 2, (N, Q)); Y = np.sin(X[:, :1]) + 0.1 * rng.randn(N, 1)
        m = Model()
        m.N = Variable()
        m.X = Variable(shape=(m.N, Q))
        m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t([0.05]))
        kernel = RBF(input_dim=Q, ARD=True, variance=t([1.2]), lengthscale=t(rng.rand(Q) + 0.8), dtype='float64')
        m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, num_inducing=M, shape=(m.N, 1), dtype='float64')
        m.Y.factor.svgp_log_pdf.jitter = 1e-6
        loop = DistributedBatchInferenceLoop(shard='rows', row_variables=[m.Y]) if dist_loop else BatchInferenceLoop()
        infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype='float64')
        infr.initialize(X=(N, Q), Y=(N, 1))
        gp = m.Y.factor
        post = gp._extra_graphs[0]
        infr.params[gp.inducing_inputs] = t(rng.uniform(-2, 2, (M, Q)))
        infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t(0.3 * rng.randn(M, 1)), t(0.1 * rng.randn(M, M)), t(rng.rand(M) + 0.3)
        ex = infr.create_executor()
        loss = loop.step(ex, [t(X), t(Y)], infr.params)
        out += [infr.params.flat.grad.detach().clone(), float(loss.detach())]
    return out


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    from mxfusion_amd import ops
    from mxfusion_amd.inference import BatchInferenceLoop, DistributedBatchInferenceLoop
    S = 4 * world
    eps = np.random.RandomState(9).randn(S, 96, 3)
    g_all, loss_all, _ = flat_gradient(BatchInferenceLoop, eps, S, 'all samples, one process')
    lo, hi = rank * (S // world), (rank + 1) * (S // world)
    g_dist, loss_loc, _ = flat_gradient(DistributedBatchInferenceLoop, eps[lo:hi], S // world, 'this rank\'s samples, exchanged by the loop')
    scale = float(g_all.abs().max())
    err = float((g_dist - g_all).abs().max()) / scale
    assert err < 1e-10, ('torch.distributed nccl exchange', rank, err)
    # (r05: the loop reduces the loss over the ranks itself -- SURVEY 8(e) "flat gradient + scalar loss": every rank holds the job's objective)
    assert abs(loss_loc - loss_all) < 1e-10 * abs(loss_all), (loss_loc, loss_all)
    # rows sharded instead of samples (models without a sample axis): the SVGP notebook model, MAP on observed inputs -- every rank takes
    # half of the rows, the KL term carries weight 1 / 2, gradient and loss are SUMMED
    g_rows, loss_rows, g_ref, loss_ref = rows_gradients()
    err_rows = float((g_rows - g_ref).abs().max()) / float(g_ref.abs().max())
    assert err_rows < 1e-10 and abs(loss_rows - loss_ref) < 1e-10 * abs(loss_ref), ('row-sharded exchange', rank, err_rows, loss_rows, loss_ref)
    # the same sum through the C ABI: local gradient with weight 1 / world (single-process loop on the shard), mxf_allreduce_sum
    g_loc, _, _ = flat_gradient(BatchInferenceLoop, eps[lo:hi], S // world, 'this rank\'s samples, not exchanged')
    g_loc = g_loc / world
    uid = torch.zeros(128, dtype=torch.uint8, device='cuda')
    if rank == 0:
        uid.copy_(torch.as_tensor(list(ops.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    ops.comm_init(world, rank, bytes(uid.cpu().tolist()))
    try:
        ops.allreduce_sum_(g_loc)
        torch.cuda.synchronize()
        err2 = float((g_loc - g_all).abs().max()) / scale
        assert err2 < 1e-10, ('mxf_allreduce_sum', rank, err2)
        b = torch.full((5,), float(rank + 1), dtype=torch.float64, device='cuda')
        ops.bcast_(b, world - 1)
        torch.cuda.synchronize()
        assert torch.equal(b, torch.full_like(b, float(world)))
    finally:
        ops.comm_destroy()
    dist.barrier()
    if rank == 0:
        print('two-rank parity ok: nccl %.2e, mxf_comm %.2e' % (err, err2))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
