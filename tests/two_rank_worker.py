"""Worker of tests/test_gpu_two_ranks.py (one process per rank, launched through torch.distributed.run): SURVEY 8(e)'s parity check with the
REAL HIP kernels in every rank.  Two modes:
  default                       one GPU per rank, torch.distributed 'nccl' (= RCCL) + the C ABI's own RCCL exchange (needs >= 2 GPUs);
  --backend gloo --same-device  every rank on cuda:0, 'gloo' collectives on DEVICE tensors (runs on a one-GPU box: per-rank library handles,
                                streams, scratch, guard slots and hipGraphs coexist in 2-4 processes; only RCCL itself is not exercised).
Besides the one-step gradient checks below, `training_parity` runs 3 Adam steps of the product loops (GradBasedInference.run, fused Adam
kernel) sharded by samples and by rows against the single-process loops: parameters equal to 1e-10, ONE collective per step.

RCCL mode:  The latent-input SVGP model of tests/test_gpu_api.py in float64 with INJECTED noise; rank r evaluates samples [r S/W, (r+1) S/W)
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


def _count_all_reduces():
    calls = []
    orig = dist.all_reduce

    def counted(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    dist.all_reduce = counted
    return calls


def _train_samples(distributed, eps, S, steps, use_graph=False):
    """`steps` Adam steps of the latent-input SVGP model through GradBasedInference.run; returns (flat parameters, the loop)."""
    from mxfusion_amd.inference import GradBasedInference, BatchInferenceLoop, DistributedBatchInferenceLoop
    m, q, alg, Y, init = build(eps, S)
    loop = DistributedBatchInferenceLoop(use_graph=use_graph) if distributed else BatchInferenceLoop(use_graph=use_graph)
    infr = GradBasedInference(alg, grad_loop=loop, dtype='float64')
    infr.initialize(Y=tuple(Y.shape))
    post = m.Y.factor._extra_graphs[0]
    qX = q[m.X].factor
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = init['qm'], init['qW'], init['qd']
    infr.params[qX.mean], infr.params[qX.variance] = init['xm'], init['xv']
    if distributed:                    # replicas may start anywhere: the loop broadcasts rank 0's parameters
        with torch.no_grad():
            infr.params.flat.add_(0.05 * dist.get_rank())
    infr.run(Y=Y, learning_rate=0.05, max_iter=steps)
    return infr.params.flat.detach().clone(), loop


def _train_rows(kind, steps):
    """MAP on the SVGP notebook model (no sample axis) through the row-sharded / single-process batch and minibatch loops."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import (MAP, GradBasedInference, BatchInferenceLoop, DistributedBatchInferenceLoop, MinibatchInferenceLoop,
                                        DistributedMinibatchInferenceLoop)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()
    rng = np.random.RandomState(12)
    N, Q, M, B = 256, 3, 16, 128
    X = rng.uniform(-2, 2, (N, Q)); Y = np.sin(X[:, :1]) + 0.1 * rng.randn(N, 1)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=t([0.05]))
    kernel = RBF(input_dim=Q, ARD=True, variance=t([1.2]), lengthscale=t(rng.rand(Q) + 0.8), dtype='float64')
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, num_inducing=M, shape=(m.N, 1), dtype='float64')
    m.Y.factor.svgp_log_pdf.jitter = 1e-6
    loop = {'batch': lambda: BatchInferenceLoop(), 'batch-rows': lambda: DistributedBatchInferenceLoop(shard='rows', row_variables=[m.Y]),
            'minibatch': lambda: MinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B}),
            'minibatch-rows': lambda: DistributedMinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B}, shard='rows')}[kind]()
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype='float64')
    infr.initialize(X=(N, Q), Y=(N, 1))
    gp = m.Y.factor
    post = gp._extra_graphs[0]
    infr.params[gp.inducing_inputs] = t(rng.uniform(-2, 2, (M, Q)))
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t(0.3 * rng.randn(M, 1)), t(0.1 * rng.randn(M, M)), t(rng.rand(M) + 0.3)
    perms = [rng.permutation(N) for _ in range(steps)]
    if 'minibatch' in kind:
        infr.run(X=t(X), Y=t(Y), learning_rate=0.05, max_iter=(steps + 1) // 2, permutations=perms)       # N / B = 2 minibatches per epoch
    else:
        infr.run(X=t(X), Y=t(Y), learning_rate=0.05, max_iter=steps)
    return infr.params.flat.detach().clone(), loop


def training_parity(rank, world, steps=3):
    """3 Adam steps of the PRODUCT loops with the real kernels in every rank == the single-process loops (run by every rank on all the
    work), parameters to 1e-10, and exactly one all-reduce per step."""
    S = 2 * world
    eps = np.random.RandomState(9).randn(S, 96, 3)
    lo, hi = rank * (S // world), (rank + 1) * (S // world)
    out = {}
    for use_graph in (False, True):
        ref, _ = _train_samples(False, eps, S, steps, use_graph)
        calls = _count_all_reduces()
        got, loop = _train_samples(True, eps[lo:hi], S // world, steps, use_graph)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-10, ('sample-sharded training', 'graph' if use_graph else 'eager', rank, err)
        assert len(calls) == steps and loop.collectives == steps, ('collectives per step', len(calls), loop.collectives, steps)
        out['samples-graph' if use_graph else 'samples'] = err
    for kind, nstep in (('batch', steps), ('minibatch', 4)):
        ref, _ = _train_rows(kind, nstep)
        calls = _count_all_reduces()
        got, loop = _train_rows(kind + '-rows', nstep)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-10, ('row-sharded training', kind, rank, err)
        assert len(calls) == nstep and loop.collectives == nstep, ('collectives per step', kind, len(calls), loop.collectives)
        out['rows-' + kind] = err
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--same-device', action='store_true', help='every rank on cuda:0 (gloo only: RCCL refuses two ranks on one GPU)')
    args = ap.parse_args()
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    if args.same_device:
        assert args.backend == 'gloo', '--same-device needs --backend gloo'
        local = 0
    torch.cuda.set_device(local)
    if args.backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        dist.init_process_group('gloo')
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
    train = training_parity(rank, world)
    if args.backend == 'gloo':         # the C ABI's exchange is RCCL: one rank per GPU only
        dist.barrier()
        if rank == 0:
            print('multi-rank parity ok (gloo on device tensors, %d ranks on cuda:%d): gradient %.2e, rows %.2e, training %s'
                  % (world, local, err, err_rows, ' '.join('%s %.1e' % kv for kv in sorted(train.items()))))
        dist.destroy_process_group()
        return
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
        print('two-rank parity ok: nccl %.2e, mxf_comm %.2e, training %s' % (err, err2, ' '.join('%s %.1e' % kv for kv in sorted(train.items()))))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
