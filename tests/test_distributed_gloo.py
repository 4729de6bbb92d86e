"""world_size-2 `gloo` test (CPU) of the N>1 path: Monte-Carlo samples sharded over ranks, ONE all-reduce of the flat
gradient per step (DistributedBatchInferenceLoop).  The per-rank objective is evaluated by the oracle here (tests may use
it); what is under test is the product's sharding / exchange logic: sharded + all-reduced == unsharded, to 1e-12."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _count_all_reduces():
    """Wrap torch.distributed.all_reduce: SURVEY 8(e) asks for ONE collective per step ('flat gradient + scalar loss')."""
    calls = []
    orig = dist.all_reduce

    def counted(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    dist.all_reduce = counted
    return calls


class _Params(object):
    """Minimal stand-in for InferenceParameters: one flat autograd leaf."""

    def __init__(self, flat):
        self.flat = flat.clone().requires_grad_(True)


def _problem():
    rng = np.random.RandomState(0)
    N, Q, M, S = 12, 2, 4, 6
    Y = rng.rand(N, 1)
    eps = rng.randn(S, N, Q)
    sizes = dict(qX_mean=(N, Q), qX_var=(N, Q), noise_var=(1,), lengthscale=(Q,), variance=(1,), qU_mean=(M, 1), qU_cov_W=(M, M),
                 qU_cov_diag=(M,), Z=(M, Q))
    flat = torch.as_tensor(rng.randn(sum(int(np.prod(s)) for s in sizes.values())) * 0.3, dtype=torch.float64)
    return Y, eps, sizes, flat


def _executor(Y, eps, sizes, params):
    k = O.RBF(2, ARD=True)

    def run(*_):
        raw, off = {}, 0
        for n, shp in sizes.items():
            cnt = int(np.prod(shp))
            raw[n] = params.flat[off:off + cnt].view(shp)
            off += cnt
        loss = O.svi_latent_svgp_loss(k, O.T(Y), raw['Z'], raw, O.T(eps), jitter=1e-6)
        return loss, loss
    return run


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mxfusion_amd.inference import DistributedBatchInferenceLoop
    Y, eps, sizes, flat = _problem()
    S = eps.shape[0]
    shard = eps[rank * (S // world):(rank + 1) * (S // world)]          # this rank's MC samples
    params = _Params(flat)
    loop = DistributedBatchInferenceLoop()
    loss = loop.step(_executor(Y, shard, sizes, params), [None], params)
    t = loss.detach().clone()
    dist.all_reduce(t)
    q.put((rank, params.flat.grad.clone().numpy(), float(t) / world))
    dist.destroy_process_group()


def test_sharded_samples_allreduce_equals_unsharded():
    Y, eps, sizes, flat = _problem()
    params = _Params(flat)
    loss, _ = _executor(Y, eps, sizes, params)()
    loss.backward()
    ref_grad, ref_loss = params.flat.grad.numpy(), float(loss)

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grad, mean_loss in outs:
        assert np.allclose(grad, ref_grad, rtol=1e-12, atol=1e-12), rank     # identical on every rank, equal to 1-process
        assert abs(mean_loss - ref_loss) < 1e-10 * abs(ref_loss)


# ---- world 8, 4 samples per rank: what the 8-GPU run of BASELINE.json configs[2] does -------------------------------
def _problem8():
    rng = np.random.RandomState(8)
    N, Q, M, S = 10, 2, 3, 32
    Y = rng.rand(N, 1)
    eps = rng.randn(S, N, Q)
    sizes = dict(qX_mean=(N, Q), qX_var=(N, Q), noise_var=(1,), lengthscale=(Q,), variance=(1,), qU_mean=(M, 1), qU_cov_W=(M, M),
                 qU_cov_diag=(M,), Z=(M, Q))
    flat = torch.as_tensor(rng.randn(sum(int(np.prod(s)) for s in sizes.values())) * 0.3, dtype=torch.float64)
    return Y, eps, sizes, flat


def _batch_loop_cls():
    from mxfusion_amd.inference import DistributedBatchInferenceLoop

    class Loop(DistributedBatchInferenceLoop):
        def _make_trainer(self, param_dict, learning_rate, optimizer):
            return _CpuAdam(param_dict, learning_rate)
    return Loop


def _worker8(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Y, eps, sizes, flat = _problem8()
    S = eps.shape[0]
    shard = eps[rank * (S // world):(rank + 1) * (S // world)]          # 4 MC samples per rank
    params = _Params(flat + (0.25 * rank))                              # replicas must start from rank 0's parameters (the loop broadcasts)
    loop = _batch_loop_cls()()
    calls = _count_all_reduces()
    loop.run(_executor(Y, shard, sizes, params), [None], params, None, learning_rate=0.05, max_iter=3)
    assert len(calls) == 3 and loop.collectives == 3, (len(calls), loop.collectives)        # one all-reduce per step: gradient + loss together
    q.put((rank, params.flat.detach().clone().numpy()))
    dist.destroy_process_group()


def test_world_8_four_samples_per_rank_equals_the_32_sample_single_process_run():
    """DistributedBatchInferenceLoop.run (the PRODUCT loop: broadcast of the start parameters, per-step all-reduce of the flat gradient, optimiser
    step) with world 8 and 4 MC samples per rank == BatchInferenceLoop.run with all 32 samples in one process: parameters after 3 Adam
    steps agree to 1e-10 on every rank."""
    from mxfusion_amd.inference import BatchInferenceLoop
    Y, eps, sizes, flat = _problem8()

    class Single(BatchInferenceLoop):
        def _make_trainer(self, param_dict, learning_rate, optimizer):
            return _CpuAdam(param_dict, learning_rate)
    params = _Params(flat)
    Single().run(_executor(Y, eps, sizes, params), [None], params, None, learning_rate=0.05, max_iter=3)
    ref = params.flat.detach().numpy()
    assert np.abs(ref - flat.numpy()).max() > 1e-2

    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in outs) == list(range(world))
    for rank, got in outs:
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-10), (rank, np.abs(got - ref).max())


# ---- minibatches x sample sharding (BASELINE.json configs[3]): the PRODUCT loop (DistributedMinibatchInferenceLoop.run) end to end ------
class _CpuAdam(object):
    """MXNet Adam on the flat CPU leaf (the product's trainer is the HIP kernel mxf_adam_step; the loop's trainer seam swaps it here)."""

    def __init__(self, params, lr):
        self.p, self.opt, self.key = params, O.MXNetAdam(lr), 'flat'

    def step(self, batch_size=1):
        with torch.no_grad():
            new = self.opt.step({self.key: self.p.flat.detach().clone()}, {self.key: self.p.flat.grad.clone()}, batch_size=batch_size)[self.key]
            self.p.flat.data.copy_(new)
        self.p.flat.grad = None


def _mb_problem():
    rng = np.random.RandomState(3)
    N, Q, M, S, B = 24, 2, 4, 4, 8
    X, Y = rng.rand(N, Q), rng.rand(N, 1)
    eps = rng.randn(S, B, Q)               # reparameterisation noise of the uncertain inputs, one row block per minibatch position
    sizes = dict(qx_var=(1,), noise_var=(1,), lengthscale=(Q,), variance=(1,), qU_mean=(M, 1), qU_cov_W=(M, M), qU_cov_diag=(M,), Z=(M, Q))
    flat = torch.as_tensor(rng.randn(sum(int(np.prod(s)) for s in sizes.values())) * 0.3, dtype=torch.float64)
    perms = [rng.permutation(N) for _ in range(3)]
    return X, Y, eps, sizes, flat, perms, B


def _mb_executor(eps, sizes, params, scaling, weight):
    """SVI objective of the uncertain-input SVGP (inputs X ~ N(Xobs, 0.01), q(X) = N(Xobs, softplus(qx_var))) on one minibatch, for the
    MC samples given by `eps`, weighted `weight` (= 1 / world for a shard)."""
    k = O.RBF(2, ARD=True)
    sp = O.softplus

    def run(Xb, Yb):
        raw, off = {}, 0
        for n, shp in sizes.items():
            cnt = int(np.prod(shp))
            raw[n] = params.flat[off:off + cnt].view(shp)
            off += cnt
        qv = sp(raw['qx_var']).reshape(1, 1, 1)
        Xs = O.normal_draw(Xb[None], qv, O.T(eps))
        kp = {'rbf_lengthscale': sp(raw['lengthscale'])[None], 'rbf_variance': sp(raw['variance'])[None]}
        lp = O.factor_sum(O.svgp_log_pdf(k, Xs, Yb[None], raw['Z'][None], sp(raw['noise_var'])[None], raw['qU_mean'][None], raw['qU_cov_W'][None],
                                        sp(raw['qU_cov_diag'])[None], kp, jitter=1e-6, log_pdf_scaling=scaling))
        lpx = O.factor_sum(O.normal_log_pdf(Xb[None], torch.full((1, 1, 1), 0.01, dtype=Xs.dtype), Xs, log_pdf_scaling=scaling))
        lq = O.factor_sum(O.normal_log_pdf(Xb[None], qv, Xs, log_pdf_scaling=scaling))
        loss = -(lp + lpx - lq) * weight
        return loss, loss
    return run


def _mb_loop_cls():
    from mxfusion_amd.inference import DistributedMinibatchInferenceLoop

    class Loop(DistributedMinibatchInferenceLoop):
        def _make_trainer(self, param_dict, learning_rate, optimizer):
            return _CpuAdam(param_dict, learning_rate)
    return Loop


def _mb_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    X, Y, eps, sizes, flat, perms, B = _mb_problem()
    S = eps.shape[0]
    shard = eps[rank * (S // world):(rank + 1) * (S // world)]
    # rank 1 starts from DIFFERENT parameters and would draw DIFFERENT shuffles: the loop must broadcast both from rank 0
    params = _Params(flat + (0.5 if rank == 1 else 0.0))
    loop = _mb_loop_cls()(batch_size=B)
    my_perms = perms if rank == 0 else [p[::-1].copy() for p in perms]
    calls = _count_all_reduces()
    loop.run(_mb_executor(shard, sizes, params, X.shape[0] / B, 1.0), [O.T(X), O.T(Y)], params, None, learning_rate=0.05, max_iter=3,
             permutations=my_perms)
    assert len(calls) == 9 and loop.collectives == 9, (len(calls), loop.collectives)        # 3 epochs x 3 minibatches, one all-reduce each
    q.put((rank, params.flat.detach().clone().numpy()))
    dist.destroy_process_group()


def test_distributed_minibatch_loop_equals_the_single_process_loop():
    """3 epochs x 3 minibatches of 8 rows, 4 MC samples: 2 ranks x 2 samples through DistributedMinibatchInferenceLoop.run == 1 process with
    all 4 samples through the same loop -- parameters after 9 Adam steps agree to 1e-10 on every rank."""
    X, Y, eps, sizes, flat, perms, B = _mb_problem()
    params = _Params(flat)
    loop = _mb_loop_cls()(batch_size=B)
    loop.run(_mb_executor(eps, sizes, params, X.shape[0] / B, 1.0), [O.T(X), O.T(Y)], params, None, learning_rate=0.05, max_iter=3,
             permutations=perms)
    ref = params.flat.detach().numpy()
    assert np.abs(ref - flat.numpy()).max() > 1e-2          # the optimiser moved

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got in outs:
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-10), (rank, np.abs(got - ref).max())


# ---- row sharding (second axis): models WITHOUT a sample axis -- the reference's svgp_regression notebook (MAP on observed X) ---------
# The PRODUCT classes run here end to end on CPU tensors -- Model / SVGPRegression / MAP / GradBasedInference / ObjectiveBlock /
# FactorGraph.log_pdf / Module.log_pdf / the distributed loops with their weights (global_weight -> kl_weight) --; only the three places that
# launch HIP kernels are swapped for the oracle: the SVGP bound of one call, softplus, Adam.
def _cpu_patches():
    import torch.nn.functional as Fn
    from mxfusion_amd import ops
    from mxfusion_amd.inference import batch_loop, minibatch_loop
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    ops.softplus = lambda x: Fn.softplus(x)
    ops.softplus_bwd_ = lambda x, dy, out: out.copy_(dy * torch.sigmoid(x))

    def _compute(self, F, variables):
        kern = O.RBF(self.model.kernel.input_dim, ARD=True)
        kp = {'rbf_' + k.split('_', 1)[1]: v for k, v in self.model.kernel.fetch_parameters(variables).items()}
        return O.svgp_log_pdf(kern, variables[self.model.X], variables[self.model.Y], variables[self.model.inducing_inputs],
                              variables[self.model.noise_var], variables[self.posterior.qU_mean], variables[self.posterior.qU_cov_W],
                              variables[self.posterior.qU_cov_diag], kp, jitter=self.jitter, log_pdf_scaling=self.log_pdf_scaling)
    SVGPRegressionLogPdf._compute = _compute

    class CpuTrainer(object):
        def __init__(self, params, lr, optimizer='adam'):
            self.p, self.opt = params, O.MXNetAdam(lr)

        def step(self, batch_size=1):
            with torch.no_grad():
                new = self.opt.step({'flat': self.p.flat.detach().clone()}, {'flat': self.p.flat.grad.clone()}, batch_size=batch_size)['flat']
                self.p.flat.data.copy_(new)
            self.p.zero_grad()
    batch_loop._Adam = CpuTrainer
    minibatch_loop._Adam = CpuTrainer


def _notebook_model(rng, N, M):
    """The model of examples/notebooks/svgp_regression.ipynb (:100-121), 2-D inputs."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    X = rng.uniform(-3, 3, (N, 2))
    Y = np.sin(X[:, :1]) + 0.3 * X[:, 1:] + 0.1 * rng.randn(N, 1)
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 2))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.05)
    m.kernel = RBF(input_dim=2, ARD=True, variance=1., lengthscale=np.ones(2))
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, inducing_inputs=None, num_inducing=M, shape=(m.N, 1))
    m.Y.factor.svgp_log_pdf.jitter = 1e-6
    return m, X, Y


def _run_rows(loop_kind, world_rank=None):
    """3+ Adam steps of MAP on the notebook model through GradBasedInference.run with the given loop; returns (flat parameters, last loss)."""
    from mxfusion_amd.inference import (MAP, GradBasedInference, MinibatchInferenceLoop, BatchInferenceLoop, DistributedMinibatchInferenceLoop,
                                        DistributedBatchInferenceLoop)
    _cpu_patches()
    rng = np.random.RandomState(11)
    N, M, B = 64, 5, 32
    m, X, Y = _notebook_model(rng, N, M)
    init = dict(Z=rng.uniform(-3, 3, (M, 2)), mu=0.3 * rng.randn(M, 1), W=0.2 * rng.randn(M, M), d=rng.uniform(0.5, 1.0, M))
    perms = [rng.permutation(N) for _ in range(2)]
    if loop_kind == 'minibatch':
        loop = MinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B})
    elif loop_kind == 'minibatch-rows':
        loop = DistributedMinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B}, shard='rows')
    elif loop_kind == 'batch':
        loop = BatchInferenceLoop()
    else:
        loop = DistributedBatchInferenceLoop(shard='rows', row_variables=[m.Y])
    infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype='float64', context=torch.device('cpu'))
    infr.initialize(X=(N, 2), Y=(N, 1))
    gp = m.Y.factor
    post = gp._extra_graphs[0]
    infr.params[gp.inducing_inputs] = init['Z']
    infr.params[post.qU_mean] = init['mu']
    infr.params[post.qU_cov_W] = init['W']
    infr.params[post.qU_cov_diag] = init['d']
    if world_rank:                               # replicas may start anywhere: the loop broadcasts rank 0's parameters
        with torch.no_grad():
            infr.params.flat.add_(0.1 * world_rank)
    losses = []
    orig_step = loop.step

    def step(*a, **k):
        out = orig_step(*a, **k)
        losses.append(float(out.detach()))
        return out
    loop.step = step
    calls = _count_all_reduces()
    if 'minibatch' in loop_kind:
        infr.run(X=X, Y=Y, learning_rate=0.05, max_iter=2, permutations=perms if not world_rank else [p[::-1].copy() for p in perms])
    else:
        infr.run(X=X, Y=Y, learning_rate=0.05, max_iter=3)
    if dist.is_initialized() and dist.get_world_size() > 1:
        assert len(calls) == len(losses) == loop.collectives, (len(calls), len(losses))     # ONE collective per step (gradient + loss)
    return infr.params.flat.detach().clone().numpy(), losses


def _rows_worker(rank, world, port, q, loop_kind):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat, losses = _run_rows(loop_kind, world_rank=rank)
    q.put((rank, flat, losses))
    dist.destroy_process_group()


def _single_worker(q, loop_kind):
    torch.set_num_threads(1)
    q.put(_run_rows(loop_kind))


@pytest.mark.parametrize('loop_kind,world', [('minibatch', 8), ('batch', 8), ('minibatch', 3)])
def test_world_8_row_sharded_loops_equal_the_single_process_loops(loop_kind, world):
    """The reference's SVGP notebook model has NO sample axis (MAP on observed inputs): with shard='rows' each of 8 ranks evaluates 1/8 of the
    rows of every (mini)batch, the module's KL term carries weight 1/8, gradient AND loss are summed over the ranks -- parameters after 4
    (minibatch) / 3 (batch) Adam steps equal the single-process loop to 1e-10 on every rank, and so does every step's loss."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p0 = ctx.Process(target=_single_worker, args=(q, loop_kind))      # (its own process: the CPU patches replace module attributes)
    p0.start()
    ref, ref_losses = q.get(timeout=300)
    p0.join(timeout=60)
    assert p0.exitcode == 0 and len(ref_losses) >= 3

    port = _free_port()           # (world 3: 32 rows over 3 ranks = 11 + 11 + 10 -- ragged shares add up all the same)
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q, loop_kind + '-rows')) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in outs) == list(range(world))
    for rank, got, losses in outs:
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-10), (rank, np.abs(got - ref).max())
        assert np.allclose(losses, ref_losses, rtol=1e-10), (rank, losses, ref_losses)        # the loss is reduced over the ranks too
