#!/usr/bin/env python
"""bench.py -- headline benchmark of the GP + SVI hot path on MI355X.

A "step" = one full SVI step (reparameterised draw of S latent-input samples, Monte-Carlo SVGP ELBO, its
reverse mode, the data-parallel gradient exchange, one Adam update) of BASELINE.json configs[2]:
    SVGPRegression, RBF(ARD), N=65536, Q=8, M=1024 inducing points, S=32 MC samples,
    StochasticVariationalInference, latent-input model of testing/modules/svgpregression_test.py:357-385,
run through the mxfusion_amd API (Model / SVGPRegression / GradBasedInference) -> C ABI -> HIP kernels.
At --gpus N>1 (one process per GPU, torch.distributed 'nccl' == RCCL) the S samples are sharded S/N per rank and
the flat gradient is summed with ONE all-reduce per step (strong scaling: the total work is fixed).

Prints ONE JSON line (rank 0).  `roofline` is the RBF Gram build at N=65536, Q=8 (HBM-write bound, the kernel
BASELINE.json's target names); `roofline_mfma` is the dominant MFMA kernel of the step; `cpu_baseline` is the oracle
(CPU restatement of the reference's op sequence, fwd+bwd+Adam) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

# three HIP streams of the step (main + two side streams) plus torch's and RCCL's own: with the default of 4 hardware queues two of the
# step's streams share one as soon as RCCL is initialised and the two Cholesky chains serialise (+1.2 ms per step at 4 samples per GPU).
# Read when the HIP runtime initialises (first device call), so it must be in the environment before that.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth(N, Q, M, seed=0):
    """SURVEY 8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3., 3., (N, Q))
    w = rng.standard_normal(Q)
    Y = np.sin(X @ w)[:, None] + 0.05 * rng.standard_normal((N, 1))
    Z = X[rng.permutation(N)[:M]].copy()
    return X, Y, Z


def build(N, Q, M, S_local, dtype, X, Y, Z, distributed, use_graph=False):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, create_Gaussian_meanfield, \
        BatchInferenceLoop, DistributedBatchInferenceLoop
    m = Model()
    m.N = Variable()
    m.X = Normal.define_variable(mean=0, variance=1, shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=Z)
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=dtype)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    q = create_Gaussian_meanfield(model=m, observed=[m.Y], dtype=dtype)
    loop = DistributedBatchInferenceLoop(use_graph=use_graph) if distributed else BatchInferenceLoop(use_graph=use_graph)
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S_local, observed=[m.Y]),
                              grad_loop=loop, dtype=dtype)
    infr.initialize(Y=(N, 1))
    post = gp._extra_graphs[0]
    dev = infr.mxnet_context
    td = torch.float32 if dtype == 'float32' else torch.float64
    infr.params[post.qU_mean] = torch.zeros(M, 1, dtype=td, device=dev)
    infr.params[post.qU_cov_W] = torch.zeros(M, M, dtype=td, device=dev)
    infr.params[post.qU_cov_diag] = torch.ones(M, dtype=td, device=dev)
    qX = q[m.X].factor
    infr.params[qX.mean] = torch.as_tensor(X, dtype=td).to(dev)
    infr.params[qX.variance] = torch.full((N, Q), 1e-2, dtype=td, device=dev)
    return m, q, infr, loop, qX


def build_minibatch(N, Q, M, B, S_local, dtype, Z, prior_var=1e-2):
    """BASELINE.json configs[3]: the SVGP config with minibatches of B rows (rv_scaling = N / B) and the MC samples sharded over the GPUs.
    Minibatches and per-row latent inputs are compatible in MXFusion's API when every factor is a sum over rows: X ~ N(Xobs, s) row-wise,
    Y ~ SVGP(X), q(X) = N(Xobs, v) with ONE shared variance parameter (the minibatch loop slices the observed rows Xobs, Y and scales all
    three factors by N / B).  S_local samples of X per step and rank."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.models.posterior import Posterior
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, DistributedMinibatchInferenceLoop
    m = Model()
    m.N = Variable()
    m.Xobs = Variable(shape=(m.N, Q))
    m.X = Normal.define_variable(mean=m.Xobs, variance=prior_var, shape=(m.N, Q), dtype=dtype)
    m.Z = Variable(shape=(M, Q), initial_value=Z)
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=dtype)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6
    q = Posterior(m)
    q.qx_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=1e-2)
    q[m.X].set_prior(Normal(mean=q[m.Xobs], variance=q.qx_var, dtype=dtype))
    loop = DistributedMinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B, m.X: N / B})
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S_local, observed=[m.Xobs, m.Y]), grad_loop=loop,
                              dtype=dtype)
    infr.initialize(Xobs=(B, Q), Y=(B, 1))
    post = gp._extra_graphs[0]
    dev = infr.mxnet_context
    td = torch.float32 if dtype == 'float32' else torch.float64
    infr.params[post.qU_mean] = torch.zeros(M, 1, dtype=td, device=dev)
    infr.params[post.qU_cov_W] = torch.zeros(M, M, dtype=td, device=dev)
    infr.params[post.qU_cov_diag] = torch.ones(M, dtype=td, device=dev)
    return m, infr, loop


def build_minibatch_rows(N, Q, M, B, dtype, Z, proxy_world=0):
    """Row-sharded data parallelism for a model WITHOUT a sample axis (the reference's svgp_regression notebook: MAP on observed inputs, one
    evaluation of the bound per minibatch): Y ~ SVGP(X), minibatches of B rows, every rank takes B / world rows of each minibatch, the KL term
    weighted 1 / world, gradient and loss summed (DistributedMinibatchInferenceLoop(shard='rows')).  proxy_world = W > 0: ONE process evaluates
    the share of one rank of a W-GPU run (B / W rows, KL weight 1 / W, no collective) -- the per-rank cost without a W-GPU node."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, DistributedMinibatchInferenceLoop
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z = Variable(shape=(M, Q), initial_value=Z)
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, inducing_inputs=m.Z, shape=(m.N, 1), dtype=dtype)
    gp = m.Y.factor
    gp.svgp_log_pdf.jitter = 1e-6

    class ProxyLoop(DistributedMinibatchInferenceLoop):
        def _world(self):
            return proxy_world

        def _rank(self):
            return 0

        def _next_permutation(self, N_, device, generator, permutations):
            return super(DistributedMinibatchInferenceLoop, self)._next_permutation(N_, device, generator, permutations)

        def _exchange(self, param_dict, loss):
            return loss

        def step(self, infr_executor, batch, param_dict, update_shape_constants=None):
            batch = [torch.tensor_split(d, proxy_world)[0] for d in batch]
            return super(DistributedMinibatchInferenceLoop, self).step(infr_executor, batch, param_dict, update_shape_constants)
    cls = ProxyLoop if proxy_world > 1 else DistributedMinibatchInferenceLoop
    loop = cls(batch_size=B, rv_scaling={m.Y: N / B}, shard='rows')
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=dtype)
    infr.initialize(X=(B, Q), Y=(B, 1))
    post = gp._extra_graphs[0]
    dev = infr.mxnet_context
    td = torch.float32 if dtype == 'float32' else torch.float64
    infr.params[post.qU_mean] = torch.zeros(M, 1, dtype=td, device=dev)
    infr.params[post.qU_cov_W] = torch.zeros(M, M, dtype=td, device=dev)
    infr.params[post.qU_cov_diag] = torch.ones(M, dtype=td, device=dev)
    return m, infr, loop


_RANK_TIMES = {}      # filled by _finish_timing: this rank's own time of the timed region, gathered over the ranks


def _mark_loop(loop, steps):
    """Called right before a timed region: remember the loop's collective counter so that the line can state collectives per step."""
    _RANK_TIMES['mark'] = (loop, getattr(loop, 'collectives', 0), steps)


def _finish_timing(t0, distributed):
    """Closing bracket of the timed region: synchronize, barrier, MAX over the ranks.  Also records every rank's own time up to its
    synchronize (before the barrier) in _RANK_TIMES['per_rank_s'] -- the spread shows load imbalance, max - own the wait in the barrier."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0
    mark = _RANK_TIMES.pop('mark', None)
    if mark is not None and distributed:
        _RANK_TIMES['collectives_per_step'] = (getattr(mark[0], 'collectives', 0) - mark[1]) / float(mark[2])
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        own = [torch.zeros(1, dtype=torch.float64, device='cuda') for _ in range(dist.get_world_size())]
        dist.all_gather(own, torch.tensor([t_own], dtype=torch.float64, device='cuda'))
        _RANK_TIMES['per_rank_s'] = [float(x) for x in own]
    else:
        _RANK_TIMES['per_rank_s'] = [t_own]
    return (dt,)


def dist_report(world, steps, grad_elems, dtype):
    """What a reader needs to check an N-GPU line without the builder in the loop: the rank count the communicator itself reports, every
    rank's own ms per step, and the cost of the step's one collective (all-reduce of a flat gradient of this model's size) measured on its
    own with HIP events -- 20 back-to-back all-reduces on RCCL's stream order, mean per call."""
    import torch.distributed as dist
    nranks = dist.get_world_size() if dist.is_initialized() else 1
    backend = _RANK_TIMES.get('backend', 'nccl')
    rep = {"ranks": nranks, "backend": backend if dist.is_initialized() else None, "same_device": bool(_RANK_TIMES.get('same_device', False)),
           "per_rank_ms_per_step": [round(t / steps * 1e3, 4) for t in _RANK_TIMES.get('per_rank_s', [])]}
    if backend == 'nccl' or not dist.is_initialized():
        rep["rccl_ranks"] = nranks
    if 'collectives_per_step' in _RANK_TIMES:
        rep["collectives_per_step"] = _RANK_TIMES['collectives_per_step']       # all-reduces the loop issued in the timed region / steps
    if dist.is_initialized():
        buf = torch.zeros(int(grad_elems) + 2, dtype=dtype, device='cuda')
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        rep["allreduce_ms"] = e0.elapsed_time(e1) / 20
        rep["allreduce_bytes"] = buf.numel() * buf.element_size()
    return rep


def time_minibatch_steps(infr, loop, Xd, Yd, B, steps, warmup, lr, distributed):
    """One step = one minibatch: slice B rows of a fixed shuffle (the same on every rank), forward + reverse mode on this rank's MC samples,
    gradient all-reduce, Trainer.step(batch_size=B)."""
    import torch.distributed as dist
    executor = infr.create_executor()
    trainer = loop._make_trainer(infr.params, lr, 'adam')
    N = Xd.shape[0]
    perm = torch.randperm(N, device=Xd.device, generator=torch.Generator(device=Xd.device).manual_seed(99))
    nb = N // B

    def one(i):
        sel = perm[(i % nb) * B:(i % nb + 1) * B]
        loss = loop.step(executor, [Xd[sel], Yd[sel]], infr.params)
        trainer.step(batch_size=B)
        return loss
    loss = None
    for i in range(warmup):
        loss = one(i)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    _mark_loop(loop, steps)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = one(warmup + i)
    return _finish_timing(t0, distributed) + (float(loss.detach()),)


def time_minibatch_run(infr, data_kw, epochs, lr, distributed):
    """The same minibatch steps through the PRODUCT entry point, GradBasedInference.run (initialise, executor, shuffles, slicing, the loop's
    epoch-loss bookkeeping, Trainer): one untimed epoch, then `epochs` timed ones; returns seconds per minibatch step."""
    import torch.distributed as dist
    N = next(iter(data_kw.values())).shape[0]
    nb = N // infr._grad_loop.batch_size
    gen = torch.Generator(device='cuda').manual_seed(5)
    infr.run(max_iter=1, learning_rate=lr, generator=gen, **data_kw)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infr.run(max_iter=epochs, learning_rate=lr, generator=gen, **data_kw)
    return _finish_timing(t0, distributed)[0] / (epochs * nb)


def build_deepgp(N, Q, M, Dh, S_local, dtype, X, Y, distributed):
    """BASELINE.json configs[4]: two chained SVGPRegression modules, first layer Matern52 + RBF (AddKernel), hidden layer H (N x Dh) with a
    mean-field q(H), second layer RBF-ARD on the sampled H.  Built purely from the API."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF, Matern52
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, StochasticVariationalInference, create_Gaussian_meanfield, \
        BatchInferenceLoop, DistributedBatchInferenceLoop
    rng = np.random.default_rng(3)
    td = torch.float32 if dtype == 'float32' else torch.float64
    side = int(math.ceil(M ** (1.0 / Dh)))
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.Z0 = Variable(shape=(M, Q), initial_value=X[rng.permutation(N)[:M]].copy())
    # second-layer inducing inputs on a lattice over the range of H with a length-scale of about one lattice spacing: Kuu stays well
    # conditioned, which the float32 streaming form needs
    side = int(math.ceil(M ** (1.0 / Dh)))
    grid = np.stack(np.meshgrid(*[np.linspace(-1.2, 1.2, side)] * Dh, indexing='ij'), -1).reshape(-1, Dh)[:M]
    m.Z1 = Variable(shape=(M, Dh), initial_value=grid + 0.01 * rng.standard_normal((M, Dh)))
    m.noise0 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.noise1 = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    k0 = Matern52(Q, ARD=True, variance=1., lengthscale=np.full(Q, 2.0), dtype=dtype) + RBF(Q, ARD=True, variance=1., lengthscale=np.full(Q, 2.0), dtype=dtype)
    k1 = RBF(Dh, ARD=True, variance=1., lengthscale=np.full(Dh, 2.4 / max(side - 1, 1)), name='rbf_top', dtype=dtype)
    m.H = SVGPRegression.define_variable(X=m.X, kernel=k0, noise_var=m.noise0, inducing_inputs=m.Z0, shape=(m.N, Dh), dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.H, kernel=k1, noise_var=m.noise1, inducing_inputs=m.Z1, shape=(m.N, 1), dtype=dtype)
    for gp in (m.H.factor, m.Y.factor):
        gp.svgp_log_pdf.jitter = 1e-5
    q = create_Gaussian_meanfield(model=m, observed=[m.X, m.Y], dtype=dtype)
    loop = DistributedBatchInferenceLoop() if distributed else BatchInferenceLoop()
    infr = GradBasedInference(StochasticVariationalInference(model=m, posterior=q, num_samples=S_local, observed=[m.X, m.Y]), grad_loop=loop,
                              dtype=dtype)
    infr.initialize(X=(N, Q), Y=(N, 1))
    dev = infr.mxnet_context
    for gp, P in ((m.H.factor, Dh), (m.Y.factor, 1)):
        post = gp._extra_graphs[0]
        infr.params[post.qU_mean] = torch.zeros(M, P, dtype=td, device=dev)
        infr.params[post.qU_cov_W] = torch.zeros(M, M, dtype=td, device=dev)
        infr.params[post.qU_cov_diag] = torch.ones(M, dtype=td, device=dev)
    qH = q[m.H].factor
    infr.params[qH.mean] = torch.as_tensor(np.tanh(X[:, :Dh]), dtype=td).to(dev)
    infr.params[qH.variance] = torch.full((N, Dh), 1e-2, dtype=td, device=dev)
    return infr, loop


def build_gp(N, Q, dtype, X, Y):
    """BASELINE.json configs[1]: GPRegression, RBF-ARD, exact Cholesky (MAP of the kernel hyper-parameters and the noise)."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, Q))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
    m.Y = GPRegression.define_variable(X=m.X, kernel=kernel, noise_var=m.noise_var, shape=(m.N, 1), dtype=dtype)
    loop = BatchInferenceLoop()
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=dtype)
    infr.initialize(X=(N, Q), Y=(N, 1))
    return m, infr, loop


class _PendulumPolicy(torch.nn.Module):
    """testing/inference/pilco_test.py:29-37: Dense(100, relu) -> Dense(1, tanh), times 2."""

    def __init__(self, ds):
        super().__init__()
        self.l1 = torch.nn.Linear(ds, 100)
        self.l2 = torch.nn.Linear(100, 1)

    def forward(self, x):
        return torch.tanh(self.l2(torch.relu(self.l1(x)))) * 2


def _pendulum_cost(state, action):
    """testing/inference/pilco_test.py:39-58."""
    return (2. * (state[:, :, 0:1] - 1) ** 2).sum(-1) + (.001 * action ** 2).sum(-1) + (.1 * state[:, :, 2:3] ** 2).sum(-1)


def bench_pilco(N, S, T, dtype, steps, warmup, use_graph, cpu_baseline=True):
    """the PILCO rollout (pilco_alg.py:72-90) as a GP-predict-in-a-loop latency benchmark.  A GPRegression dynamics
    model on N conditioning points (3 state + 1 action inputs, 3 outputs; the pendulum shapes of testing/inference/pilco_test.py), S
    trajectories rolled T time steps under a Dense(100)-Dense(1) policy; one step = rollout + reverse pass + Adam on the policy."""
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import GPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, GradTransferInference, PILCOAlgorithm, BatchInferenceLoop
    from mxfusion_amd.inference.batch_loop import _Adam
    td = torch.float64 if dtype == 'float64' else torch.float32
    rng = np.random.RandomState(0)
    X = rng.rand(N, 4)
    Y = np.stack([np.sin(X @ rng.randn(4)) for _ in range(3)], 1) + 0.05 * rng.randn(N, 3)
    s0 = rng.rand(S, 3)
    t = lambda a: torch.as_tensor(a, dtype=td).cuda()
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 4))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.kernel = RBF(input_dim=4, variance=1, lengthscale=1, ARD=True, dtype=dtype)
    m.Y = GPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, shape=(m.N, 3), dtype=dtype)
    m.Y.factor.gp_log_pdf.jitter = 1e-6
    infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.X, m.Y]), dtype=dtype)
    infr.run(X=t(X), Y=t(Y), max_iter=3, learning_rate=0.1)
    torch.manual_seed(0)
    policy = _PendulumPolicy(3).to(td)
    ref_state = {k: v.clone() for k, v in policy.state_dict().items()}
    policy.cuda()
    s0d = t(s0)
    alg = PILCOAlgorithm(model=m, observed=[m.X, m.Y], cost_function=_pendulum_cost, policy=policy, n_time_steps=T,
                         initial_state_generator=lambda n: s0d, num_samples=S)
    loop = BatchInferenceLoop(use_graph=bool(use_graph))
    ip = GradTransferInference(alg, infr_params=infr.params, train_params=list(policy.parameters()), grad_loop=loop, dtype=dtype)
    Xd, Yd = t(X), t(Y)
    ip.initialize(X=Xd, Y=Yd)
    ex = ip.create_executor()
    opt = _Adam(ip.params, 1e-3)
    first = None
    for _ in range(max(warmup, 3 if use_graph else 1)):
        loss = loop.step(ex, [Xd, Yd], ip.params)
        first = float(loss.detach()) if first is None else first
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = loop.step(ex, [Xd, Yd], ip.params)
        opt.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"metric": "policy-gradient steps/sec, PILCO rollout over a GPRegression dynamics model (SURVEY 8(f) rank 4)", "value": 1.0 / dt,
           "unit": "policy-gradient steps/sec", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3,
           "us_per_rollout_time_step": dt / T * 1e6, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64" if dtype == 'float64' else "f32", "data": "synthetic",
           "config": {"workload": "PILCOAlgorithm: GPRegression RBF-ARD N=%d (4 inputs, 3 outputs), %d trajectories x %d time steps, "
                                  "Dense(100)-Dense(1) policy; step = rollout + reverse pass + Adam%s" % (N, S, T, ", hipGraph replay" if use_graph else "")},
           "first_loss": first, "last_loss": float(loss.detach())}
    if cpu_baseline:
        from oracle import gp_oracle as O
        k = O.RBF(4, ARD=True)
        ls, var, noise = (infr.params[v].double().cpu() for v in (m.kernel.lengthscale, m.kernel.variance, m.noise_var))
        kp = {'rbf_lengthscale': ls[None], 'rbf_variance': var[None]}
        post = O.gp_log_pdf(k, O.T(X)[None], O.T(Y)[None], noise[None], kp, jitter=1e-6, return_posterior=True)[1]
        pol = _PendulumPolicy(3).double()
        pol.load_state_dict({k_: v.double() for k_, v in ref_state.items()})
        pred = lambda xt: O.gp_predict(k, xt, noise[None], post[0][None], post[1][None], post[2][None], kp)
        times = []
        for _ in range(3):
            for p_ in pol.parameters():
                p_.grad = None
            c0 = time.perf_counter()
            ref = O.pilco_rollout(pred, pol, _pendulum_cost, O.T(s0), T)
            ref.backward()
            times.append(time.perf_counter() - c0)
        out["cpu_baseline"] = {"value": 1.0 / min(times), "unit": "policy-gradient steps/sec", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "the same rollout + reverse pass through oracle/gp_oracle.py (torch-CPU float64), best of 3",
                               "first_loss": float(ref.detach())}
    return out


def cpu_baseline_gp(N, Q, X, Y, reps=2):
    """The oracle's MAP step of the exact GP (Gram, potrf, trsm, autograd backward, Adam; float64, torch-CPU LAPACK/BLAS) at the full N."""
    from oracle import gp_oracle as O
    T = O.T
    kern = O.RBF(Q, ARD=True)
    raw = {'lengthscale': O.inv_softplus(T(np.ones(Q))), 'variance': O.inv_softplus(T([1.0])), 'noise_var': O.inv_softplus(T([0.01]))}
    opt = O.MXNetAdam(1e-3)
    hw = os.cpu_count() or 1
    th = min(hw, 32)
    torch.set_num_threads(th)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        loss = O.map_gp_loss(kern, T(X), T(Y), lv, jitter=0.)
        loss.backward()
        raw = opt.step({k: v.detach() for k, v in lv.items()}, {k: v.grad for k, v in lv.items()})
        times.append(time.perf_counter() - t0)
    t = min(times)
    return {"value": 1.0 / t, "unit": "MAP-steps/sec", "cores": th, "kind": "port",
            "sample": "oracle MAP step of the exact GP at the full N=%d (fwd + autograd bwd + Adam, float64, torch-CPU, %d threads): %.2f s" % (N, th, t)}


def time_steps_multi(infr, loop, data, steps, warmup, lr, distributed):
    from mxfusion_amd.inference.batch_loop import _Adam
    import torch.distributed as dist
    executor = infr.create_executor()
    trainer = _Adam(infr.params, lr)
    loss = None
    for _ in range(warmup):
        loss = loop.step(executor, data, infr.params)
        trainer.step(batch_size=1)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    _mark_loop(loop, steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = loop.step(executor, data, infr.params)
        trainer.step(batch_size=1)
    return _finish_timing(t0, distributed) + (float(loss.detach()),)


def set_trained_like(m, infr, Q, M, td):
    """The parameters a 300-step optimisation of the bench model ends near (tests/probes/train_probe.py): length-scale 2.2 (cond_1(Kuu) ~ 1.4e3),
    noise 0.02, a non-trivial q(u).  At the initial point (length-scale 1, qU_W = 0) 94 % of the f16 operand planes are zero and Kuu ~ I."""
    rngq = np.random.default_rng(11)
    qm_t, qW_t, qd_t = 0.3 * rngq.standard_normal((M, 1)), 0.4 * rngq.standard_normal((M, M)) / math.sqrt(M), rngq.uniform(0.05, 0.5, M)
    gp_ = m.Y.factor
    post_ = gp_._extra_graphs[0]
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=td).cuda()
    infr.params[gp_.kernel.lengthscale] = tt(np.full(Q, 2.2))
    infr.params[m.noise_var] = tt([0.02])
    infr.params[post_.qU_mean], infr.params[post_.qU_cov_W], infr.params[post_.qU_cov_diag] = tt(qm_t), tt(qW_t), tt(qd_t)


def step_breakdown(infr, loop, Yd, lr, M, SB):
    """In-step durations of the training call's bulk kernels (HIP events inside the library on the stream each kernel runs on,
    mxf_svgp_timing) over three further steps, and the rooflines they give: the Gram -> planes passes against HBM (4 bytes per covariance
    written), the T and Psi2 / Phi products against the f16 matrix pipe / 3."""
    from mxfusion_amd import _lib
    from mxfusion_amd.inference.batch_loop import _Adam
    dev = torch.cuda.current_device()
    executor = infr.create_executor()
    trainer = _Adam(infr.params, lr)
    _lib.svgp_timing(dev, True)
    acc, n = {}, 0
    try:
        for _ in range(3):
            loop.step(executor, [Yd], infr.params)
            trainer.step(batch_size=1)
            for k, v in _lib.svgp_timing_read(dev).items():
                acc[k] = acc.get(k, 0.0) + v
            n += 1
    finally:
        _lib.svgp_timing(dev, False)
    ms = {k: v / n for k, v in acc.items()}
    out = {"step_breakdown_ms": {k: round(v, 4) for k, v in ms.items()}}
    pb = 4.0 * M * SB                                # two f16 planes per covariance
    whitened = 'v_gemm' in ms                        # (whitened form: the second planes of the step are V^T, written by the V product's own epilogue)
    for key, name in (('planes_a', 'first'), ('planes_b', 'second')):
        if key in ms and not (whitened and key == 'planes_b'):
            out.setdefault("roofline_planes", {})[name] = {"bound": "hbm", "achieved": pb / ms[key] / 1e6, "peak": 8000.0, "unit": "GB/s",
                                                           "frac": pb / ms[key] / 1e6 / 8000.0, "ms_in_step": ms[key], "algorithmic_bytes": pb}
    if "roofline_planes" in out:
        out["roofline_planes"]["kernel"] = ("gram_planes_lean_kernel (Gram written as two f16 planes); whitened form: one Gram planes pass, the planes of V and V^T "
                                            "come out of the V product (step_breakdown_ms.v_gemm; planes_b = the reduction of its partial sums of U)")
    if whitened:
        f = 2.0 * M * M * SB * sum(r + 1 for r in range(M // 256)) / (M // 256) ** 2 if M % 256 == 0 else 2.0 * M * M * SB
        out["roofline_mfma_v_in_step"] = {"bound": "mfma", "kernel": "V = L^-1 Kuf inside the step (triangular A: flops counted per 256-row tile, k up to the tile's last row)",
                                          "achieved": f / ms['v_gemm'] / 1e9, "peak": 2500.0 / 3, "unit": "TFLOP/s", "frac": f / ms['v_gemm'] / 1e9 / (2500.0 / 3),
                                          "ms_in_step": ms['v_gemm'], "bytes_written": 2 * pb}
    if 't_gemm' in ms:
        f = 2.0 * M * M * SB
        out["roofline_mfma_in_step"] = {"bound": "mfma", "kernel": "T product inside the step", "achieved": f / ms['t_gemm'] / 1e9, "peak": 2500.0 / 3,
                                        "unit": "TFLOP/s", "frac": f / ms['t_gemm'] / 1e9 / (2500.0 / 3), "ms_in_step": ms['t_gemm']}
    return out


def time_steps(infr, loop, Yd, steps, warmup, lr, distributed):
    from mxfusion_amd.inference.batch_loop import _Adam
    import torch.distributed as dist
    executor = infr.create_executor()
    trainer = _Adam(infr.params, lr)
    loss = None
    for _ in range(warmup):
        loss = loop.step(executor, [Yd], infr.params)
        trainer.step(batch_size=1)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    _mark_loop(loop, steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = loop.step(executor, [Yd], infr.params)
        trainer.step(batch_size=1)
    return _finish_timing(t0, distributed) + (float(loss.detach()),)


def _pmc_traffic(stem):
    """(bytes per launch, source) from the newest profiles/r<NN>_<stem>.json that holds `hbm_traffic_bytes_per_launch`, else (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s.json' % stem)), reverse=True):
        try:
            with open(f) as fh:
                pj = json.load(fh)
        except (OSError, ValueError):
            continue
        if 'hbm_traffic_bytes_per_launch' in pj:
            return int(pj['hbm_traffic_bytes_per_launch']), 'profiles/%s (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)' % os.path.basename(f)
    return None, None


def box_spread():
    """The same default command on other boxes of this pool (committed JSON lines, newest round's profiles/r<NN>_bench_full*.json): the step and the
    Gram roofline vary by a few per cent with the box's power-limited clocks -- reported next to this run's own numbers."""
    import glob
    out = []
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_full*.json')))
    newest = max([os.path.basename(f)[:3] for f in files] or ['r00'])
    for f in files:
        if not os.path.basename(f).startswith(newest):
            continue
        try:
            with open(f) as fh:
                d = json.loads(fh.read().strip().splitlines()[-1])
            out.append({"file": 'profiles/' + os.path.basename(f), "ms_per_step": d.get("ms_per_step"), "roofline_frac": (d.get("roofline") or {}).get("frac"),
                        "roofline_mfma_frac": (d.get("roofline_mfma") or {}).get("frac")})
        except (OSError, ValueError, IndexError):
            continue
    return out


def gram_roofline(N, Q, dtype, reps=200):
    """RBF Gram at N x N, Q: algorithmic bytes = N*N*sizeof written + 2*N*Q*sizeof read, timed with HIP
    events on the stream the kernel is launched on (torch's current stream).  200 timed launches behind 20 untimed ones.  The launch time
    RAMPS after the idle gap in front of this measurement (the 17 GB output is allocated first): rocprofv3 per-launch durations of the r03
    trace are 3.05, 3.71 | 3.23, 3.07, 2.89, 2.88, 2.79, ... and 2.61 - 2.68 ms from the 15th launch on (clocks coming back up).  A long window
    reports the sustained rate, and a rocprofv3 --stats average of the same command (all 220 launches) agrees with it to ~1 %; with 2 + 40
    launches the ramp was a third of the window (0.79 instead of 0.80 - 0.81 of 8 TB/s)."""
    from mxfusion_amd import ops
    td = torch.float32 if dtype == 'float32' else torch.float64
    X = torch.rand(1, N, Q, device='cuda', dtype=td) * 6 - 3
    ls = torch.ones(1, Q, device='cuda', dtype=td)
    var = torch.ones(1, 1, device='cuda', dtype=td)
    out = torch.empty(1, N, N, device='cuda', dtype=td)
    for _ in range(20):
        ops.gram('rbf', X, None, ls, var, True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gram('rbf', X, None, ls, var, True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = (N * N + 2 * N * Q) * out.element_size()
    del out
    torch.cuda.empty_cache()
    # HBM bytes per launch from the PMC passes of exactly this launch (separate rocprofv3 --pmc runs, WRITE_SIZE * 1024 + 2 * FETCH_SIZE *
    # 1024 as MI355X_MICROARCH.md prescribes for gfx950): read from the NEWEST committed summary profiles/r<NN>_gram_pmc.json (produced by
    # profiles/run_profiles_r<NN>.sh + profiles/pmc_summary.py); only collected for the headline shape / dtype.  null if there is none.
    traffic, traffic_src = _pmc_traffic('gram_pmc') if (N == 65536 and Q == 8 and dtype == 'float32') else (None, None)
    return {"bound": "hbm", "kernel": "gram_lean_kernel<%s,8,RBF,%s> N=%d Q=%d" % (dtype, "12 rows" if dtype == "float32" else "16 rows", N, Q), "achieved": nbytes / ms / 1e6, "peak": 8000.0,
            "unit": "GB/s", "frac": nbytes / ms / 1e6 / 8000.0, "traffic": traffic, "traffic_source": traffic_src, "ms_per_launch": ms,
            "algorithmic_bytes": nbytes, "write_only_pattern_ceiling_GBps": 6930.0,
            "ceiling_note": "fastest pure-store pattern measured on this part (tests/probes/gram_variants.hip, one row per workgroup): "
                            "6.93 TB/s; hipMemsetAsync 6.15 TB/s"}


def mfma_roofline(M, SB, dtype, reps=12):
    """The dominant MFMA kernel of the step: T = H0 Kuf_all  (M x M x SB).  float32: gemm_split_kernel -- f32 operands as two scaled f16
    terms, three f16 MFMA products, f32 accumulate (f32-equivalent accuracy); its peak is the dense f16 MFMA peak / 3.
    float64: gemm_kernel on v_mfma_f64_16x16x4_f64.
    12 timed launches behind 1 untimed: behind the idle gap of the operand set-up the launch time ramps down (r03 trace: 11.9 | 11.5, 11.2, 11.0 ms)
    as the clocks come back up; three launches reported the ramp."""
    from mxfusion_amd import ops
    fl = 2.0 * M * M * SB
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dtype == 'float32':
        A = torch.randn(M, M, device='cuda')
        Bt = torch.rand(M, SB, device='cuda')                 # the K-major operand AS STORED: the planes of Kuf that Psi2 reads
        pa, pb = ops.f16x2_split(A), ops.f16x2_split(Bt)
        del Bt
        w = torch.randn(M, device='cuda')
        out = torch.empty(M, SB, device='cuda')
        # as the training step runs it (r06): second operand K-major through LDS-DMA + transposing LDS reads, T written in 16-column blocks
        # (what the reverse pass reads), the row U = w^T Kuf formed from the same fragments
        run = lambda: ops.gemm_f16x2_planes_kmajor(pa, pb, M, SB, M, out=out, blocked=True, w=w)
        name, peak, extra = "gemm_f16x2_bt_kernel<true> (f32 = 2 scaled f16 terms, 3 MFMA products; K-major second operand via ds_read_b64_tr_b16; C in 16-column blocks; fused U row) %dx%dx%d" % (M, SB, M), 2500.0 / 3.0, \
            {"peak_note": "dense f16 MFMA peak 2500 TFLOP/s / 3 products; the f32 MFMA peak is 157.3", "f32_mfma_peak": 157.3}
    else:
        A = torch.randn(1, M, M, device='cuda', dtype=torch.float64)
        B = torch.randn(1, M, SB, device='cuda', dtype=torch.float64)
        out = torch.empty(1, M, SB, device='cuda', dtype=torch.float64)
        run = lambda: ops.gemm(A, B, out=out)
        name, peak, extra = "gemm_kernel<float64,NN> %dx%dx%d" % (M, SB, M), 78.6, {}
    run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    torch.cuda.empty_cache()
    traffic, traffic_src = _pmc_traffic('gemm_tbt_pmc') if (dtype == 'float32' and M == 1024 and SB == 2097152) else (None, None)
    r = {"bound": "mfma", "kernel": name, "achieved": fl / ms / 1e9, "peak": peak, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / peak,
         "traffic": traffic, "traffic_source": traffic_src, "ms_per_launch": ms, "algorithmic_flops": fl}
    r.update(extra)
    return r


def mfma_roofline_psi2(M, SB, reps=12):
    """The second MFMA kernel of the float32 step: Psi2 = Kuf Kuf^T (M x M x SB, lower blocks only) on gemm_f16x2_wide_kernel (128 x 256
    tiles, B fragments straight from global memory).  Flops counted for the lower triangle incl. the diagonal blocks it computes in full."""
    from mxfusion_amd import ops
    C = torch.rand(M, SB, device='cuda')
    pc = ops.f16x2_split(C)
    del C
    psi = torch.zeros(M, M, device='cuda')
    run = lambda: ops.gemm_f16x2_planes(pc, pc, M, M, SB, out=psi, lower_only=True)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    torch.cuda.empty_cache()
    fl = 2.0 * (M * (M + 1) / 2.0) * SB
    traffic, traffic_src = _pmc_traffic('gemm_psi2_pmc') if (M == 1024 and SB == 2097152) else (None, None)
    return {"bound": "mfma", "kernel": "gemm_f16x2_wide_kernel_256lo (lower blocks, split-K; waves above the diagonal idle) %dx%dx%d" % (M, M, SB), "achieved": fl / ms / 1e9,
            "peak": 2500.0 / 3.0, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / (2500.0 / 3.0), "traffic": traffic, "traffic_source": traffic_src,
            "ms_per_launch": ms, "algorithmic_flops": fl}


def cpu_baseline(N, Q, M, S, X, Y, Z):
    """The oracle (float64 CPU restatement of the reference op sequence incl. autograd backward + MXNet-Adam) on this
    box's host cores, on a bounded sample: one SVI step at S=1, B=8192 rows; the full step is S * N/B such sub-steps
    (linear in S and B: BASELINE.md 3), so steps/s = 1 / (t_sub * S * N/B).  The BLAS thread count is calibrated on a
    small problem first (all 256 hardware threads is far from the fastest setting) and reported as `cores`."""
    from oracle import gp_oracle as O
    T = O.T
    kern = O.RBF(Q, ARD=True)
    rng = np.random.default_rng(1)

    def substep(Bs, reps):
        raw = {'qX_mean': T(X[:Bs]), 'qX_var': O.inv_softplus(T(np.full((Bs, Q), 1e-2))), 'noise_var': O.inv_softplus(T([0.01])),
               'lengthscale': O.inv_softplus(T(np.ones(Q))), 'variance': O.inv_softplus(T([1.0])), 'qU_mean': T(np.zeros((M, 1))),
               'qU_cov_W': T(np.zeros((M, M))), 'qU_cov_diag': O.inv_softplus(T(np.ones(M))), 'Z': T(Z)}
        opt = O.MXNetAdam(1e-3)
        times = []
        for it in range(reps):
            eps = T(rng.standard_normal((1, Bs, Q)))
            t0 = time.perf_counter()
            lv = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
            loss = O.svi_latent_svgp_loss(kern, T(Y[:Bs]), lv['Z'], lv, eps, jitter=1e-6, log_pdf_scaling=N / Bs)
            loss.backward()
            raw = opt.step({k: v.detach() for k, v in lv.items()}, {k: v.grad for k, v in lv.items()})
            times.append(time.perf_counter() - t0)
        return min(times)

    hw = os.cpu_count() or 1
    best_t, best_th = None, 1
    for th in sorted({min(hw, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(th)
        t = substep(min(1024, N), 2)
        if best_t is None or t < best_t:
            best_t, best_th = t, th
    torch.set_num_threads(best_th)
    Bs = min(8192, N)
    t_sub = substep(Bs, 3)
    full = t_sub * S * (N / Bs)
    return {"value": 1.0 / full, "unit": "ELBO-steps/sec", "cores": best_th, "kind": "port",
            "sample": "oracle SVI sub-step (fwd+autograd bwd+Adam, float64, torch-CPU BLAS, %d threads = fastest of 8..128 on this "
                      "%d-thread host) at S=1, B=%d of N=%d, M=%d: %.3f s; full step = S*N/B = %d sub-steps (linear extrapolation)"
                      % (best_th, hw, Bs, N, M, t_sub, int(S * N / Bs)),
            "seconds_per_substep": t_sub}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--dtype', default='float32', choices=['float32', 'float64'])
    ap.add_argument('--N', type=int, default=65536)
    ap.add_argument('--Q', type=int, default=8)
    ap.add_argument('--M', type=int, default=1024)
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--workload', default='svgp', choices=['svgp', 'gp', 'deepgp', 'pilco'], help="'svgp' = the headline (configs[2]); 'gp' = configs[1] (exact GP); 'deepgp' = configs[4]")
    ap.add_argument('--minibatch', type=int, default=0, help="svgp workload: minibatch size (0 = full batch = configs[2]; 8192 = configs[3])")
    ap.add_argument('--shard', default='samples', choices=['samples', 'rows'], help="--minibatch: what the ranks divide -- the MC samples of the "
                    "uncertain-input model (configs[3]) or, 'rows', the rows of every minibatch of the observed-input MAP model (no sample axis)")
    ap.add_argument('--proxy-world', type=int, default=0, help="--minibatch --shard rows on ONE GPU: evaluate the share of one rank of a W-GPU run "
                    "(B / W rows, KL weight 1 / W, no collective)")
    ap.add_argument('--hidden', type=int, default=2, help='hidden-layer width of the deep GP workload')
    ap.add_argument('--horizon', type=int, default=100, help='time steps of the PILCO rollout workload')
    ap.add_argument('--graph', type=int, default=0, help='1: capture forward + reverse pass of a step into a hipGraph after two eager steps')
    ap.add_argument('--force-dist', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend of an N > 1 run: 'nccl' (= RCCL over xGMI, one rank per GPU; the default and what a measurement uses) or "
                         "'gloo' on device tensors (the same product loops and HIP kernels, collectives staged through the host)")
    ap.add_argument('--same-device', action='store_true',
                    help='with --backend gloo: every rank runs on cuda:0 -- a software check of the multi-process path on a one-GPU box, not a measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the roofline micro-measurements and the f64 cross-check')
    ap.add_argument('--f32-form', default='auto', choices=['auto', 'explicit', 'whitened', 'float64'],
                    help="float32 SVGP calls: 'auto' = the per-module guard picks (explicit-inverse up to cond 1e3, whitened up to 1e6, float64 above); "
                         "the others force one form for a measurement")
    ap.add_argument('--trained-like', action='store_true', help='svgp workload: time the step at the trained-like parameters (length-scale 2.2, '
                    'noise 0.02, non-trivial q(u)) instead of the initial point')
    ap.add_argument('--no-f32-guard', action='store_true', help='disable the automatic float64 fallback of the float32 SVGP step above cond_1(Kuu) 3e3 '
                                                                    '(DESIGN.md section 5): raw float32 timing of an ill-conditioned model')
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to the C stdout at start-up (seen after the JSON when stdout
    # is a pipe), so file descriptor 1 is pointed at stderr for the rest of the run and the JSON goes to the saved descriptor
    sys.stdout.flush()
    _json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        with os.fdopen(os.dup(_json_fd), 'w') as f:
            f.write(json.dumps(obj) + '\n')

    # --gpus N without a launcher around us: start the N ranks ourselves (one process per GPU through torch.distributed.run, exactly the
    # command line the driver uses) and let rank 0's JSON line through; never a silent one-rank run that reports n_gpus = 1.
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import socket
        import subprocess
        ndev = torch.cuda.device_count()
        if args.same_device and args.backend != 'gloo':
            raise SystemExit('bench.py --same-device needs --backend gloo (RCCL refuses two ranks on one GPU)')
        if ndev < (1 if args.same_device else args.gpus):
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible to this process' % (args.gpus, ndev))
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.dup2(_json_fd, 1)                       # the children inherit the real stdout (rank 0 writes the one JSON line to it)
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.no_f32_guard:
        from mxfusion_amd.modules.gp_modules._fused import Float32Guard
        Float32Guard.enabled = False
    if args.f32_form != 'auto':
        from mxfusion_amd.modules.gp_modules._fused import Float32Guard
        Float32Guard.force = {'explicit': Float32Guard.EXPLICIT, 'whitened': Float32Guard.WHITENED, 'float64': Float32Guard.F64}[args.f32_form]

    def guard_report():
        """float32 validity of what was timed: the largest cond_1(Kuu + jitter I) the training calls published and the level every SVGP
        module ended on (explicit-inverse float32 up to 1e3, whitened float32 up to 1e6, float64 above: DESIGN.md section 5)."""
        from mxfusion_amd.modules.gp_modules._fused import Float32Guard
        torch.cuda.synchronize()
        return Float32Guard.report(torch.device('cuda', torch.cuda.current_device()))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != max(1, args.gpus) and not args.force_dist:
        raise SystemExit('bench.py: launched with WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    local_rank = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.device_count() < min(world, 1 + local_rank):
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (rank, torch.cuda.device_count()))
    distributed = world > 1 or args.force_dist      # --force-dist: run the RCCL code path (init, broadcast, all-reduce, barriers) even with one rank
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.force_dist and world == 1:          # stand-alone run of the RCCL code path: supply what the launcher would
            for k, v in (('RANK', '0'), ('WORLD_SIZE', '1'), ('LOCAL_RANK', '0'), ('MASTER_PORT', '29517')):
                os.environ.setdefault(k, v)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
    _RANK_TIMES['backend'], _RANK_TIMES['same_device'] = args.backend, bool(args.same_device)
    if args.samples % world:
        raise SystemExit('--samples must be divisible by the number of GPUs')
    S_local = args.samples // world
    torch.manual_seed(1234 + rank)

    if args.workload == 'pilco':       # GP-predict-in-a-loop latency (single GPU; the rollout is a serial chain)
        out = bench_pilco(1000 if args.N == 65536 else args.N, 64 if args.samples == 32 else args.samples, args.horizon,
                          'float64' if args.dtype == 'float64' else 'float32', args.steps, args.warmup, args.graph, not args.no_cpu_baseline)
        if rank == 0:
            emit(out)
        return
    if args.workload == 'gp':          # secondary workload: BASELINE.json configs[1] (exact GP, N=8192 D=8; does not shard: replicas only)
        N, Q = (8192 if args.N == 65536 else args.N), args.Q
        X, Y, _ = synth(N, Q, 1)
        m, infr, loop = build_gp(N, Q, args.dtype, X, Y)
        td = torch.float32 if args.dtype == 'float32' else torch.float64
        data = [torch.as_tensor(X, dtype=td).cuda(), torch.as_tensor(Y, dtype=td).cuda()]
        dt, last_loss = time_steps_multi(infr, loop, data, args.steps, args.warmup, args.lr, False)
        out = {"metric": "MAP-steps/sec, exact GPRegression (BASELINE.json configs[1])", "value": world * args.steps / dt, "unit": "MAP-steps/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == 'float32' else "f64", "data": "synthetic",
               "config": {"workload": "GPRegression RBF-ARD N=%d Q=%d: Gram + potrf + trsm + closed-form reverse mode + Adam per step; "
                                      "dense N x N Cholesky does not shard (replicas only)" % (N, Q)},
               "last_loss": last_loss, "potrf_info": int(m.Y.factor.gp_log_pdf._last_info.abs().sum()),
               # the float64 MFMA work of a step as the library performs it: potrf N^3/3 + trtri N^3/3 + the lower triangle of
               # K^-1 = L^-T L^-1 N^3/3 (the reverse mode needs K^-1; the solves against Y are N^2) = N^3 -- a roofline numerator for the
               # 78.6 TFLOP/s float64 matrix peak (the dense-operator count 3 N^3 of r02 exceeded that peak: the kernels skip the
               # triangular halves it counted)
               "algorithmic_flops_per_step": float(N) ** 3}
        if args.dtype == 'float64':
            ach = float(N) ** 3 / (dt / args.steps) / 1e12
            out["roofline_f64_mfma"] = {"bound": "mfma", "kernel": "potrf + trtri + lower L^-T L^-1 (float64 v_mfma_f64_16x16x4_f64)", "achieved": ach,
                                        "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6,
                                        "note": "whole step (incl. Gram, reverse pass, Adam) in the denominator"}
        if rank == 0 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_gp(N, Q, X, Y)
        if rank == 0:
            emit(out)
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if args.workload == 'deepgp':      # secondary workload: BASELINE.json configs[4] (2-layer SVGP deep GP, Matern52+RBF, N=131072 D=16 M=512/layer)
        N, Q, M, Dh = (131072 if args.N == 65536 else args.N), (16 if args.Q == 8 else args.Q), (512 if args.M == 1024 else args.M), args.hidden
        X, Y, _ = synth(N, Q, M)
        infr, loop = build_deepgp(N, Q, M, Dh, S_local, args.dtype, X, Y, distributed)
        td = torch.float32 if args.dtype == 'float32' else torch.float64
        data = [torch.as_tensor(X, dtype=td).cuda(), torch.as_tensor(Y, dtype=td).cuda()]
        dt, last_loss = time_steps_multi(infr, loop, data, args.steps, args.warmup, args.lr, distributed)
        rep = dist_report(world, args.steps, infr.params.flat.numel(), td)          # collective: every rank
        if rank == 0:
            emit(dict({
                "metric": "ELBO-steps/sec, 2-layer SVGP deep GP (BASELINE.json configs[4])", "value": args.steps / dt, "unit": "ELBO-steps/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.dtype == 'float32' else "f64", "data": "synthetic",
                "config": {"workload": "deep GP: SVGPRegression(Matern52+RBF, Q=%d) -> H (N x %d, mean-field q(H)) -> SVGPRegression(RBF-ARD), N=%d, "
                                       "M=%d per layer, %d MC samples" % (Q, Dh, N, M, args.samples), "samples_per_gpu": S_local},
                "last_loss": last_loss}, **rep, **guard_report()))
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    N, Q, M = args.N, args.Q, args.M
    X, Y, Z = synth(N, Q, M)
    if args.minibatch and args.shard == 'rows':      # row-sharded data parallelism for a model without a sample axis (second axis)
        B = args.minibatch
        td = torch.float32 if args.dtype == 'float32' else torch.float64
        m, infr, loop = build_minibatch_rows(N, Q, M, B, args.dtype, Z, proxy_world=args.proxy_world)
        Xd, Yd = torch.as_tensor(X, dtype=td).cuda(), torch.as_tensor(Y, dtype=td).cuda()
        dt, last_loss = time_minibatch_steps(infr, loop, Xd, Yd, B, args.steps, args.warmup, args.lr, distributed)
        run_s = time_minibatch_run(infr, {'X': Xd, 'Y': Yd}, max(2, args.steps // (N // B)), args.lr, distributed)
        rep = dist_report(world, args.steps, infr.params.flat.numel(), td)
        w = args.proxy_world if args.proxy_world > 1 else world
        if rank == 0:
            emit(dict({"metric": "ELBO-steps/sec (minibatch steps), SVGP MAP on observed inputs N=65k D=8 M=1024 minibatch=%d, rows sharded" % B,
                  "value": args.steps / dt, "unit": "ELBO-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": dt / args.steps * 1e3, "ms_per_step_through_run": run_s * 1e3, "higher_is_better": True, "scaling": "strong",
                  "vs_baseline": None, "dtype": "f32" if args.dtype == 'float32' else "f64", "data": "synthetic",
                  "config": {"workload": "SVGPRegression RBF-ARD N=%d Q=%d M=%d, MAP on observed inputs (no sample axis: the reference's svgp_regression "
                                         "notebook), minibatch %d (rv_scaling %g), step = minibatch ELBO + reverse mode + grad all-reduce + Adam"
                                         % (N, Q, M, B, N / B),
                             "rows_per_rank": B // w, "kl_weight": 1.0 / w,
                             "parallelism": ("one-GPU proxy of one rank of a %d-GPU run (no collective)" % w) if args.proxy_world > 1 else
                                            "rows of each minibatch sharded x%d, 1 RCCL all-reduce of the flat gradient + the loss per step" % world},
                  "last_loss": last_loss, "potrf_info": int(m.Y.factor.svgp_log_pdf._last_info.abs().sum())}, **rep, **guard_report()))
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if args.minibatch:          # BASELINE.json configs[3]: minibatch x sample sharding
        B = args.minibatch
        td = torch.float32 if args.dtype == 'float32' else torch.float64
        m, infr, loop = build_minibatch(N, Q, M, B, S_local, args.dtype, Z)
        Xd_, Yd_ = torch.as_tensor(X, dtype=td).cuda(), torch.as_tensor(Y, dtype=td).cuda()
        dt, last_loss = time_minibatch_steps(infr, loop, Xd_, Yd_, B, args.steps, args.warmup, args.lr, distributed)
        # the same steps through GradBasedInference.run (what a user calls): r04's loop synchronised the device on every minibatch
        run_s = time_minibatch_run(infr, {'Xobs': Xd_, 'Y': Yd_}, max(2, args.steps // (N // B)), args.lr, distributed)
        rep = dist_report(world, args.steps, infr.params.flat.numel(), td)          # collective: every rank
        rep["ms_per_step_through_run"] = run_s * 1e3
        if world == 1 and not args.no_extras and args.samples % 8 == 0 and args.samples >= 8:
            # one rank's share of an 8-GPU run of this config on THIS GPU (samples / 8, no collective): the projection the scaling curve will be
            # read against -- the sample-independent M x M chain is the floor of a rank's step
            del infr, m, loop
            torch.cuda.empty_cache()
            m8, infr8, loop8 = build_minibatch(N, Q, M, B, args.samples // 8, args.dtype, Z)
            dt8, _ = time_minibatch_steps(infr8, loop8, Xd_, Yd_, B, args.steps, args.warmup, args.lr, False)
            rep["per_rank_proxy"] = {"samples": args.samples // 8, "ms_per_step": dt8 / args.steps * 1e3,
                                     "projected_speedup_8gpu": dt / dt8, "note": "one GPU evaluating one rank's share; before the all-reduce"}
            m = m8
        if rank == 0:
            emit(dict({"metric": "ELBO-steps/sec (minibatch steps), SVGP N=65k D=8 M=1024 minibatch=%d (BASELINE.json configs[3])" % B,
                  "value": args.steps / dt, "unit": "ELBO-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                  "dtype": "f32" if args.dtype == 'float32' else "f64", "data": "synthetic",
                  "config": {"workload": "SVGPRegression RBF-ARD N=%d Q=%d M=%d, uncertain inputs X ~ N(Xobs, 1e-2), q(X) = N(Xobs, v), minibatch %d "
                                         "(rv_scaling %g), %d MC samples, step = minibatch ELBO + reverse mode + grad all-reduce + Adam"
                                         % (N, Q, M, B, N / B, args.samples),
                             "samples_per_gpu": S_local, "parallelism": "mc-samples sharded x%d, 1 RCCL all-reduce of the flat gradient/step" % world},
                  "last_loss": last_loss, "potrf_info": int(m.Y.factor.svgp_log_pdf._last_info.abs().sum())}, **rep, **guard_report()))
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    m, q, infr, loop, qX = build(N, Q, M, S_local, args.dtype, X, Y, Z, distributed, use_graph=args.graph)
    td = torch.float32 if args.dtype == 'float32' else torch.float64
    if args.trained_like:
        set_trained_like(m, infr, Q, M, td)
    Yd = torch.as_tensor(Y, dtype=td).cuda()
    dt, last_loss = time_steps(infr, loop, Yd, args.steps, args.warmup, args.lr, distributed)
    rep = dist_report(world, args.steps, infr.params.flat.numel(), td)              # collective: every rank

    out = {
        "metric": "ELBO-steps/sec + RBF Gram GB/s, SVGP N=65k D=8 M=1024, 1->8 MI355X",
        "value": args.steps / dt, "unit": "ELBO-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if args.dtype == 'float32' else "f64", "data": "synthetic",
        "config": {"workload": "SVGPRegression RBF-ARD N=%d Q=%d M=%d, %d MC samples (latent-input model), "
                               "StochasticVariationalInference step = draw + ELBO + reverse mode + grad all-reduce + Adam" % (N, Q, M, args.samples),
                   "samples_per_gpu": S_local, "parallelism": "mc-samples sharded x%d, 1 RCCL all-reduce of the flat gradient/step" % world,
                   "core_precision": "float64 (M x M factorisations), streaming %s" % args.dtype},
        "last_loss": last_loss,
        # LAPACK-style info of the last step's Cholesky factorisations (0 = every pivot positive)
        "potrf_info": int(m.Y.factor.svgp_log_pdf._last_info.abs().sum()),
    }
    out.update(rep)
    out.update(guard_report())
    out["parameters"] = "trained-like (length-scale 2.2, noise 0.02, non-trivial q(u))" if args.trained_like else "initial point (length-scale 1, q(u) = N(0, I))"
    out["f32_form"] = args.f32_form
    if rank == 0 and world == 1 and not args.no_extras and args.dtype == 'float32':
        out.update(step_breakdown(infr, loop, Yd, args.lr, M, N * S_local))
    if rank == 0 and world == 1 and not args.no_extras and args.dtype == 'float32' and not args.trained_like and args.f32_form == 'auto':
        # the regime the step is used in: the same step at trained-like parameters (explicit form: cond ~ 1.4e3 < 3e3), and in the
        # whitened float32 form the guard selects above cond 1e3
        from mxfusion_amd.modules.gp_modules._fused import Float32Guard
        nst = max(3, args.steps // 2)
        set_trained_like(m, infr, Q, M, td)
        dtt, _ = time_steps(infr, loop, Yd, nst, 2, args.lr, False)
        out["ms_per_step_trained_like"] = dtt / nst * 1e3
        out["value_trained_like"] = nst / dtt           # ELBO-steps/sec in the regime training spends its time in (whitened float32 form)
        out["trained_like"] = dict(guard_report(), **step_breakdown(infr, loop, Yd, args.lr, M, N * S_local))
        Float32Guard.force = Float32Guard.WHITENED
        try:
            dtw, _ = time_steps(infr, loop, Yd, nst, 2, args.lr, False)
            out["ms_per_step_whitened"] = dtw / nst * 1e3
            out["whitened"] = step_breakdown(infr, loop, Yd, args.lr, M, N * S_local)
        finally:
            Float32Guard.force = None
    if rank == 0 and world == 1 and not args.no_extras:
        del infr, m, q
        torch.cuda.empty_cache()
        if args.samples % 8 == 0 and args.samples >= 8 and not args.trained_like:
            # one rank's share of an 8-GPU run (samples / 8) on THIS GPU, no collective: what the 1 -> 8 curve will be read against
            m8, q8, infr8, loop8, _ = build(N, Q, M, args.samples // 8, args.dtype, X, Y, Z, False)
            dt8, _ = time_steps(infr8, loop8, Yd, max(args.steps, 10), max(args.warmup, 3), args.lr, False)
            out["per_rank_proxy"] = {"samples": args.samples // 8, "ms_per_step": dt8 / max(args.steps, 10) * 1e3,
                                     "projected_speedup_8gpu": (dt / args.steps) / (dt8 / max(args.steps, 10)),
                                     "note": "one GPU evaluating one rank's share; before the all-reduce (allreduce of the 8.4 MB flat gradient: ~0.1 ms on a ring over xGMI)"}
            del infr8, m8, q8
            torch.cuda.empty_cache()
        out["other_boxes"] = box_spread()
        out["roofline"] = gram_roofline(N, Q, args.dtype)
        out["roofline_mfma"] = mfma_roofline(M, N * S_local, args.dtype)
        if args.dtype == 'float32':
            out["roofline_mfma_psi2"] = mfma_roofline_psi2(M, N * S_local)
        other = 'float64' if args.dtype == 'float32' else 'float32'
        out["roofline_" + ("f64" if other == 'float64' else "f32")] = gram_roofline(N, Q, other)
        # the same step in the other precision (f64 = the parity precision of the reference's tests), plus the
        # agreement of the two ELBO values on identical parameters and noise
        torch.manual_seed(1234)
        m2, q2, infr2, loop2, _ = build(N, Q, M, S_local, other, X, Y, Z, False)
        Y2 = torch.as_tensor(Y, dtype=torch.float64 if other == 'float64' else torch.float32).cuda()
        dt2, loss2 = time_steps(infr2, loop2, Y2, max(2, args.steps // 3), 1, args.lr, False)
        out["other_dtype"] = {"dtype": "f64" if other == 'float64' else "f32", "value": max(2, args.steps // 3) / dt2, "unit": "ELBO-steps/sec"}
        del infr2, m2, q2
        torch.cuda.empty_cache()
        from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
        eps64 = torch.randn(4, N, Q, dtype=torch.float64, device='cuda', generator=torch.Generator(device='cuda').manual_seed(7))
        for key, trained in (("elbo_f32_vs_f64_rel", False), ("elbo_f32_vs_f64_rel_trained_like", True)):
            # the initial parameters of the timed run (length-scale 1: Kuu ~ I), and a trained-like set -- length-scale 2.2 (where a 300-step
            # optimisation of this model ends, tests/probes/train_probe.py; cond(Kuu + 1e-6 I) ~ 1.4e3), noise 0.02, a non-trivial q(u)
            vals = {}
            for dname in ('float32', 'float64'):
                tdd = torch.float32 if dname == 'float32' else torch.float64
                mm, qq, ii, ll, qx = build(N, Q, M, 4, dname, X, Y, Z, False)
                if trained:
                    set_trained_like(mm, ii, Q, M, tdd)
                qx._rand_gen = MockRandomGenerator(eps64.to(tdd))     # identical injected noise in both precisions
                ex = ii.create_executor()
                # evaluated WITH the reverse mode requested: that is the call the timed step makes (float32: Grams as split planes, both big
                # GEMMs on the f16 matrix pipe); the forward-only call would run the plain f32-MFMA kernels instead
                vals[dname] = float(ex(torch.as_tensor(Y, dtype=tdd).cuda())[0].detach())
                del mm, qq, ii, ex
                torch.cuda.empty_cache()
            out[key] = abs(vals['float32'] - vals['float64']) / abs(vals['float64'])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, Q, M, args.samples, X, Y, Z)
    if rank == 0:
        emit(out)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
