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
