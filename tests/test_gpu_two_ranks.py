"""SURVEY 8(e) parity on real RCCL: world_size 2, one process per GPU -- gradient of the sharded run == gradient of the one-process run to
1e-10 (float64, identical injected noise), through BOTH exchanges (torch.distributed 'nccl' inside DistributedBatchInferenceLoop, and the
C ABI's mxf_comm_init / mxf_allreduce_sum / mxf_bcast).  Skipped where fewer than two GPUs are visible (the gloo tests cover the logic
there); tests/two_rank_worker.py is the per-rank program."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two visible GPUs')
def test_two_ranks_gradient_equals_one_rank_gradient_on_rccl():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(root, 'tests', 'two_rank_worker.py')], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and 'two-rank parity ok' in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def _launch(nproc, script_args, timeout=1500):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
                           '127.0.0.1', '--master-port', str(port)] + script_args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('world', [2, 4])
def test_ranks_sharing_one_gpu_train_like_one_process(world):
    """Row (e) with the REAL kernels in more than one process, on the one GPU a test box has: `world` ranks on cuda:0, 'gloo' collectives on
    device tensors.  Sample-sharded (eager and hipGraph-replayed) and row-sharded (batch and minibatch) product loops, float64, identical
    injected noise: parameters after 3-4 Adam steps equal the single-process loops to 1e-10, one all-reduce per step (the worker asserts)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _launch(world, [os.path.join(root, 'tests', 'two_rank_worker.py'), '--backend', 'gloo', '--same-device'])
    assert out.returncode == 0 and 'multi-rank parity ok' in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_two_ranks_on_one_gpu_through_gloo():
    """`bench.py --gpus 2 --backend gloo --same-device`: the bench's own launcher, per-rank handles, barrier + max-over-ranks timing and the
    one-collective exchange with two real ranks (reduced size: this checks the path, it is not a measurement)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--same-device', '--N', '8192',
                          '--M', '256', '--samples', '4', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-extras'], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['backend'] == 'gloo' and line['same_device'] is True
    assert line['collectives_per_step'] == 1 and len(line['per_rank_ms_per_step']) == 2
    assert line['potrf_info'] == 0 and line['value'] > 0


def test_bench_refuses_more_gpus_than_are_visible():
    """`bench.py --gpus N` without a launcher spawns its N ranks itself; with fewer than N devices it must fail loudly, never run one rank
    and report n_gpus = 1."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and 'visible' in (out.stderr + out.stdout)
    assert '"n_gpus"' not in out.stdout
