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
