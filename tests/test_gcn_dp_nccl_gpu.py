"""Two, four and eight RCCL ranks, one per MI355X (each skipped where fewer GPUs are visible: always on a one-GPU box): the
captured data-parallel epoch against the eager one.
Every other N > 1 test runs on gloo; this is the one that exercises RCCL collectives inside a captured HIP graph with more
than one rank (models/gcn_dp.py::GraphedShardedEpoch; the model it shards is models/gcn.py:32-44)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world,chunked', [(2, False), (4, False), (8, False), (2, True), (8, True)])
def test_nccl_ranks_graphed_epoch_equals_eager(world, chunked):
    """``chunked``: the same with ``DCR_DP_CHUNKED_GATHER=1`` (the exchange as two asynchronous all-gathers inside the capture)."""
    import torch
    if torch.cuda.device_count() < world:  # (counting devices does not initialise the GPU in this process)
        pytest.skip(f'needs {world} GPUs')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_nccl_worker.py')
    procs = []
    for r in range(world):                 # fresh child processes, one per rank
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   DCR_DP_CHUNKED_GATHER='1' if chunked else '0')
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail('a rank did not finish within 600 s')
        outs.append((p.returncode, out))
    assert all(rc == 0 for rc, _ in outs), outs
