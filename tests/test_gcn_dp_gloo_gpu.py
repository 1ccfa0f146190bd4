"""Two gloo ranks sharing the one GPU of the box: the data-parallel model (models/gcn_dp.py, sharding the model of
models/gcn.py:32-44) with the one-kernel first layer on each rank's own rows against the same model with the separate kernels.
The RCCL test next door needs as many GPUs as ranks and is skipped on a one-GPU box; this one runs there."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_gloo_ranks_on_one_gpu_one_kernel_first_layer():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_gloo_worker.py')
    procs = []
    for r in range(2):                 # fresh child processes, one per rank
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
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
