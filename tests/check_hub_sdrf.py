"""GPU-box checker (tests/test_checkers_gpu.py runs one of the two cases; uses the oracle): a few SDRF iterations on a graph with two adjacent hubs of
8,500-9,000 neighbours, through the public entry point, against the C oracle.  Last run: identical (oracle 15 s, GPU 0.02-0.26 s)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data
from oracle import c_oracle
from rewiring.sdrf_no_cuda import sdrf_no_cuda


def run(cases=((float('inf'), 3), (30.0, 3))):
    n = 30000
    rng = np.random.Generator(np.random.PCG64(5))
    src, dst = [], []
    for hub, d in ((0, 9000), (1, 8500)):
        leaves = rng.choice(np.arange(10, n), size=d, replace=False)
        src += [hub] * d; dst += leaves.tolist()
    src.append(0); dst.append(1)
    ex = rng.integers(10, n, size=(2, 40000))
    src += ex[0].tolist(); dst += ex[1].tolist()
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    ok = True
    for tau, loops in cases:
        np.random.seed(4)
        t = time.time(); want = c_oracle.sdrf(ei, n, 'bfc', loops, True, 0.5, tau, nthreads=16); t1 = time.time() - t
        np.random.seed(4)
        t = time.time(); got = sdrf_no_cuda(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', loops, True, 0.5, tau).edge_index.numpy(); t2 = time.time() - t
        same = np.array_equal(got, want)
        ok = ok and same
        print(tau, 'identical' if same else 'DIFFERENT', f'oracle {t1:.1f}s gpu {t2:.2f}s', flush=True)
    return ok


if __name__ == '__main__':
    sys.exit(0 if run() else 1)
