"""Randomised parity sweep of the bfc_cuda compatibility mode on the GPU box (tests/test_checkers_gpu.py runs it with a small fixed budget): random directed and
undirected graphs, the dense float32 curvature / post-delta kernels and whole sdrf_cuda_bfc runs against
oracle/bfc_cuda_oracle.py (which is pinned to the reference's own outputs).  Usage: SECONDS_BUDGET=120 python tests/fuzz_bfc_cuda.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from curvature.bfc_cuda import balanced_forman_curvature, balanced_forman_post_delta
from dcr.data import Data
from oracle import bfc_cuda_oracle as bo
from rewiring.sdrf_cuda_bfc import sdrf_cuda_bfc



def run(seed=1, seconds=120.0, graphs=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    t_end = time.time() + seconds
    n_graphs = n_vals = n_runs = 0
    while time.time() < t_end and (graphs is None or n_graphs < graphs):
        n = int(rng.integers(4, 110))
        p = float(rng.uniform(0.03, 0.4))
        undirected = bool(rng.integers(0, 2))
        m = rng.random((n, n)) < p
        np.fill_diagonal(m, False)
        if undirected:
            m = m | m.T
        src, dst = np.nonzero(m)
        if src.size == 0:
            continue
        ei = np.stack([src, dst]).astype(np.int64)
        N = int(ei.max()) + 1
        A = np.zeros((N, N), dtype=np.float32)
        A[ei[0], ei[1]] = 1.0
        Ad = torch.from_numpy(A).cuda()
        C = balanced_forman_curvature(Ad, numerics='bfc_cuda').cpu().numpy()
        want = bo.balanced_forman_curvature(A)
        assert np.array_equal(C.view(np.uint32), want.view(np.uint32)), ('C', n, p, undirected)
        n_vals += int(A.sum())
        x, y = int(src[0]), int(dst[0])
        xn = list(np.nonzero(A[x])[0]) + [x]
        yn = list(np.nonzero(A[:, y])[0]) + [y]
        D = balanced_forman_post_delta(Ad, x, y, [int(t) for t in xn], [int(t) for t in yn], numerics='bfc_cuda').cpu().numpy()
        wantD = bo.balanced_forman_post_delta(A, x, y, xn, yn)
        assert np.array_equal(D.view(np.uint32), wantD.view(np.uint32)), ('D', n, p, undirected)
        loops = int(rng.integers(2, 14))
        tau = float('inf') if rng.random() < 0.25 else float(rng.uniform(1, 80))
        bound = float(rng.uniform(0.05, 1.0))
        rem = bool(rng.random() < 0.85)
        seed = int(rng.integers(1 << 30))
        ta, tb = [], []
        np.random.seed(seed)
        w = bo.sdrf_cuda_bfc(ei, n, loops, rem, bound, tau, undirected, trace=ta)
        np.random.seed(seed)
        g = sdrf_cuda_bfc(Data(edge_index=torch.from_numpy(ei), num_nodes=n), loops, rem, bound, tau, undirected, trace=tb,
                          numerics='bfc_cuda').edge_index.numpy()
        assert len(ta) == len(tb) and np.array_equal(w, g), ('sdrf', n, p, undirected, loops, tau, bound, rem, seed)
        for a, b in zip(ta, tb):
            assert a['argmin'] == b['argmin'] and a['improvements'] == b['improvements'] and a['choice'] == b['choice']
            assert [list(e) for e in a['events']] == [list(e) for e in b['events']]
        n_graphs += 1
        n_runs += 1
        if n_graphs % 25 == 0:
            print(f'{n_graphs} graphs, {n_vals} curvature values, {n_runs} SDRF runs: all identical', flush=True)
    print(f'DONE {n_graphs} graphs, {n_vals} curvature values, {n_runs} SDRF runs: all identical', flush=True)
    return n_graphs, n_vals, n_runs


if __name__ == '__main__':
    run(int(os.environ.get('SEED', 1)), float(os.environ.get('SECONDS_BUDGET', 120)))
