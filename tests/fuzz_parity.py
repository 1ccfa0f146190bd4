"""Randomised parity sweep on the GPU box: many small and medium graphs of different families, every curvature kind and
every pass engine, full pass + incremental pass + SDRF runs against the C oracle.  tests/test_checkers_gpu.py runs it with a
small fixed budget; the long version: SECONDS_BUDGET=240 python tests/fuzz_parity.py  (under tests/ because it uses the oracle)"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data
from dcr.graph import DcrGraph
from oracle import c_oracle
from rewiring.sdrf_no_cuda import sdrf_no_cuda

def random_graph(rng, hub_prob):
    kind = rng.integers(0, 6)
    if rng.random() < hub_prob:  # hubs joined to each other with degrees beyond the LDS tables (device-memory path)
        n = int(rng.integers(20000, 45000)); hubs = int(rng.integers(2, 4))
        src, dst = [], []
        for h in range(hubs):
            d = int(rng.integers(7000, min(18000, n - 100)))
            src += [h] * d; dst += rng.choice(np.arange(hubs, n), size=d, replace=False).tolist()
        for a in range(hubs):
            for b in range(a + 1, hubs):
                src.append(a); dst.append(b)
        ex = rng.integers(hubs, n, (2, int(rng.integers(0, 2 * n))))
        return synthetic.coalesced_edge_index(np.concatenate([np.array(src), ex[0]]),
                                              np.concatenate([np.array(dst), ex[1]]), n), n
    if kind == 0:
        n = int(rng.integers(5, 400)); p = float(rng.uniform(0.01, 0.3))
        return synthetic.erdos_renyi_graph(n, p, seed=int(rng.integers(1 << 30)))
    if kind == 1:
        n = int(rng.integers(20, 3000)); m = int(rng.integers(1, 12))
        return synthetic.powerlaw_graph(n, min(m, n - 1), seed=int(rng.integers(1 << 30)))
    if kind == 2:
        return synthetic.grid_graph(int(rng.integers(2, 30)), int(rng.integers(2, 30)))
    if kind == 3:  # a few big stars joined by random edges (hubs with many degree-1 neighbours)
        n = int(rng.integers(50, 6000)); hubs = int(rng.integers(1, 5))
        src = rng.integers(0, hubs, n); dst = np.arange(n)
        ex = rng.integers(0, n, (2, int(rng.integers(0, n))))
        return synthetic.coalesced_edge_index(np.concatenate([src, ex[0]]), np.concatenate([dst, ex[1]]), n), n
    if kind == 4:  # dense: every class of table in one small graph
        n = int(rng.integers(100, 420)); p = float(rng.uniform(0.3, 0.9))
        return synthetic.erdos_renyi_graph(n, p, seed=int(rng.integers(1 << 30)))
    n = int(rng.integers(3, 40))  # tiny
    ex = rng.integers(0, n, (2, int(rng.integers(1, 4 * n))))
    return synthetic.coalesced_edge_index(ex[0], ex[1], n), n



def run(seed=1, seconds=240.0, graphs=None, engines=('nc', 'edge', 'h2'), hub_prob=0.04, verbose=True):
    """Returns (graphs, edge values, SDRF runs) checked; raises AssertionError at the first difference."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t_end = time.time() + seconds
    n_graphs = n_edges_checked = n_sdrf = 0
    while time.time() < t_end and (graphs is None or n_graphs < graphs):
        ei, n = random_graph(rng, hub_prob)
        if ei.shape[1] == 0:
            continue
        n_graphs += 1
        C = c_oracle.CGraph(ei, n)
        for impl in engines:
            os.environ['DCR_PASS'] = impl
            G = DcrGraph(ei, n)
            os.environ.pop('DCR_PASS')
            for ct in ('bfc', '1d', 'augmented', 'haantjes'):
                eu, ev, cv = G.curvature_all(ct)
                ou, ov, oc = C.curv_all(ct, nthreads=8)
                bad = np.nonzero(cv != oc)[0]
                assert np.array_equal(eu, ou) and np.array_equal(ev, ov) and bad.size == 0, \
                    (impl, ct, n, ei.shape, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:5]])
                n_edges_checked += cv.shape[0]
        # random edits + incremental passes
        G.curvature_pass('bfc')
        for step in range(12):
            u, v = (int(t) for t in rng.integers(0, n, 2))
            if u == v:
                continue
            if C.has_edge(u, v):
                G.remove_edge(u, v); C.remove_edge(u, v)
            else:
                G.add_edge(u, v); C.add_edge(u, v)
            if step % 4 == 3:
                G.curvature_pass('bfc', incremental=True)
                assert np.array_equal(G.curvature_read()[2], C.curv_all('bfc', nthreads=8)[2]), ('incremental', n, step)
        # a short SDRF run through the public entry point
        if ei.shape[1] >= 4 and n <= 1500:
            ct = ('bfc', 'augmented', 'haantjes', '1d')[int(rng.integers(0, 4))]
            tau = float(rng.choice([float('inf'), 5.0, 50.0, 163.0]))
            bound = float(rng.choice([0.0, 0.5, 0.95, 3.0]))
            seed = int(rng.integers(1 << 20)); loops = int(rng.integers(1, 25)); rem = bool(rng.integers(0, 2))
            np.random.seed(seed)
            try:
                want = c_oracle.sdrf(ei, n, ct, loops, rem, bound, tau, nthreads=8); err_w = None
            except ValueError as e:
                want, err_w = None, str(e)
            np.random.seed(seed)
            try:
                got = sdrf_no_cuda(Data(edge_index=torch.from_numpy(ei), num_nodes=n), ct, loops, rem, bound, tau,
                                   incremental=bool(rng.integers(0, 2))).edge_index.numpy(); err_g = None
            except ValueError as e:
                got, err_g = None, str(e)
            assert (err_w is None) == (err_g is None), (ct, tau, bound, seed, err_w, err_g)
            if want is not None:
                assert np.array_equal(got, want), ('sdrf', ct, tau, bound, seed, loops, rem, n)
            n_sdrf += 1
        if verbose and n_graphs % 25 == 0:
            print(f'{n_graphs} graphs, {n_edges_checked} edge values, {n_sdrf} SDRF runs: all identical', flush=True)
    if verbose:
        print(f'DONE {n_graphs} graphs, {n_edges_checked} edge values, {n_sdrf} SDRF runs: all identical', flush=True)
    return n_graphs, n_edges_checked, n_sdrf


if __name__ == '__main__':
    run(seed=int(os.environ.get('SEED', 1)), seconds=float(os.environ.get('SECONDS_BUDGET', 240)),
        hub_prob=float(os.environ.get('HUB_PROB', 0.04)))
