"""Randomised cross-engine sweep on the GPU box (round 5): medium-sized graphs — where the C oracle is too slow to sweep many — through
the two-hop kernels, the node-centric class kernels and the edge-by-edge kernels: the three must give the same bits (each is
pinned to the oracle on the small graphs of tests/fuzz_parity.py), and so must an incremental pass behind a few edits.
SECONDS_BUDGET, SEED from the environment."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
from dcr.graph import DcrGraph


def graph(rng):
    kind = int(rng.integers(0, 3))
    n = int(rng.integers(5000, 150000))
    if kind == 0:
        return synthetic.powerlaw_graph(n, int(rng.integers(1, 14)), seed=int(rng.integers(1 << 30)))
    if kind == 1:
        d = int(rng.integers(2, 24))
        ex = rng.integers(0, n, size=(2, n * d // 2))
        return synthetic.coalesced_edge_index(ex[0], ex[1], n), n
    n = min(n, 60000)   # a few hubs on top of a sparse random graph
    hubs = int(rng.integers(1, 6))
    src, dst = [], []
    for h in range(hubs):
        d = int(rng.integers(500, min(6000, n - hubs - 1)))
        src += [h] * d
        dst += rng.choice(np.arange(hubs, n), size=d, replace=False).tolist()
    ex = rng.integers(hubs, n, size=(2, 3 * n))
    return synthetic.coalesced_edge_index(np.concatenate([np.array(src), ex[0]]), np.concatenate([np.array(dst), ex[1]]), n), n


def one(ei, n, mode):
    os.environ.pop('DCR_NC_FINE_FULL', None)
    os.environ['DCR_PASS'] = 'h2' if mode == 'h2' else 'nc'
    if mode == 'edges':
        os.environ['DCR_NC_FINE_FULL'] = str(1 << 40)
    G = DcrGraph(ei, n)
    os.environ.pop('DCR_PASS')
    return G, G.curvature_all('bfc')


def run(seed=1, seconds=300.0, verbose=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    t_end = time.time() + seconds
    graphs = values = 0
    while time.time() < t_end:
        ei, n = graph(rng)
        E = ei.shape[1] // 2
        G, (eu, ev, ref) = one(ei, n, 'nc')
        for mode in ('h2', 'edges'):
            if mode == 'edges' and E > 250000:
                continue
            H, (hu, hv, hc) = one(ei, n, mode)
            bad = np.flatnonzero(hc.view(np.int64) != ref.view(np.int64))
            assert np.array_equal(hu, eu) and np.array_equal(hv, ev) and bad.size == 0, (mode, n, E, bad[:5], hc[bad[:5]], ref[bad[:5]])
            values += hc.shape[0]
            H.close()
        # a few edits among the first nodes (the hubs of every family here), an incremental pass, against a full one
        os.environ.pop('DCR_NC_FINE_FULL', None)
        for _ in range(3):
            a, b = int(rng.integers(0, min(n, 50))), int(rng.integers(0, n))
            if a != b:
                (G.remove_edge if G.has_edge(a, b) else G.add_edge)(a, b)
        G.curvature_pass('bfc', incremental=True)
        inc = G.curvature_read()[2].copy()
        G.curvature_pass('bfc')
        full = G.curvature_read()[2]
        assert np.array_equal(inc.view(np.int64), full.view(np.int64)), ('incremental', n, E)
        G.close()
        graphs += 1
        if verbose and graphs % 10 == 0:
            print(f'{graphs} graphs, {values} edge values: the engines agree', flush=True)
    print(f'DONE {graphs} graphs, {values} edge values: the engines agree, incremental passes equal full ones', flush=True)
    return graphs


if __name__ == '__main__':
    run(int(os.environ.get('SEED', 1)), float(os.environ.get('SECONDS_BUDGET', 300)))
