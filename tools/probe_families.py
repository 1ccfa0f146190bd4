"""Curvature pass on graphs of different shapes (timing only; parity on such families is tests/fuzz_parity.py)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
from dcr.graph import DcrGraph


def run(name, ei, n):
    G = DcrGraph(ei, n)
    G.curvature_pass('bfc')
    G.profile_reset()
    for _ in range(5):
        G.curvature_pass('bfc')
    ms, cnt = G.profile_read()
    E = G.number_of_edges()
    deg = np.bincount(ei[0], minlength=n)
    print(f'{name:34s} N={n:8d} E={E:9d} max deg {deg.max():6d}  pass {ms / cnt:8.3f} ms  {E / (ms / cnt) / 1e3:8.1f} M edges/s', flush=True)
    G.close()


rng = np.random.Generator(np.random.PCG64(3))
run('preferential attachment m=10', *synthetic.powerlaw_graph(100000, 10, seed=12345))
run('preferential attachment m=5', *synthetic.powerlaw_graph(200000, 5, seed=1))
run('preferential attachment m=20', *synthetic.powerlaw_graph(50000, 20, seed=2))
run('preferential attachment m=2', *synthetic.powerlaw_graph(500000, 2, seed=4))
n = 100000
ex = rng.integers(0, n, size=(2, 1000000))
run('uniform random, mean degree 20', synthetic.coalesced_edge_index(ex[0], ex[1], n), n)
run('grid 1000 x 1000', *synthetic.grid_graph(1000, 1000))
n = 200000
src = rng.integers(0, 20, n); dst = np.arange(n); ex = rng.integers(0, n, size=(2, 800000))
run('20 stars of 10k leaves + random', synthetic.coalesced_edge_index(np.concatenate([src, ex[0]]), np.concatenate([dst, ex[1]]), n), n)
n = 3000
ex = rng.integers(0, n, size=(2, 1200000))
run('dense random, mean degree ~640', synthetic.coalesced_edge_index(ex[0], ex[1], n), n)
