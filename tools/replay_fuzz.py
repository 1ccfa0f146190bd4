"""Replays the graph sequence of tests/fuzz_parity.py for one seed without the GPU work, then runs the engines' full passes on the
graphs FROM..TO only and reports the first error or difference (how the rare failures of the long sweeps are pinned down).
SEED, FROM, TO, HUB_PROB from the environment."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd'), os.path.join(REPO, 'tests')]
import numpy as np
import fuzz_parity as F
from dcr.graph import DcrGraph
from oracle import c_oracle
seed, lo, hi = int(os.environ.get('SEED', 1)), int(os.environ.get('FROM', 1)), int(os.environ.get('TO', 10))
hub_prob = float(os.environ.get('HUB_PROB', 0.04))
rng = np.random.Generator(np.random.PCG64(seed))
idx = 0
while idx < hi:
    ei, n = F.random_graph(rng, hub_prob)
    if ei.shape[1] == 0:
        continue
    idx += 1
    if idx >= lo:
        deg = np.bincount(ei[0], minlength=n)
        print(f'graph {idx}: n={n} E={ei.shape[1] // 2} max degree {int(deg.max())}', flush=True)
        C = c_oracle.CGraph(ei, n)
        for impl in ('nc', 'edge', 'h2'):
            os.environ['DCR_PASS'] = impl
            G = DcrGraph(ei, n)
            os.environ.pop('DCR_PASS')
            for ct in ('bfc', '1d', 'augmented', 'haantjes'):
                try:
                    cv = G.curvature_all(ct)[2]
                    oc = C.curv_all(ct, nthreads=8)[2]
                    if not np.array_equal(cv, oc):
                        print(f'  DIFFERENT: {impl} {ct}: {int((cv != oc).sum())} edges', flush=True)
                except Exception as e:
                    print(f'  ERROR: {impl} {ct} (engine {G.pass_engine()}): {e!r}', flush=True)
    for step in range(12):
        rng.integers(0, n, 2)
    if ei.shape[1] >= 4 and n <= 1500:
        rng.integers(0, 4); rng.choice([float('inf'), 5.0, 50.0, 163.0]); rng.choice([0.0, 0.5, 0.95, 3.0])
        rng.integers(1 << 20); rng.integers(1, 25); rng.integers(0, 2); rng.integers(0, 2)
