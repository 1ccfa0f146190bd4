"""Where one SDRF iteration on the north-star graph spends its time (production call sequence of SdrfRun.step;
host wall-clock around each library call, so each figure includes its host sync).  Not part of the bench contract."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.graph import DcrGraph
from rewiring.sdrf_no_cuda import draw_index
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
G = DcrGraph(ei, n)
np.random.seed(0)
T = {k: 0.0 for k in ('pass+argmin', 'improvements', 'draw', 'tail')}
ncs = []
iters = int(os.environ.get('ITERS', 100))
for it in range(iters + 5):
    if it == 5:
        T = {k: 0.0 for k in T}
        t_all = time.perf_counter()
    t = time.perf_counter(); x, y, _ = G.curvature_pass_argmin('bfc'); T['pass+argmin'] += time.perf_counter() - t
    t = time.perf_counter(); imp, _, _ = G.improvements(x, y, 'bfc', want_candidates=False); T['improvements'] += time.perf_counter() - t
    ncs.append(imp.shape[0])
    t = time.perf_counter(); idx = draw_index(imp, 163.0); T['draw'] += time.perf_counter() - t
    t = time.perf_counter(); G.sdrf_tail_at(idx, True, 0.95); T['tail'] += time.perf_counter() - t
total = time.perf_counter() - t_all
for k, v in T.items():
    print(f'{k:14s} {v / iters * 1e3:8.3f} ms/iter')
print(f'total          {total / iters * 1e3:8.3f} ms/iter; candidates: mean {np.mean(ncs):.0f} max {max(ncs)}')
