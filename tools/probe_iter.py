import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data
from dcr.graph import DcrGraph
from utils.softmax import softmax
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
G = DcrGraph(ei, n)
np.random.seed(0)
T = {k: 0.0 for k in ('pass', 'argmin', 'imp', 'softmax', 'choice', 'cand', 'tail')}
ncs = []
for it in range(60):
    t = time.perf_counter(); G.curvature_pass('bfc'); T['pass'] += time.perf_counter() - t
    t = time.perf_counter(); x, y, _ = G.argext(False); T['argmin'] += time.perf_counter() - t
    t = time.perf_counter(); imp, _, _ = G.improvements(x, y, 'bfc'); T['imp'] += time.perf_counter() - t
    ncs.append(imp.shape[0])
    t = time.perf_counter(); p = softmax(np.array(imp), tau=163); T['softmax'] += time.perf_counter() - t
    t = time.perf_counter(); from rewiring.sdrf_no_cuda import choice_index; idx = choice_index(p); T['choice'] += time.perf_counter() - t
    t = time.perf_counter(); k, l = G.candidate_at(idx); T['cand'] += time.perf_counter() - t
    t = time.perf_counter(); G.sdrf_tail((k, l), True, 0.95); T['tail'] += time.perf_counter() - t
for k, v in T.items(): print(f'{k:8s} {v / 60 * 1e3:8.3f} ms/iter')
print('candidates: mean', np.mean(ncs), 'max', max(ncs))
