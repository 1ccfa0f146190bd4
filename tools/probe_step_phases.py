"""Host-side phase times of the SDRF iteration at the bench shape (S100k): improvement pipeline + its synchronisation, the
host draw, the fused tail + pass + first minimum.  Not part of the bench contract."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import numpy as np
import torch
from dcr import synthetic
from dcr.data import Data
from rewiring import sdrf_no_cuda as S

ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
np.random.seed(0)
inc = os.environ.get('INC', '0') == '1'
run = S.SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.95, 163.0, incremental=inc)
G = run.G
for _ in range(5):
    run.step()
x, y = run._next_argmin
T = {'improvements': 0.0, 'draw': 0.0, 'draw_mulexp': 0.0, 'draw_sum': 0.0, 'tail_pass': 0.0}
K = int(os.environ.get('K', 40))
ncand = 0
G.profile_reset()
for _ in range(K):
    t0 = time.perf_counter()
    imp, ci, cj = G.improvements(x, y, 'bfc', want_candidates=False)
    t1 = time.perf_counter()
    a = np.asarray(imp)
    e = S._exp_scratch[:a.shape[0]] if S._exp_scratch.shape[0] >= a.shape[0] else np.empty_like(a)
    ta = time.perf_counter()
    np.multiply(a, 163.0, out=e); np.exp(e, out=e)
    tb = time.perf_counter()
    e.sum()
    tc = time.perf_counter()
    st = np.random.get_state()
    t1b = time.perf_counter()
    idx = S.draw_index(imp, 163.0)
    t2 = time.perf_counter()
    (k, l), removed, nxt = G.sdrf_tail_at_pass_argmin(idx, True, 0.95, 'bfc', incremental=inc)
    t3 = time.perf_counter()
    x, y = nxt[:2]
    T['improvements'] += t1 - t0; T['draw'] += t2 - t1b; T['tail_pass'] += t3 - t2
    T['draw_mulexp'] += tb - ta; T['draw_sum'] += tc - tb
    ncand += a.shape[0]
print({k: round(v / K * 1e3, 4) for k, v in T.items()}, 'pass_ms', round((lambda m, c: m / max(c, 1))(*G.profile_read()), 4), 'candidates', ncand // K, flush=True)
