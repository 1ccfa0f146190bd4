"""Ad-hoc timing probe on the north-star graph (not part of the bench contract)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
from dcr.graph import DcrGraph

t0 = time.time()
ei, n = synthetic.powerlaw_graph(int(os.environ.get('N', 100000)), 10, seed=12345)
print('gen', time.time() - t0, 's  E=', ei.shape[1] // 2, flush=True)
deg = np.bincount(ei[0], minlength=n)
print('max deg', deg.max(), 'sum d^2', float((deg.astype(np.float64) ** 2).sum()))
t0 = time.time(); G = DcrGraph(ei, n); print('upload', time.time() - t0, flush=True)
t0 = time.time(); G.curvature_pass('bfc'); print('first pass', time.time() - t0, flush=True)
G.profile_reset()
for _ in range(5):
    G.curvature_pass('bfc')
ms, cnt = G.profile_read()
print('pass ms', ms / cnt, flush=True)
B = G.bfc_algorithmic_bytes()
print('alg bytes', B, 'GB/s', B / (ms / cnt * 1e-3) / 1e9, flush=True)
x, y, v = G.argext(False); print('argmin', x, y, v, 'deg', G.degree(x), G.degree(y))
for tau in (float('inf'), 163):
    t0 = time.time()
    if tau == float('inf'):
        nc = G.improvements_count(x, y); idx = G.improvements_argmax()
    else:
        imp, _, _ = G.improvements(x, y); nc = imp.shape[0]
        t1 = time.time()
        e = np.exp(np.array(imp) * tau); p = e / e.sum(); idx = np.random.choice(nc, p=p)
        print('  host softmax+draw', time.time() - t1)
    print('improvements tau', tau, 'n_cand', nc, time.time() - t0, 's')
from dcr.data import Data
from rewiring.sdrf_no_cuda import sdrf_no_cuda
import torch
d = Data(edge_index=torch.from_numpy(ei), num_nodes=n)
for tau in (float('inf'), 163):
    np.random.seed(0)
    t0 = time.time(); out = sdrf_no_cuda(d, 'bfc', 20, True, 0.95, tau); dt = time.time() - t0
    print('sdrf 20 iters tau', tau, dt, 's ->', 20 / dt, 'it/s (incl. upload+export)', flush=True)
