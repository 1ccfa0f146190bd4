"""SDRF iterations with the incremental pass on the north-star graph (timing / profiling aid)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data
from rewiring.sdrf_no_cuda import SdrfRun
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
np.random.seed(0)
run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.95, 163.0, incremental=True)
for _ in range(5):
    run.step()
run.G.profile_reset()
iters = int(os.environ.get('ITERS', 200))
t = time.perf_counter()
for _ in range(iters):
    run.step()
el = time.perf_counter() - t
ms, cnt = run.G.profile_read()
print(f'{el / iters * 1e3:.3f} ms per iteration, incremental pass {ms / cnt:.3f} ms', flush=True)
