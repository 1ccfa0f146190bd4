"""Incremental SDRF iterations at the bench shape: step and pass time; DCR_NC_TRACE=1 prints when the waves of every class of
the node-centric pass started and made their last progress (K small then)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import numpy as np
import torch
from dcr import synthetic
from dcr.data import Data
from rewiring import sdrf_no_cuda as S
ei, n = synthetic.powerlaw_graph(int(os.environ.get("N", 100000)), int(os.environ.get("M", 10)), seed=12345)
K = int(os.environ.get('K', 200))
np.random.seed(0)
run = S.SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.95, 163.0, incremental=True)
for _ in range(8):
    run.step()
run.G.profile_reset()
t0 = time.perf_counter()
for _ in range(K):
    run.step()
el = time.perf_counter() - t0
ms, cnt = run.G.profile_read()
print(f'incremental step {el / K * 1e3:.4f} ms  pass {ms / max(cnt, 1):.4f} ms x{cnt}  device draws {run.device_draws} host {run.host_draws}', flush=True)
