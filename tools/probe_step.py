"""SDRF iteration time and pass time at the bench shape, device draw against host draw (DCR_DEVICE_DRAW), same process."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import numpy as np
import torch
from dcr import synthetic
from dcr.data import Data
from rewiring import sdrf_no_cuda as S
ei, n = synthetic.powerlaw_graph(int(os.environ.get("N", 100000)), int(os.environ.get("M", 10)), seed=12345)
inc = os.environ.get('INC', '0') == '1'
K = int(os.environ.get('K', 60))
for mode in ('1', '0', '1', '0'):
    os.environ['DCR_DEVICE_DRAW'] = mode
    np.random.seed(0)
    run = S.SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.95, 163.0, incremental=inc)
    for _ in range(8):
        run.step()
    run.G.profile_reset()
    t0 = time.perf_counter()
    for _ in range(K):
        run.step()
    el = time.perf_counter() - t0
    ms, cnt = run.G.profile_read()
    print(f'device_draw={mode} step {el / K * 1e3:.4f} ms  pass {ms / max(cnt, 1):.4f} ms x{cnt}  device draws {run.device_draws} host {run.host_draws}', flush=True)
