"""SDRF iteration variants interleaved in ONE run (blocks of 10 iterations, round robin), so that drift of the box cancels:
host draw against device draw."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import numpy as np
import torch
from dcr import synthetic
from dcr.data import Data
from rewiring import sdrf_no_cuda as S
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
inc = os.environ.get('INC', '0') == '1'
np.random.seed(0)
run = S.SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.95, float(os.environ.get('TAU', '163')), incremental=inc)
for _ in range(10):
    run.step()
variants = {'host': (False, None, '0'), 'device': (True, None, '0')}
acc = {k: [0.0, 0.0, 0] for k in variants}
for rnd in range(int(os.environ.get('ROUNDS', 10))):
    for name, (dd, sync, pre) in variants.items():
        run.device_draw = dd
        os.environ.pop('DCR_DD_SYNC', None)
        if sync:
            os.environ['DCR_DD_SYNC'] = sync
        os.environ['DCR_ESET_PREBUILD'] = pre
        run.step(); run.step()
        run.G.profile_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            run.step()
        el = time.perf_counter() - t0
        ms, cnt = run.G.profile_read()
        a = acc[name]
        a[0] += el; a[1] += ms; a[2] += cnt
for name, (el, ms, cnt) in acc.items():
    print(f'{name:>16}: step {el / cnt * 1e3:.4f} ms  pass {ms / cnt:.4f} ms  other {(el / cnt * 1e3) - ms / cnt:.4f} ms', flush=True)
