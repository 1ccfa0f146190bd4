"""GPU-box checker: a long SDRF run on the north-star graph, full recompute against the incremental pass, final edge lists
compared, and compared again with the list of a run whose draws are all made on the host; then the final graph's full pass
against the C oracle on sampled edges.  tests/test_checkers_gpu.py runs it with
200 iterations; the long version: ITERS=3000 python tests/check_soak.py (BOUND=-1.19 TAU=180: with removals)"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data
from oracle import c_oracle
from rewiring.sdrf_no_cuda import SdrfRun


def run(iters=3000, samples=20000, tau=163.0, bound=0.95):
    """tau, bound: BASELINE's configs[2] by default; bound=-1.19, tau=180 removes an edge in most iterations (tests/golden/sdrf_s100k_removal_oracle.json)"""
    ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
    out = {}
    for inc in (False, True):
        np.random.seed(0)
        run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, bound, tau, incremental=inc)
        t = time.time()
        done = 0
        for _ in range(iters):
            if not run.step():
                break
            done += 1
        print(f'incremental={inc}: {done} iterations in {time.time() - t:.1f} s, {run.G.number_of_edges()} edges', flush=True)
        out[inc] = run.result().edge_index.numpy()
        if inc:
            run.G.curvature_pass('bfc')
            eu, ev, cv = run.G.curvature_read()
    assert np.array_equal(out[False], out[True]), 'edge lists differ'
    # the same run once more with every draw on the host (numpy's own exp, sum and cumsum on the downloaded improvements):
    # the device-side draw must have picked numpy's index in every one of the iterations above, at ~190k candidates each
    os.environ['DCR_DEVICE_DRAW'] = '0'
    try:
        np.random.seed(0)
        run_h = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, bound, tau)
    finally:
        os.environ.pop('DCR_DEVICE_DRAW', None)
    t = time.time()
    for _ in range(done):
        if not run_h.step():
            break
    print(f'host draws: {done} iterations in {time.time() - t:.1f} s; device draws of the first run: {run.device_draws} '
          f'(undecided, redone on the host: {run.host_draws})', flush=True)
    assert np.array_equal(run_h.result().edge_index.numpy(), out[True]), 'device-side draws and host draws diverge'
    C = c_oracle.CGraph(out[True], n)
    rng = np.random.Generator(np.random.PCG64(1))
    pick = rng.choice(eu.shape[0], size=min(samples, eu.shape[0]), replace=False)
    want = C.curv_edges(eu[pick], ev[pick], 'bfc', nthreads=16)
    assert np.array_equal(cv[pick], want), 'curvatures of the rewired graph differ from the oracle'
    print(f'soak ok: identical edge lists, {samples} sampled curvatures of the rewired graph identical to the oracle', flush=True)
    return done


if __name__ == '__main__':
    run(int(os.environ.get('ITERS', 3000)), tau=float(os.environ.get('TAU', 163.0)), bound=float(os.environ.get('BOUND', 0.95)))
