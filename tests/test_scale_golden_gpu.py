"""BASELINE.json's two synthetic sizes end to end against traces recorded with the pinned C oracle
(tools/make_golden_scale.py -> tests/golden/sdrf_s100k_oracle.json, sdrf_s1m_oracle.json): configs[2] (100k nodes / 1M edges)
and the rewiring step of configs[4] (1M nodes / 10M edges).  Per iteration of rewiring/sdrf_no_cuda.py:22-66 the arg-min
edge and its curvature, the candidate list and the improvement vector (SHA-256 of their bytes), the drawn index, the added
and the removed edge; then the final edge list (SHA-256 of the int64 edge_index), numpy's stream position, and sampled +
heaviest-edge curvatures of the initial and of the rewired graph.  Bit-exact throughout."""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _graph(fix):
    from dcr import synthetic
    g = fix['graph']
    ei, n = synthetic.powerlaw_graph(g['n'], g['m'], seed=g['seed'])
    assert ei.shape[1] // 2 == g['edges'] and _sha(ei) == g['edge_index_sha256']   # the generator is part of the fixture
    return ei, n


def _check_curvatures(G, block, what):
    eu, ev, cv = G.curvature_all('bfc')
    key = eu.astype(np.int64) * G.num_nodes + ev
    order = np.argsort(key)
    want_key = np.array(block['u'], dtype=np.int64) * G.num_nodes + np.array(block['v'], dtype=np.int64)
    pos = order[np.searchsorted(key[order], want_key)]
    assert np.array_equal(key[pos], want_key), what
    want = np.array([float.fromhex(h) for h in block['bfc_hex']])
    bad = np.flatnonzero(cv[pos].view(np.int64) != want.view(np.int64))
    assert bad.size == 0, (what, bad[:5], cv[pos][bad[:5]], want[bad[:5]])


def _traced_run(ei, n, fix, loops):
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    np.random.seed(fix['numpy_seed'])
    trace = []
    run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', fix['remove_edges'], fix['removal_bound'], fix['tau'],
                  trace=trace)
    _check_curvatures(run.G, fix['initial_curvature'], 'initial graph')
    for i in range(loops):
        more = run.step(more=i + 1 < loops)
        got, want = trace[-1], fix['iterations'][i]
        assert got['argmin'] == want['argmin'], (i, got['argmin'], want['argmin'])
        assert len(got['candidates']) == want['n_candidates'], i
        assert _sha(np.array(got['candidates'], dtype=np.int32).reshape(-1, 2)) == want['candidates_sha256'], (i, 'candidate list')
        assert _sha(np.array(got['improvements'], dtype=np.float64)) == want['improvements_sha256'], (i, 'improvement vector')
        assert got['choice'] == want['choice'] and got['added'] == want['added'] and got['removed'] == want['removed'], (i, got['choice'], want)
        got.pop('candidates'), got.pop('improvements')   # (tens of MB of Python lists per iteration)
        assert more
    return run


def _untraced_run(ei, n, fix, loops, incremental):
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    np.random.seed(fix['numpy_seed'])
    run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', fix['remove_edges'], fix['removal_bound'], fix['tau'],
                  incremental=incremental)
    for i in range(loops):
        assert run.step(more=i + 1 < loops)
    return run


def _final_checks(run, fix, what, curvatures=True):
    out = run.result().edge_index.numpy()
    assert out.dtype == np.int64 and out.shape[1] // 2 == fix['final']['edges'], what
    assert _sha(out) == fix['final']['edge_index_sha256'], what
    assert float(np.random.random_sample()).hex() == fix['numpy_next_uniform_hex'], what   # numpy's stream is where the oracle left it
    if curvatures:
        _check_curvatures(run.G, fix['rewired_curvature'], what + ': rewired graph')


def test_s100k_sdrf_follows_the_oracle_trace():
    """configs[2]: the 100k-node / 1M-edge graph, tau = 163, bound 0.95, seed 0, 25 iterations — traced (improvements on the
    host, numpy's draw), untraced (draw on the device, fused tail + next pass) and with the incremental pass."""
    fix = load_golden('sdrf_s100k_oracle.json')
    ei, n = _graph(fix)
    loops = len(fix['iterations'])
    assert loops >= 25
    _final_checks(_traced_run(ei, n, fix, loops), fix, 'traced')
    run = _untraced_run(ei, n, fix, loops, incremental=False)
    assert run.device_draws + run.host_draws == loops - 1    # (the last iteration goes the plain way: no pass follows it)
    _final_checks(run, fix, 'device draw')
    _final_checks(_untraced_run(ei, n, fix, loops, incremental=True), fix, 'incremental', curvatures=False)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'sdrf_s1m_oracle.json')),
                    reason='tests/golden/sdrf_s1m_oracle.json not recorded')
def test_s1m_rewiring_step_follows_the_oracle_trace():
    """configs[4], rewiring half: the 1M-node / 10M-edge graph, three iterations, traced and untraced; 5,000 sampled + the 300
    heaviest edges' curvatures before and after."""
    fix = load_golden('sdrf_s1m_oracle.json')
    ei, n = _graph(fix)
    loops = len(fix['iterations'])
    _final_checks(_traced_run(ei, n, fix, loops), fix, 'traced')
    _final_checks(_untraced_run(ei, n, fix, loops, incremental=False), fix, 'device draw', curvatures=False)
