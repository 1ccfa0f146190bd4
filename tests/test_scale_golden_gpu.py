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
        x, y, n_cand = run.last          # every iteration of the untraced legs: the arg-min edge and its candidate count
        want = fix['iterations'][i]
        assert [x, y] == want['argmin'] and n_cand == want['n_candidates'], (i, (x, y, n_cand), want['argmin'], want['n_candidates'])
    return run


def _final_checks(run, fix, what, curvatures=True):
    out = run.result().edge_index.numpy()
    assert out.dtype == np.int64 and out.shape[1] // 2 == fix['final']['edges'], what
    assert _sha(out) == fix['final']['edge_index_sha256'], what
    assert float(np.random.random_sample()).hex() == fix['numpy_next_uniform_hex'], what   # numpy's stream is where the oracle left it
    if curvatures:
        _check_curvatures(run.G, fix['rewired_curvature'], what + ': rewired graph')


TRACED = 25   # iterations replayed with the improvements on the host (candidate list and improvement vector hashed): ~1 s each


def test_s100k_sdrf_follows_the_oracle_trace():
    """configs[2] AS WRITTEN: the 100k-node / 1M-edge graph, tau = 163, bound 0.95, seed 0, all 500 iterations (round 5; 25 in
    round 4) — the first 25 traced (improvements on the host, numpy's draw: candidate list and improvement vector against their
    SHA-256), then the whole run untraced (draw on the device, fused tail + next pass) and with the incremental pass: arg-min
    edge and candidate count of EVERY iteration, final edge list, numpy's stream position, sampled + heaviest curvatures."""
    fix = load_golden('sdrf_s100k_oracle.json')
    ei, n = _graph(fix)
    loops = len(fix['iterations'])
    assert loops >= 25
    _traced_run(ei, n, fix, min(TRACED, loops))
    run = _untraced_run(ei, n, fix, loops, incremental=False)
    assert run.device_draws + run.host_draws == loops - 1    # (the last iteration goes the plain way: no pass follows it)
    _final_checks(run, fix, 'device draw')
    _final_checks(_untraced_run(ei, n, fix, loops, incremental=True), fix, 'incremental', curvatures=False)


def test_s100k_removal_branch_follows_the_oracle_trace():
    """rewiring/sdrf_no_cuda.py:57-63 at full size (round 5): the stale arg-max, the exclusion of the edge just added and the
    conditional removal had only ever run against the oracle below 2,485 nodes — no curvature of these graphs exceeds the
    reference's bounds.  tests/golden/sdrf_s100k_removal_oracle.json: S100k, Citeseer's tau = 180, bound -1.19 (20 edges of the
    initial graph lie above it): 40 iterations, the first 20 remove, the last 20 do not.  Traced, untraced, incremental."""
    fix = load_golden('sdrf_s100k_removal_oracle.json')
    removed = [it['removed'] for it in fix['iterations']]
    assert sum(r is not None for r in removed) >= 5 and sum(r is None for r in removed) >= 5   # both outcomes of :62
    ei, n = _graph(fix)
    loops = len(fix['iterations'])
    _final_checks(_traced_run(ei, n, fix, loops), fix, 'traced')
    _final_checks(_untraced_run(ei, n, fix, loops, incremental=False), fix, 'device draw')
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


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'sdrf_s1m_removal_oracle.json')),
                    reason='tests/golden/sdrf_s1m_removal_oracle.json not recorded')
def test_s1m_removal_step_follows_the_oracle_trace():
    """One iteration of the removal case at 1M nodes / 10M edges (tau = 180, bound -1.19): the added AND the removed edge, the
    final edge list, sampled + heaviest curvatures of the rewired graph."""
    fix = load_golden('sdrf_s1m_removal_oracle.json')
    assert all(it['removed'] is not None for it in fix['iterations'])
    ei, n = _graph(fix)
    _final_checks(_traced_run(ei, n, fix, len(fix['iterations'])), fix, 'traced')
