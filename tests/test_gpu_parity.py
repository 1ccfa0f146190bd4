"""Parity of the HIP path (through the C ABI) against the reference-generated golden
fixtures and against the CPU oracle on seeded inputs.  Bit-exact everywhere:
curvatures are float64 and compared with ==, edge lists are integers."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def fh(s):
    return float.fromhex(s)


@pytest.fixture(scope='module')
def dcr():
    from dcr.graph import DcrGraph
    return DcrGraph


@pytest.fixture(scope='module')
def oracle():
    from oracle import c_oracle
    return c_oracle


def test_kat(dcr):
    for k in load_golden('kat_curvature.json')['kat']:
        G = dcr(np.array(k['edge_index']), k['num_nodes'])
        assert G.curvature_edge(k['u'], k['v']) == fh(k['bfc']), k['graph']
        assert G.curvature_edge(k['v'], k['u']) == fh(k['bfc']), k['graph']


@pytest.mark.parametrize('fname', ['fullpass_small.json', 'fullpass_sampled.json'])
def test_fullpass_golden(dcr, fname):
    for name, rec in load_golden(fname)['graphs'].items():
        G = dcr(np.array(rec['edge_index']), rec['num_nodes'])
        for ct, want in (('bfc', [fh(h) for h in rec['bfc']]), ('1d', rec['1d']), ('augmented', rec['augmented']),
                         ('haantjes', rec['haantjes'])):
            eu, ev, cv = G.curvature_all(ct)
            got = {(u, v): c for u, v, c in zip(eu.tolist(), ev.tolist(), cv.tolist())}
            if not rec['sampled']:
                assert [[u, v] for u, v in zip(eu.tolist(), ev.tolist())] == rec['edges'], (name, 'edge order')
            for (u, v), w in zip(rec['edges'], want):
                assert got[(u, v)] == float(w), (name, ct, u, v)


def _run_case(case):
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import sdrf_no_cuda
    import torch
    tau = float('inf') if case['tau'] == 'inf' else case['tau']
    data = Data(edge_index=torch.tensor(case['edge_index']), num_nodes=case['num_nodes'])
    trace = []
    np.random.seed(case['seed'])
    out = sdrf_no_cuda(data, case['curv_type'], case['loops'], case.get('remove_edges', True), case['removal_bound'],
                       tau, trace=trace)
    return trace, out.edge_index.numpy()


@pytest.mark.parametrize('fname', ['sdrf_traces_small.json', 'sdrf_traces_medium.json'])
def test_sdrf_traces_golden(fname):
    for case in load_golden(fname)['cases']:
        label = {k: case[k] for k in ('graph', 'curv_type', 'loops', 'tau', 'seed')}
        if case['error']:
            with pytest.raises(ValueError):
                _run_case(case)
            continue
        trace, final = _run_case(case)
        ref = case['iterations']
        assert len(trace) == len(ref), label
        for it, (a, b) in enumerate(zip(trace, ref)):
            assert b['argmin'] is None or a['argmin'] == b['argmin'], (label, it)
            assert a['candidates'] == b['candidates'], (label, it)
            assert [float(v).hex() for v in a['improvements']] == [fh(h).hex() for h in b['improvements']], (label, it)
            assert a['choice'] == b['choice'], (label, it)
            assert a['added'] == b['added'], (label, it)
            assert a['removed'] == b['removed'], (label, it)
        assert final.tolist() == case['final_edge_index'], label


def test_sdrf_untraced_paths_match_golden():
    """The production paths (no trace: candidates stay on the device; tau=inf: device arg-max) give the same
    final edge list as the traced path checked above."""
    from dcr.data import Data
    from rewiring.rewire import rewire
    import torch
    for case in load_golden('sdrf_traces_small.json')['cases']:
        if case['error'] or not case.get('remove_edges', True):
            continue
        tau = float('inf') if case['tau'] == 'inf' else case['tau']
        data = Data(edge_index=torch.tensor(case['edge_index']), num_nodes=case['num_nodes'])
        np.random.seed(case['seed'])
        ei = rewire(data, case['curv_type'], case['loops'], case['removal_bound'], tau)
        assert ei.tolist() == case['final_edge_index'], case['graph']


@pytest.mark.parametrize('n,m,seed', [(3000, 4, 1), (1500, 12, 2), (400, 40, 3)])
def test_fullpass_vs_oracle(dcr, oracle, n, m, seed):
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(n, m, seed=seed)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    for ct in ('bfc', 'augmented', 'haantjes', '1d'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        assert np.array_equal(cv, oc), ct
    # integer ingredients on a few of the heaviest edges
    eu, ev = G.edges()
    deg = np.bincount(ei[0], minlength=nn)
    heavy = np.argsort(-(deg[eu] + deg[ev]))[:20]
    for e in heavy.tolist():
        assert G.bfc_ingredients(int(eu[e]), int(ev[e])).tolist() == C.ingredients(int(eu[e]), int(ev[e])).tolist()


def test_dense_graph_hub_bins(dcr, oracle):
    """Erdos-Renyi with large degrees: exercises the workgroup-per-edge bins."""
    from dcr import synthetic
    ei, nn = synthetic.erdos_renyi_graph(1400, 0.45, seed=5)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    eu, ev, cv = G.curvature_all('bfc')
    rng = np.random.Generator(np.random.PCG64(1))
    pick = rng.choice(eu.shape[0], size=300, replace=False)
    want = C.curv_edges(eu[pick], ev[pick], 'bfc', nthreads=8)
    assert np.array_equal(cv[pick], want)


@pytest.mark.parametrize('ct', ['bfc', 'augmented'])
def test_improvements_vs_oracle(dcr, oracle, ct):
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(600, 6, seed=11)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    eu, ev = G.edges()
    deg = np.bincount(ei[0], minlength=nn)
    order = np.argsort(-(deg[eu] * deg[ev]))
    picks = order[:4].tolist() + order[len(order) // 2:len(order) // 2 + 3].tolist() + order[-3:].tolist()
    for e in picks:
        x, y = int(eu[e]), int(ev[e])
        imp, ci, cj = G.improvements(x, y, ct, want_candidates=True)
        oi, oj = C.candidates(x, y)
        assert np.array_equal(ci, oi) and np.array_equal(cj, oj), (x, y)
        want = C.improvements(x, y, oi, oj, ct)
        assert np.array_equal(np.array(imp), want), (x, y)
        if len(oi):
            assert G.improvements_argmax() == int(np.argmax(want))


def test_sdrf_vs_oracle_medium(oracle):
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import sdrf_no_cuda
    import torch
    ei, nn = synthetic.powerlaw_graph(1200, 5, seed=21)
    for tau, bound, seed in ((163, 0.95, 0), (float('inf'), 0.5, 1), (20, 0.3, 2)):
        np.random.seed(seed)
        want = oracle.sdrf(ei, nn, 'bfc', 25, True, bound, tau, nthreads=8)
        np.random.seed(seed)
        got = sdrf_no_cuda(Data(edge_index=torch.from_numpy(ei), num_nodes=nn), 'bfc', 25, True, bound, tau)
        assert np.array_equal(got.edge_index.numpy(), want), (tau, bound)


def test_row_overflow_relayout(dcr, oracle):
    """Keep adding edges at one node until its slack is exhausted: the re-layout must keep
    row order and the stale curvature buffer aligned."""
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(300, 3, seed=4)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    G.curvature_pass('bfc')
    hub = int(np.argmax(np.bincount(ei[0], minlength=nn)))
    added = 0
    for v in range(nn):
        if v != hub and not C.has_edge(hub, v):
            rem, _ = G.sdrf_tail((hub, v), True, 1e9)  # bound too high: nothing is removed
            assert rem is None
            C.add_edge(hub, v)
            added += 1
            if added == 60:
                break
    assert np.array_equal(G.to_edge_index(), C.to_edge_index())
    eu, ev, cv = G.curvature_all('bfc')
    ou, ov, oc = C.curv_all('bfc')
    assert np.array_equal(eu, ou) and np.array_equal(ev, ov) and np.array_equal(cv, oc)


def test_container_ops(dcr):
    ei = np.array([[1, 2, 2, 3, 0, 0, 1, 2], [0, 0, 1, 2, 1, 2, 2, 3]])
    G = dcr(ei, 5)
    assert G.number_of_edges() == 4
    assert G.neighbors(2) == [0, 1, 3]
    assert G.has_edge(0, 1) and not G.has_edge(0, 3) and not G.has_edge(4, 4)
    G.add_edge(0, 3)
    G.add_edge(3, 0)  # no-op, position unchanged
    assert G.neighbors(0) == [1, 2, 3] and G.degree(3) == 2
    G.remove_edge(0, 2)
    assert G.neighbors(0) == [1, 3] and G.neighbors(2) == [1, 3]
    with pytest.raises(KeyError):
        G.remove_edge(0, 2)
    with pytest.raises(ValueError):
        G.add_edge(1, 1)
    with pytest.raises(ValueError):
        dcr(np.array([[0, 1], [0, 0]]), 2)  # self-loop
    assert G.to_edge_index().tolist() == [[0, 0, 1, 1, 2, 2, 3, 3], [1, 3, 0, 2, 1, 3, 0, 2]]
    eu, ev = G.edges()
    assert list(zip(eu.tolist(), ev.tolist())) == [(0, 1), (0, 3), (1, 2), (2, 3)]


def test_unsorted_duplicate_input(dcr, oracle):
    """Non-coalesced input: order of first appearance with dst <= src decides row order."""
    rng = np.random.Generator(np.random.PCG64(9))
    n = 50
    src = rng.integers(0, n, 400)
    dst = rng.integers(0, n, 400)
    keep = src != dst
    ei = np.stack([np.concatenate([src[keep], dst[keep]]), np.concatenate([dst[keep], src[keep]])])
    perm = rng.permutation(ei.shape[1])
    ei = ei[:, perm]
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    assert np.array_equal(G.to_edge_index(), C.to_edge_index())
    eu, ev, cv = G.curvature_all('bfc')
    _, _, oc = C.curv_all('bfc')
    assert np.array_equal(cv, oc)
