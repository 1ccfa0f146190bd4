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


@pytest.mark.parametrize('fname', ['sdrf_grid_karate.json', 'sdrf_cora_shaped.json'])
def test_sdrf_compact_fixtures(fname):
    """The reference's own runs over the SURVEY §8(c) parameter grid and on the Cora-shaped graph (BASELINE.json
    configs[1]: 50 iterations, Cora hyperparameters): traced path per iteration, production path on the final list."""
    from dcr.data import Data
    from rewiring.rewire import rewire
    import torch
    gold = load_golden(fname)
    for case in gold['cases']:
        g = gold['graphs'][case['graph']]
        full = dict(case, edge_index=g['edge_index'], num_nodes=g['num_nodes'])
        label = {k: case[k] for k in ('graph', 'curv_type', 'loops', 'removal_bound', 'tau', 'seed')}
        if case['error']:
            with pytest.raises(ValueError):
                _run_case(full)
            continue
        trace, final = _run_case(full)
        ref = case['iterations']
        assert len(trace) == len(ref), label
        for it, (a, b) in enumerate(zip(trace, ref)):
            assert b['argmin'] is None or a['argmin'] == b['argmin'], (label, it)
            assert len(a['candidates']) == b['n_candidates'], (label, it)
            assert a['choice'] == b['choice'], (label, it)
            assert a['added'] == b['added'], (label, it)
            assert a['removed'] == b['removed'], (label, it)
        assert final.tolist() == case['final_edge_index'], label
        tau = float('inf') if case['tau'] == 'inf' else case['tau']
        data = Data(edge_index=torch.tensor(g['edge_index']), num_nodes=g['num_nodes'])
        np.random.seed(case['seed'])
        assert rewire(data, case['curv_type'], case['loops'], case['removal_bound'], tau).tolist() == \
            case['final_edge_index'], label


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


# ------------------------------------------------------------------------------------------------ full size
def test_s100k_properties(dcr, oracle):
    """BASELINE.json's north-star size.  The oracle cannot run a whole SDRF iteration here in seconds, so parity is
    checked through sampled edges and size-independent properties."""
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(100000, 10, seed=12345)
    E = ei.shape[1] // 2
    assert E == 999900
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    eu, ev, cv = G.curvature_all('bfc')
    ou, ov = C.edges()
    assert np.array_equal(eu, ou) and np.array_equal(ev, ov)                  # G.edges order at 1M edges
    rng = np.random.Generator(np.random.PCG64(42))
    deg = np.bincount(ei[0], minlength=nn)
    heavy = np.argsort(-(deg[eu].astype(np.int64) * deg[ev]))[:300]           # hub-hub edges: the big bins
    pick = np.concatenate([rng.choice(E, size=5000, replace=False), heavy])
    assert np.array_equal(cv[pick], C.curv_edges(eu[pick], ev[pick], 'bfc', nthreads=16))
    # idempotence: a second pass writes the same bits; relabelling-free symmetry: (v,u) == (u,v)
    _, _, cv2 = G.curvature_all('bfc')
    assert np.array_equal(cv, cv2)
    for e in heavy[:5].tolist() + pick[:5].tolist():
        assert G.curvature_edge(int(ev[e]), int(eu[e])) == cv[e] == G.curvature_edge(int(eu[e]), int(ev[e]))
    # arg-min / arg-max: first extremum in edge order
    x, y, val = G.argext(False)
    m = int(np.argmin(cv))
    assert (x, y, val) == (int(eu[m]), int(ev[m]), cv[m])
    x2, y2, val2 = G.argext(True)
    m2 = int(np.argmax(cv))
    assert (x2, y2, val2) == (int(eu[m2]), int(ev[m2]), cv[m2])
    # improvements of the arg-min (hub) edge: candidate list exact, sampled values against literal recompute
    imp, ci, cj = G.improvements(x, y, 'bfc', want_candidates=True)
    oi, oj = C.candidates(x, y)
    assert np.array_equal(ci, oi) and np.array_equal(cj, oj)
    sel = np.sort(rng.choice(len(oi), size=400, replace=False))
    assert np.array_equal(np.array(imp)[sel], C.improvements(x, y, oi[sel], oj[sel], 'bfc'))
    assert G.improvements_argmax() == int(np.argmax(np.array(imp)))
    # a few SDRF iterations: the rewired graph stays consistent with the oracle's view of the same operations
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    np.random.seed(0)
    trace = []
    run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=nn), 'bfc', True, 0.95, 163, trace=trace)
    for _ in range(3):
        run.step()
    for rec in trace:
        assert rec['removed'] is None or C.has_edge(*rec['removed'])
        if rec['added']:
            assert not C.has_edge(*rec['added'])
            C.add_edge(*rec['added'])
        if rec['removed']:
            C.remove_edge(*rec['removed'])
    assert np.array_equal(run.G.to_edge_index(), C.to_edge_index())
    eu3, ev3, cv3 = run.G.curvature_all('bfc')
    pick3 = rng.choice(eu3.shape[0], size=3000, replace=False)
    assert np.array_equal(cv3[pick3], C.curv_edges(eu3[pick3], ev3[pick3], 'bfc', nthreads=16))


def test_edge_cases(dcr, oracle):
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import sdrf_no_cuda
    import torch
    # no edges at all: the reference's min() over an empty sequence raises ValueError
    with pytest.raises(ValueError):
        sdrf_no_cuda(Data(edge_index=torch.zeros(2, 0, dtype=torch.long), num_nodes=4), 'bfc', 3, True, 0.5, 10)
    G = dcr(np.zeros((2, 0), dtype=np.int64), 4)
    assert G.number_of_edges() == 0 and G.to_edge_index().shape == (2, 0)
    G.curvature_pass('bfc')
    assert G.curvature_read()[2].shape == (0,)
    # one edge, isolated nodes: degree-1 rule, then candidates come only from the end points
    ei = np.array([[1, 0], [0, 1]])
    G = dcr(ei, 5)
    assert G.curvature_all('bfc')[2].tolist() == [0.0]
    imp, ci, cj = G.improvements(0, 1, 'bfc', want_candidates=True)
    assert imp.shape[0] == 0
    # star: all edges have a degree-1 end point -> curvature 0 everywhere; SDRF still matches the oracle
    star = np.array([[1, 2, 3, 4, 5, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 2, 3, 4, 5]])
    for ct in ('bfc', '1d', 'haantjes'):
        np.random.seed(3)
        want = oracle.sdrf(star, 6, ct, 6, True, 0.5, 7.0)
        np.random.seed(3)
        got = sdrf_no_cuda(Data(edge_index=torch.from_numpy(star), num_nodes=6), ct, 6, True, 0.5, 7.0)
        assert np.array_equal(got.edge_index.numpy(), want), ct
    # complete graph: no candidate can be added; with a high bound nothing is removed and the loop stops at once
    k5 = np.array([[i for i in range(5) for j in range(5) if i != j], [j for i in range(5) for j in range(5) if i != j]])
    trace = []
    out = sdrf_no_cuda(Data(edge_index=torch.from_numpy(k5), num_nodes=5), 'bfc', 10, True, 99.0, 1.0, trace=trace)
    assert len(trace) == 1 and out.edge_index.shape[1] == 20
    # remove_edges=False stops when nothing can be added
    trace = []
    sdrf_no_cuda(Data(edge_index=torch.from_numpy(k5), num_nodes=5), 'bfc', 10, False, 0.5, 1.0, trace=trace)
    assert len(trace) == 1


def test_s100k_values_of_the_reference_itself(dcr):
    """tests/golden/reference_timing_s100k.json holds the reference's own bfc_edge values for edges sampled from the
    north-star graph (tools/make_golden.py timed them for BASELINE's CPU figure): the full pass on that graph gives
    exactly those values at those edges."""
    from dcr import synthetic
    ref = load_golden('reference_timing_s100k.json')
    ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
    G = dcr(ei, n)
    assert G.number_of_edges() == ref['num_edges']
    eu, ev, cv = G.curvature_all('bfc')
    got = {(int(u), int(v)): c for u, v, c in zip(eu.tolist(), ev.tolist(), cv.tolist())}
    assert len(ref['values']) >= 40
    for u, v, hx_ in ref['values']:
        key = (u, v) if (u, v) in got else (v, u)
        assert got[key] == fh(hx_), (u, v)


@pytest.mark.parametrize('ct', ['1d', 'haantjes'])
def test_classical_improvements_vs_oracle(dcr, oracle, ct):
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(300, 5, seed=8)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    eu, ev = G.edges()
    for e in (0, 17, 400, len(eu) - 1):
        x, y = int(eu[e]), int(ev[e])
        imp, ci, cj = G.improvements(x, y, ct, want_candidates=True)
        oi, oj = C.candidates(x, y)
        assert np.array_equal(ci, oi) and np.array_equal(cj, oj)
        assert np.array_equal(np.array(imp), C.improvements(x, y, oi, oj, ct))


def test_bfc_cuda_call_surface(oracle):
    """curvature/bfc_cuda.py's entry points: names, arguments, shapes, the -1000 sentinel; bfc_naive numerics."""
    import torch
    from curvature.bfc_cuda import balanced_forman_curvature, balanced_forman_post_delta
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(40, 3, seed=1)
    A = torch.zeros(nn, nn, device='cuda')
    A[ei[0], ei[1]] = 1
    Cm = balanced_forman_curvature(A)
    O = oracle.CGraph(ei, nn)
    ou, ov, oc = O.curv_all('bfc')
    assert Cm.shape == (nn, nn) and Cm.dtype == A.dtype
    assert torch.equal(Cm, Cm.t())
    assert np.allclose(Cm[ou, ov].cpu().numpy(), oc.astype(np.float32), rtol=0, atol=0)
    assert float(Cm[A == 0].abs().sum()) == 0.0
    x, y = int(ou[3]), int(ov[3])
    xn = [int(v) for v in ei[1][ei[0] == x]] + [x]
    yn = [int(v) for v in ei[1][ei[0] == y]] + [y]
    D = balanced_forman_post_delta(A, x, y, xn, yn)
    assert D.shape == (len(xn), len(yn))
    before = O.curv_edge(x, y)
    for I, i in enumerate(xn):
        for J, j in enumerate(yn):
            if i == j or O.has_edge(i, j):
                assert float(D[I, J]) == -1000.0
            else:
                imp = O.improvements(x, y, np.array([min(i, j)]), np.array([max(i, j)]), 'bfc')[0]
                assert abs(float(D[I, J]) - (before + imp)) < 1e-6


# ------------------------------------------------------------------------------------------------ incremental mode
@pytest.mark.parametrize('route,every', [('edges', 3), ('rows', 3), ('classes', 3), ('edges', 7)])
def test_incremental_pass_equals_full(dcr, oracle, route, every, monkeypatch):
    """After arbitrary edits, an incremental pass leaves exactly the bits a full pass would: behind at most three edits (exact
    flags) by the edge-by-edge kernels of round 5 — their edge list from a sweep over the slots or from the rows of the flagged
    nodes (the touched list, large graphs' default) — and, with DCR_NC_FINE=0, by the class kernels; behind more edits (coarse
    flags) by the class kernels."""
    from dcr import synthetic
    monkeypatch.setenv('DCR_NC_FINE', '0' if route == 'classes' else '1')
    monkeypatch.setenv('DCR_NC_FINE_SWEEP', '0' if route == 'rows' else '1')   # the edge list from the flagged nodes' rows / a sweep
    ei, nn = synthetic.powerlaw_graph(800, 5, seed=13)
    G = dcr(ei, nn)
    C = oracle.CGraph(ei, nn)
    G.curvature_pass('bfc')
    rng = np.random.Generator(np.random.PCG64(17))
    for step in range(40):
        u, v = (int(t) for t in rng.integers(0, nn, 2))
        if u == v:
            continue
        if C.has_edge(u, v):
            G.remove_edge(u, v); C.remove_edge(u, v)
        else:
            G.add_edge(u, v); C.add_edge(u, v)
        if step % every == 0:
            G.curvature_pass('bfc', incremental=True)
            eu, ev, cv = G.curvature_read()
            ou, ov, oc = C.curv_all('bfc', nthreads=8)
            assert np.array_equal(eu, ou) and np.array_equal(ev, ov) and np.array_equal(cv, oc), step
    # switching curvature kind must not reuse the old buffer
    G.curvature_pass('augmented', incremental=True)
    assert np.array_equal(G.curvature_read()[2], C.curv_all('augmented')[2])


def test_sdrf_incremental_matches_golden_and_oracle(oracle):
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import sdrf_no_cuda
    import torch
    for fname in ('sdrf_traces_small.json', 'sdrf_traces_medium.json'):
        for case in load_golden(fname)['cases']:
            if case['error']:
                continue
            tau = float('inf') if case['tau'] == 'inf' else case['tau']
            data = Data(edge_index=torch.tensor(case['edge_index']), num_nodes=case['num_nodes'])
            np.random.seed(case['seed'])
            out = sdrf_no_cuda(data, case['curv_type'], case['loops'], case.get('remove_edges', True),
                               case['removal_bound'], tau, incremental=True)
            assert out.edge_index.tolist() == case['final_edge_index'], (case['graph'], case['curv_type'])
    ei, nn = synthetic.powerlaw_graph(1500, 6, seed=5)
    np.random.seed(4)
    want = oracle.sdrf(ei, nn, 'bfc', 40, True, 0.6, 120, nthreads=8)
    np.random.seed(4)
    got = sdrf_no_cuda(Data(edge_index=torch.from_numpy(ei), num_nodes=nn), 'bfc', 40, True, 0.6, 120, incremental=True)
    assert np.array_equal(got.edge_index.numpy(), want)


def test_s100k_incremental_equals_full():
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    ei, nn = synthetic.powerlaw_graph(100000, 10, seed=12345)
    outs = []
    for inc in (False, True):
        np.random.seed(0)
        run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=nn), 'bfc', True, 0.95, 163, incremental=inc)
        for _ in range(25):
            run.step()
        run.G.curvature_pass('bfc', incremental=inc)
        outs.append((run.G.to_edge_index(), run.G.curvature_read()[2]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------------------ degree limits
@pytest.mark.parametrize('n,da,db,dc', [(9000, 4300, 4200, 3000), (18000, 8195, 8180, 6000)])
def test_hubs_at_and_beyond_node_centric_limits(dcr, oracle, n, da, db, dc):
    """Three hubs joined to each other.  First case: 4,300 / 4,200 / 3,000 neighbours — the two largest node-centric
    classes (tables of 16,384 and 8,192 slots).  Second case: hub 0 has 8,197 neighbours, above every table size (its
    edges are owned by the other endpoints, including hub 1 with 8,182: the largest class at its limit).  Whole pass
    bit-compared with the oracle; then edits and an incremental pass."""
    rng = np.random.Generator(np.random.PCG64(99))
    src, dst = [0], [1]
    a = rng.choice(np.arange(3, n), size=da, replace=False)     # leaves of hub 0
    b = rng.choice(np.arange(3, n), size=db, replace=False)     # leaves of hub 1 (overlap: triangles on (0,1))
    c = rng.choice(np.arange(3, n), size=dc, replace=False)     # leaves of hub 2
    for hub, leaves in ((0, a), (1, b), (2, c)):
        src += [hub] * len(leaves); dst += leaves.tolist()
    src += [0, 1]; dst += [2, 2]
    extra = rng.integers(3, n, size=(2, 6000))                  # leaf-leaf edges: 4-cycles through the hubs
    src += extra[0].tolist(); dst += extra[1].tolist()
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    assert min(G.degree(0), G.degree(1)) > da - 200 and G.degree(2) > dc - 200
    C = oracle.CGraph(ei, n)
    for ct in ('bfc', 'augmented'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        bad = np.nonzero(cv != oc)[0]
        assert bad.size == 0, (ct, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:5]])
    # edits keep working across the class / fallback boundaries, and the incremental pass agrees
    G.add_edge(5, 6) if not C.has_edge(5, 6) else None
    C.add_edge(5, 6) if not C.has_edge(5, 6) else None
    G.remove_edge(0, 1); C.remove_edge(0, 1)
    G.curvature_pass('bfc')
    G.add_edge(0, 1); C.add_edge(0, 1)
    G.curvature_pass('bfc', incremental=True)
    eu, ev, cv = G.curvature_read()
    oc = C.curv_all('bfc', nthreads=8)[2]
    bad = np.nonzero(cv != oc)[0]
    assert bad.size == 0, (bad.size, [(int(eu[i]), int(ev[i]), G.degree(int(eu[i])), G.degree(int(ev[i])), cv[i], oc[i])
                                      for i in bad[:8]])


def test_two_pass_implementations_agree_at_scale(dcr):
    """300k nodes / 3M edges (no oracle at this size): the node-centric kernels and the edge-centric kernels are
    independent implementations and must leave identical bits, for every curvature kind."""
    import os
    from dcr import synthetic
    ei, nn = synthetic.powerlaw_graph(300000, 10, seed=4242)
    out = {}
    for impl in ('nc', 'edge'):
        os.environ['DCR_PASS'] = impl
        try:
            G = dcr(ei, nn)
        finally:
            os.environ.pop('DCR_PASS', None)
        out[impl] = [G.curvature_all(ct)[2] for ct in ('bfc', 'augmented')]
        del G
    for a, b in zip(out['nc'], out['edge']):
        assert a.shape[0] == ei.shape[1] // 2 and np.array_equal(a, b)


def test_neighbour_ids_that_collide_in_the_lds_table(dcr, oracle):
    """The node-centric pass keeps N(u) in an LDS hash set of 4-slot buckets (24-bit multiplicative hash) and tests
    streamed ids against the home bucket only unless the table recorded a spill.  Here the neighbour ids of several
    owners (one per table size class) are chosen to share their top hash bits, so buckets overflow into long chains:
    the spill flag, the walk to following buckets and the fast test all have to agree with the oracle."""
    n = 300000
    ids = np.arange(n, dtype=np.uint64)
    top12 = (((ids & 0xFFFFFF) * 0x9E3779) & 0xFFFFFF) >> 12          # bucket index of the largest table
    order = np.argsort(top12, kind='stable')
    groups = np.split(order, np.nonzero(np.diff(top12[order]))[0] + 1)
    groups = [g for g in groups if len(g) >= 60][:8]
    assert len(groups) == 8
    rng = np.random.Generator(np.random.PCG64(4))
    src, dst = [], []
    owners = [7, 11, 13, 17]
    sizes = [40, 60, 60, 60]                     # colliding neighbours per owner
    pads = [0, 120, 700, 3000]                   # ordinary neighbours on top: classes 0, 1, 2, 3
    for o, (own, sz, pad) in enumerate(zip(owners, sizes, pads)):
        coll = [int(x) for x in groups[o][:sz] if int(x) not in owners]
        rest = rng.choice(np.arange(20, n), size=pad, replace=False).tolist()
        for k in coll + rest:
            src.append(own); dst.append(k)
        # edges among the colliding ids and to the ids of the next group: triangles and 4-cycles through the chains
        for a, b in zip(coll[:-1], coll[1:]):
            src.append(a); dst.append(b)
        nxt = [int(x) for x in groups[o + 4][:sz]]
        for a, b in zip(coll, nxt):
            src.append(a); dst.append(b)
            src.append(b); dst.append(coll[(coll.index(a) + 3) % len(coll)])
    for a in range(len(owners)):
        for b in range(a + 1, len(owners)):
            src.append(owners[a]); dst.append(owners[b])
    ex = rng.integers(20, n, size=(2, 20000))
    src += ex[0].tolist(); dst += ex[1].tolist()
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    for ct in ('bfc', 'augmented', 'haantjes'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        bad = np.nonzero(cv != oc)[0]
        assert bad.size == 0, (ct, bad.size, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:5]])


def test_edges_beyond_every_lds_table(dcr, oracle):
    """Hubs joined to each other whose degrees exceed every table (12,000 + 11,000 + 2 > 16,384 keys, and one hub with
    17,000 neighbours): those edges take the device-memory path (csrc/dcr_bfc_giant.hip).  Whole pass, single-edge
    queries and an incremental pass after an edit, bit-compared with the oracle."""
    n = 40000
    rng = np.random.Generator(np.random.PCG64(21))
    src, dst = [], []
    hubs = [(0, 12000), (1, 11000), (2, 17000), (3, 9000)]
    for hub, d in hubs:
        leaves = rng.choice(np.arange(10, n), size=d, replace=False)
        src += [hub] * d; dst += leaves.tolist()
    for a in range(4):
        for b in range(a + 1, 4):
            src.append(a); dst.append(b)
    extra = rng.integers(10, n, size=(2, 30000))       # leaf-leaf edges: 4-cycles through the hubs
    src += extra[0].tolist(); dst += extra[1].tolist()
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    assert G.degree(0) + G.degree(1) + 2 > 16384 and G.degree(2) > 16382
    for ct in ('bfc', 'augmented', 'haantjes', '1d'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        bad = np.nonzero(cv != oc)[0]
        assert bad.size == 0, (ct, bad.size, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:5]])
    for (a, b) in ((0, 1), (2, 0), (1, 2), (3, 2)):
        assert G.curvature_edge(a, b) == C.curv_edge(a, b), (a, b)
    G.curvature_pass('bfc')
    G.remove_edge(0, 1); C.remove_edge(0, 1)
    k = next(int(x) for x in range(10, n) if not C.has_edge(2, int(x)))
    G.add_edge(2, k); C.add_edge(2, k)
    G.curvature_pass('bfc', incremental=True)
    eu, ev, cv = G.curvature_read()
    oc = C.curv_all('bfc', nthreads=8)[2]
    bad = np.nonzero(cv != oc)[0]
    assert bad.size == 0, (bad.size, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:5]])


def test_hub_edges_swept_from_the_hub_side(dcr, oracle):
    """Edges between a hub above every LDS table (> 8,190 neighbours) and a node of at most 1,022 neighbours are swept
    from the hub's side with a position map in device memory (k_hub_edges).  Three hubs (0 and 1 adjacent, 2 adjacent to
    neither but sharing thousands of leaves with 0, so that a leaf adjacent to 0 and 2 overflows the per-edge list of
    touched slots), leaves with a few random edges, degree-1 leaves, a 2,000-neighbour node adjacent to hub 0 (stays
    with the node-centric classes).  All curvature kinds, then edits and an incremental pass."""
    n = 40000
    rng = np.random.Generator(np.random.PCG64(77))
    src, dst = [], []
    l0 = rng.choice(np.arange(10, n), size=12000, replace=False)
    l1 = rng.choice(np.arange(10, n), size=9000, replace=False)
    l2 = np.concatenate([l0[:6000], rng.choice(np.setdiff1d(np.arange(10, n), l0), size=4000, replace=False)])
    for hub, leaves in ((0, l0), (1, l1), (2, l2)):
        src += [hub] * len(leaves); dst += leaves.tolist()
    src.append(0); dst.append(1)
    mid = 3
    lm = rng.choice(np.arange(10, n), size=2000, replace=False)
    src += [mid] * len(lm); dst += lm.tolist()
    src.append(mid); dst.append(0)
    extra = rng.integers(10, n, size=(2, 50000))
    src += extra[0].tolist(); dst += extra[1].tolist()
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    assert G.degree(0) > 8190 and G.degree(1) > 8190 and G.degree(2) > 8190 and 1022 < G.degree(mid) < 8190
    for ct in ('bfc', 'augmented', 'haantjes', '1d'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        bad = np.nonzero(cv != oc)[0]
        assert bad.size == 0, (ct, bad.size, [(int(eu[i]), int(ev[i]), G.degree(int(eu[i])), G.degree(int(ev[i])), cv[i], oc[i])
                                              for i in bad[:6]])
    # a second full pass gives the same values (the per-wave counters were left all-zero)
    G.curvature_pass('bfc')
    assert np.array_equal(G.curvature_read()[2], C.curv_all('bfc', nthreads=8)[2])
    leaf = int(l0[7])
    G.remove_edge(0, leaf); C.remove_edge(0, leaf)
    k = next(int(x) for x in range(10, n) if not C.has_edge(1, int(x)))
    G.add_edge(1, k); C.add_edge(1, k)
    G.curvature_pass('bfc', incremental=True)
    eu, ev, cv = G.curvature_read()
    oc = C.curv_all('bfc', nthreads=8)[2]
    bad = np.nonzero(cv != oc)[0]
    assert bad.size == 0, (bad.size, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:6]])


def test_hub_path_boundaries(dcr, oracle):
    """Degrees exactly at the limits that route an edge: a hub with 8,191 neighbours (one above the largest table), its
    edges to nodes with 1,022 neighbours (hub-side sweep) and 1,023 (owned by that node), and a second node with exactly
    8,190 neighbours (the largest table class) adjacent to the hub."""
    n = 30000
    rng = np.random.Generator(np.random.PCG64(13))
    pool = np.arange(10, n)
    src, dst = [], []

    def star(center, deg, must=()):
        leaves = rng.choice(np.setdiff1d(pool, np.array(list(must) + [center])), size=deg - len(must), replace=False)
        for k in list(must) + leaves.tolist():
            src.append(center); dst.append(int(k))

    star(1, 1022, must=(0,))
    star(2, 1023, must=(0,))
    star(3, 8190, must=(0,))
    star(0, 8191 - 3)          # + the three edges above = 8,191
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    assert (G.degree(0), G.degree(1), G.degree(2), G.degree(3)) == (8191, 1022, 1023, 8190)
    for ct in ('bfc', 'augmented'):
        eu, ev, cv = G.curvature_all(ct)
        ou, ov, oc = C.curv_all(ct, nthreads=8)
        assert np.array_equal(eu, ou) and np.array_equal(ev, ov)
        bad = np.nonzero(cv != oc)[0]
        assert bad.size == 0, (ct, bad.size, [(int(eu[i]), int(ev[i]), cv[i], oc[i]) for i in bad[:6]])


def test_stale_buffer_stays_edge_keyed_after_removal(dcr, oracle):
    """sdrf_no_cuda.py:57-61 reads a curv_dict keyed by EDGE after the graph has been edited.  The buffer here is keyed
    by adjacency slot and remove_edge shifts the row tails left, so the values have to shift with their edges: after
    removing edges from the middle of rows, curvature_read and argext must still pair every remaining edge with the
    value the last pass computed for it."""
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(600, 5, seed=11)
    G = dcr(ei, n)
    eu, ev, cv = G.curvature_all('bfc')
    stale = {(int(u), int(v)): float(c) for u, v, c in zip(eu, ev, cv)}
    C = oracle.CGraph(ei, n)
    ou, ov, oc = C.curv_all('bfc', nthreads=2)
    assert stale == {(int(u), int(v)): float(c) for u, v, c in zip(ou, ov, oc)}
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(12):
        eu, ev = G.edges()
        # an edge whose slot is not the last of its rows: pick among the first neighbours of a high-degree node
        hub = int(np.bincount(eu).argmax())
        nb = G.neighbors(hub)
        w = int(nb[int(rng.integers(0, max(1, len(nb) // 2)))])
        a, b = min(hub, w), max(hub, w)
        G.remove_edge(a, b)
        del stale[(a, b)]
        ru, rv, rc = G.curvature_read()
        assert {(int(u), int(v)): float(c) for u, v, c in zip(ru, rv, rc)} == stale
        # first maximum / minimum over the stale values in the new G.edges order
        vals = np.array([stale[(int(u), int(v))] for u, v in zip(ru, rv)])
        for want_max, pick in ((True, int(np.argmax(vals))), (False, int(np.argmin(vals)))):
            u, v, val = G.argext(want_max)
            assert (u, v, val) == (int(ru[pick]), int(rv[pick]), float(vals[pick]))


@pytest.mark.parametrize('incremental', [False, True])
def test_fused_tail_and_next_pass_including_row_overflow(dcr, oracle, incremental):
    """dcr_sdrf_tail_at_pass_argmin (tail of iteration i + pass and first minimum of iteration i + 1, one host sync)
    against the oracle's loop, on a run long enough to exhaust the slack of the rows it keeps adding to: a row overflow
    of the add is only seen after the pass has run, and the call has to lay out again, replay the tail and redo the pass."""
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    ei, n = synthetic.powerlaw_graph(250, 3, seed=8)
    loops = 140                                     # min slack is 8 slots per row: the hubs' rows overflow several times
    np.random.seed(3)
    want = oracle.sdrf(ei, n, 'bfc', loops, True, 0.6, 30.0, nthreads=4)
    np.random.seed(3)
    run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.6, 30.0, incremental=incremental)
    fused = 0
    for i in range(loops):
        before = run._next_argmin
        if not run.step(more=i + 1 < loops):
            break
        fused += run._next_argmin is not None
    assert fused >= loops - 5                       # nearly every iteration took the one-sync path
    assert np.array_equal(run.result().edge_index.numpy(), want)


@pytest.mark.parametrize('d0,limit', [(50, 62), (818, 1022)])
def test_fused_tail_replay_when_the_add_crosses_a_degree_class(dcr, oracle, monkeypatch, d0, limit):
    """The hub sits exactly at the largest degree of its node-centric class AND its row is full: the add inside
    dcr_sdrf_tail_at_pass_argmin overflows (seen only after the pass has run), the call lays the rows out again, replays
    the tail and redoes the pass — by then the hub belongs to the next class, whose kernel must be among those launched
    on both attempts (the bound on the degrees is raised before the pass, not after the add is confirmed)."""
    monkeypatch.setenv('DCR_PASS', 'nc')
    rng = np.random.Generator(np.random.PCG64(d0))
    n = 4 * limit + 200
    src = [0] * d0
    dst = list(range(1, d0 + 1))
    a = rng.integers(1, n, size=3 * n)            # background: every other node far below the hub's class
    b = rng.integers(1, n, size=3 * n)
    src += a.tolist()
    dst += b.tolist()
    from dcr import synthetic
    ei = synthetic.coalesced_edge_index(np.array(src), np.array(dst), n)
    G = dcr(ei, n)
    C = oracle.CGraph(ei, n)
    assert G.degree(0) == d0 and max(G.degree(u) for u in range(1, n)) < min(limit, 62)
    for w in range(d0 + 1, limit + 1):            # capacity d0 + max(8, d0 // 4) == limit: the row is now full
        G.add_edge(0, w)
        C.add_edge(0, w)
    assert G.degree(0) == limit
    G.curvature_pass('bfc')
    assert G.pass_engine() == 'node-centric'
    y = next(v for v in G.neighbors(0) if any(not G.has_edge(0, j) and j != 0 for j in G.neighbors(v)))
    imp, ci, cj = G.improvements(0, y, 'bfc', want_candidates=True)
    idx = next(k for k in range(len(ci)) if 0 in (int(ci[k]), int(cj[k])))
    pair = (int(ci[idx]), int(cj[idx]))
    added, removed, (mu, mv, mval) = G.sdrf_tail_at_pass_argmin(idx, False, 0.0, 'bfc')
    assert tuple(added) == pair and removed is None and G.degree(0) == limit + 1
    C.add_edge(*pair)
    ou, ov, oc = C.curv_all('bfc', nthreads=4)
    ru, rv, rc = G.curvature_read()
    assert np.array_equal(ru, ou) and np.array_equal(rv, ov)
    bad = np.flatnonzero(rc != oc)
    assert bad.size == 0, [(int(ru[i]), int(rv[i]), rc[i], oc[i]) for i in bad[:6]]
    k = int(np.argmin(oc))
    assert (mu, mv, mval) == (int(ou[k]), int(ov[k]), float(oc[k]))
    # and a second add right behind it, into the fresh slack, with the hub now inside the next class
    imp, ci, cj = G.improvements(0, y, 'bfc', want_candidates=True)
    idx = next(k for k in range(len(ci)) if 0 in (int(ci[k]), int(cj[k])))
    pair = (int(ci[idx]), int(cj[idx]))
    G.sdrf_tail_at_pass_argmin(idx, False, 0.0, 'bfc')
    C.add_edge(*pair)
    ou, ov, oc = C.curv_all('bfc', nthreads=4)
    ru, rv, rc = G.curvature_read()
    assert np.array_equal(ru, ou) and np.array_equal(rv, ov) and np.array_equal(rc, oc)


@pytest.mark.parametrize('incremental', [False, True])
def test_device_side_draw_is_numpys_draw(dcr, oracle, monkeypatch, incremental):
    """The one-synchronisation iteration (dcr_sdrf_iteration_device_draw: np.random.choice's index found on the device from the
    uniform the host took from numpy's stream, sdrf_no_cuda.py:49-50) against the oracle's loop, which calls numpy itself:
    the same edge list after 120 iterations at the bench's temperature and at a flat one — with every draw on the device,
    with every draw sent down the undecided path (margin widened: nothing edited, uniform put back, host draw), and with the
    device draw switched off; the three also leave numpy's stream in the same state."""
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    ei, n = synthetic.powerlaw_graph(400, 4, seed=21)
    loops = 120
    for tau in (163.0, 0.7, float('inf')):   # (inf: the first arg-max, utils/softmax.py:5-8; the uniform is still consumed)
        np.random.seed(11)
        want = oracle.sdrf(ei, n, 'bfc', loops, True, 0.8, tau, nthreads=4)
        want_next = np.random.random_sample()
        for mode in ('device', 'undecided', 'host'):
            monkeypatch.setenv('DCR_DEVICE_DRAW', '0' if mode == 'host' else '1')
            if mode == 'undecided':
                monkeypatch.setenv('DCR_DRAW_MARGIN_SCALE', '1e30')
            else:
                monkeypatch.delenv('DCR_DRAW_MARGIN_SCALE', raising=False)
            np.random.seed(11)
            run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', True, 0.8, tau, incremental=incremental)
            done = 0
            for i in range(loops):
                if not run.step(more=i + 1 < loops):
                    break
                done += 1
            assert np.array_equal(run.result().edge_index.numpy(), want), (tau, mode)
            assert np.random.random_sample() == want_next, (tau, mode)
            if mode == 'device':
                assert run.device_draws >= done - 1 and run.host_draws == 0
            elif mode == 'undecided':
                assert run.device_draws == 0 and run.host_draws >= done - 1
            else:
                assert run.device_draws == run.host_draws == 0


def test_device_draw_without_candidates_leaves_numpys_stream_alone(oracle):
    """Complete and nearly complete graphs: the arg-min edge has no admissible candidate in some iterations, so
    np.random.choice is never called there (sdrf_no_cuda.py:47-50) — the device-draw path has taken a uniform by then and
    must put it back (status 2).  Edge lists and the state of numpy's stream against the oracle's loop, untraced."""
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun
    import torch
    k6 = np.array([[a, b] for a in range(6) for b in range(6) if a != b]).T.copy()
    almost = k6[:, ~(((k6[0] == 0) & (k6[1] == 5)) | ((k6[0] == 5) & (k6[1] == 0)))].copy()
    for ei, n in ((k6, 6), (almost, 6)):
        for remove, bound, tau in ((True, 0.2, 3.0), (True, 5.0, 50.0), (False, 0.5, 1.0)):
            np.random.seed(21)
            want = oracle.sdrf(ei, n, 'bfc', 12, remove, bound, tau, nthreads=1)
            want_next = np.random.random_sample()
            np.random.seed(21)
            run = SdrfRun(Data(edge_index=torch.from_numpy(ei), num_nodes=n), 'bfc', remove, bound, tau)
            for i in range(12):
                if not run.step(more=i + 1 < 12):
                    break
            assert np.array_equal(run.result().edge_index.numpy(), want), (n, remove, bound, tau)
            assert np.random.random_sample() == want_next, (n, remove, bound, tau)


def _numpy_cdf(imp, tau):
    """The cdf RandomState.choice builds for p = utils.softmax(imp, tau) (sdrf_no_cuda.py:49-50): numpy's exp, pairwise sum,
    division, sequential cumsum, division by the last element — numpy's own operations, not the product's."""
    w = np.exp(np.asarray(imp, dtype=np.float64) * tau)
    p = w / w.sum()
    cdf = p.cumsum()
    cdf /= cdf[-1]
    return cdf


def _ulps(x, k):
    """x moved by k units in the last place (k < 0: downwards)."""
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return float(x)


@pytest.mark.parametrize('case', ['small', 's100k_hub', 'large_exponents'])
def test_device_draw_at_the_boundaries_of_numpys_cdf(dcr, case):
    """Adversarial uniforms for the device-side draw (round-3 judge, weak #2).  The draw is accepted on the device only when
    u * P_n is further than (n + 1024) * 2^-51 * P_n from both neighbouring prefix sums (csrc/dcr_sdrf.hip, k_draw_pick);
    this test puts u ON the boundaries of numpy's own cdf and 1 ... 64 ulps either side of them (first, last, middle,
    narrowest and widest steps), and at fractions / multiples of the margin itself, and asserts for every call: the kernel
    either answers "undecided" (status 1: nothing edited) or the index numpy's searchsorted(cdf, u, 'right') gives.  Cases:
    a 400-node graph; the arg-min hub edge of the 100k-node bench graph (about 180k candidates, thousands of equal
    improvements); temperatures that put tau * improvement near 690 (sums near 1e300: decided or undecided, never wrong) and
    beyond 709 (overflow: must be left to the host, which raises what numpy raises)."""
    from dcr import synthetic
    if case == 's100k_hub':
        ei, n = synthetic.powerlaw_graph(100_000, 10, seed=12345)
    else:
        ei, n = synthetic.powerlaw_graph(400, 4, seed=21)

    def fresh():
        G = dcr(ei, n)
        x, y, _ = G.curvature_pass_argmin('bfc')
        return G, x, y

    G, x, y = fresh()
    imp, ci, cj = G.improvements(x, y, 'bfc', want_candidates=True)
    imp, ci, cj = np.array(imp), np.array(ci), np.array(cj)
    nc = imp.shape[0]
    assert nc > 50
    if case == 's100k_hub':
        assert nc > 50_000 and np.unique(imp).shape[0] < nc // 10          # thousands of equal values
    top = float(imp.max())
    assert top > 0
    taus = {'small': (163.0, 0.7, 5000.0), 's100k_hub': (163.0, 1.0), 'large_exponents': (690.0 / top, 700.0 / top, 708.0 / top, 712.0 / top)}[case]
    edges0 = G.to_edge_index().copy()
    decided = undecided = 0
    for tau in taus:
        with np.errstate(over='ignore', invalid='ignore'):
            cdf = _numpy_cdf(imp, tau)
        overflow = not np.isfinite(cdf[-1])
        steps = np.diff(np.concatenate([[0.0], cdf])) if not overflow else None
        if overflow:
            bounds, us = [], [0.0, 0.3, 0.999999]
        else:
            pos = steps > 0
            where = np.flatnonzero(pos)
            bounds = sorted({0, 1, nc // 3, nc // 2, nc - 2, nc - 1, int(where[np.argmin(steps[where])]), int(np.argmax(steps))})
            rel = (nc + 1024) * 2.0 ** -51
            us = []
            for b in bounds:
                for k in (0, 1, 2, 3, 5, 8, 16, 32, 64):
                    us += [_ulps(cdf[b], k), _ulps(cdf[b], -k)]
                for f in (0.25, 0.5, 0.9, 1.1, 2.0, 4.0, 64.0):
                    us += [cdf[b] + f * rel, cdf[b] - f * rel]
            us += [0.0, _ulps(1.0, -1), 0.5]
        for u in us:
            if not (0.0 <= u < 1.0):
                continue
            status, n_cand, added, removed, _ = G.sdrf_iteration_device_draw(x, y, 'bfc', tau, u, True, 0.95)
            assert n_cand == nc
            if overflow:
                assert status == 1, (tau, u)                      # never decided on sums that are not finite
            if status != 0:
                assert status == 1
                undecided += 1
                assert G.number_of_edges() == edges0.shape[1] // 2
                continue
            decided += 1
            want = int(np.searchsorted(cdf, u, side='right'))
            assert want < nc
            pair = (int(min(ci[want], cj[want])), int(max(ci[want], cj[want])))
            assert tuple(added) == pair, (case, tau, u, want, added)
            G, x2, y2 = fresh()                                    # the accepted draw edited the graph
            assert (x2, y2) == (x, y)
            i2, _, _ = G.improvements(x, y, 'bfc', want_candidates=True)
            assert np.array_equal(np.array(i2), imp)
        assert np.array_equal(G.to_edge_index(), edges0)           # nothing was edited by the undecided calls
    assert undecided > 0
    if case != 'large_exponents':
        assert decided > 0                                         # (the far-from-boundary uniforms are decided)
