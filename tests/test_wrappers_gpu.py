"""The reference-named Python entry points (curvature/bfc_naive.py:7,43; curvature/classical_curvatures.py:6,31) on the
HIP path, called the way the reference's callers call them: on the device graph and on a ``networkx.Graph`` — including
one whose edges were added and removed after construction, so that adjacency order differs from id order — checked
against the Python oracle replaying the same edit sequence and against the KAT fixtures of the reference itself."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _edited_networkx_graph():
    """Same edit sequence on a networkx.Graph and on the oracle's graph."""
    import networkx as nx
    from oracle.sdrf_oracle import OGraph
    rng = np.random.Generator(np.random.PCG64(42))
    n = 60
    G = nx.Graph()
    G.add_nodes_from(range(n))
    O = OGraph(n)
    for _ in range(240):
        u, v = (int(t) for t in rng.integers(0, n, size=2))
        if u != v:
            G.add_edge(u, v)
            O.add_edge(u, v)
    for _ in range(40):
        edges = list(G.edges)
        u, v = edges[int(rng.integers(0, len(edges)))]
        G.remove_edge(u, v)
        O.remove_edge(u, v)
    for _ in range(80):  # late additions: appended to rows that already hold larger ids
        u, v = (int(t) for t in rng.integers(0, n, size=2))
        if u != v:
            G.add_edge(u, v)
            O.add_edge(u, v)
    return G, O


def test_bfc_naive_entry_points_on_kat_fixtures():
    import networkx as nx
    from curvature.bfc_naive import bfc, bfc_edge
    from dcr.graph import DcrGraph
    for k in load_golden('kat_curvature.json')['kat']:
        ei = np.array(k['edge_index'])
        want = float.fromhex(k['bfc'])
        G = DcrGraph(ei, k['num_nodes'])
        assert bfc_edge(G, k['u'], k['v']) == want, k['graph']
        N = nx.Graph()
        N.add_nodes_from(range(k['num_nodes']))
        N.add_edges_from((int(a), int(b)) for a, b in zip(ei[0], ei[1]) if b <= a)
        got = bfc_edge(N, k['u'], k['v'])
        assert got == want, k['graph']
        if want == 0 and min(N.degree(k['u']), N.degree(k['v'])) == 1:
            assert isinstance(got, int)  # bfc_naive.py:18-19 returns the int 0
        bfc(N)
        assert N[k['u']][k['v']]['bfc'] == want, k['graph']


def test_wrappers_follow_networkx_adjacency_order():
    from curvature.bfc_naive import bfc, bfc_edge
    from curvature.classical_curvatures import as_dcr_graph, compute_curvature_edge, compute_curvature_graph
    from oracle import sdrf_oracle as so
    G, O = _edited_networkx_graph()
    n = G.number_of_nodes()
    assert [list(G.adj[u]) for u in range(n)] == [list(O.adj[u]) for u in range(n)]
    assert [list(e) for e in G.edges] == [list(e) for e in O.edges()]
    D = as_dcr_graph(G)
    # every row in networkx's order, hence G.edges order too
    assert [D.neighbors(u) for u in range(n)] == [list(G.adj[u]) for u in range(n)]
    eu, ev = D.edges()
    assert list(zip(eu.tolist(), ev.tolist())) == list(G.edges)
    for ct in ('bfc', '1d', 'augmented', 'haantjes'):
        cd = compute_curvature_graph(G, ct)
        flat = [(u, v, c) for u, row in cd.items() for v, c in row.items()]
        assert [(u, v) for u, v, _ in flat] == list(G.edges), ct         # dict insertion order = G.edges order
        for u, v, c in flat:
            assert c == so.curvature_edge(O, u, v, ct), (ct, u, v)
            if ct != 'bfc':
                assert isinstance(c, int)
        for (u, v) in list(G.edges)[::7]:
            assert compute_curvature_edge(G, (u, v), ct) == so.curvature_edge(O, u, v, ct)
            assert compute_curvature_edge(G, (v, u), ct) == so.curvature_edge(O, v, u, ct)
    # first extrema through this entry: ties resolve in networkx's edge order
    cd = compute_curvature_graph(G, 'bfc')
    x, y = min(G.edges, key=lambda e: cd[e[0]][e[1]])
    D.curvature_pass('bfc')
    assert D.argext(False)[:2] == (x, y)
    x, y = max(G.edges, key=lambda e: cd[e[0]][e[1]])
    assert D.argext(True)[:2] == (x, y)
    bfc(G)
    for u, v in list(G.edges)[::5]:
        assert G[u][v]['bfc'] == so.bfc_edge(O, u, v) == bfc_edge(G, u, v)
    assert compute_curvature_graph(G, 'nope') is None  # classical_curvatures.py:44-46 falls through and returns None
    with pytest.raises(Exception):
        compute_curvature_edge(G, (0, 1), 'nope')


def test_gcn_caches_cannot_alias_a_recycled_tensor():
    """The normalised adjacency and Â·X are cached per (tensor, version).  A tensor freed and re-allocated at the same
    address with the same shape must not hit the entry of the old one."""
    import torch
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models.gcn import GCN, dense_reference_logits
    dev = torch.device('cuda', 0)
    n = 300
    torch.manual_seed(0)
    x = torch.randn(n, 24, device=dev)
    y = torch.randint(0, 4, (n,), device=dev)
    ei1, _ = synthetic.powerlaw_graph(n, 3, seed=1)
    ei2 = ei1.copy()
    ei2[:, :] = synthetic.powerlaw_graph(n, 3, seed=2)[0]   # same shape, different edges
    assert ei1.shape == ei2.shape and not np.array_equal(ei1, ei2)
    model = GCN(Dataset(Data(x=x, edge_index=torch.from_numpy(ei1).to(dev), y=y, num_nodes=n), 4), hidden=[8]).to(dev)
    model.eval()
    seen = set()
    for ei in (ei1, ei2, ei1, ei2):
        t = torch.from_numpy(ei).to(dev)
        seen.add(t.data_ptr())
        data = Data(x=x, edge_index=t, y=y, num_nodes=n)
        got = model(data)
        want = dense_reference_logits(model, x, t, n)
        assert (got.double() - want).abs().max().item() < 1e-5
        del t, data
    # the same for the feature matrix (cached Â·X of the first layer)
    t = torch.from_numpy(ei1).to(dev)
    for s in (1, 2, 3):
        torch.manual_seed(s)
        xs = torch.randn(n, 24, device=dev)
        got = model(Data(x=xs, edge_index=t, y=y, num_nodes=n))
        want = dense_reference_logits(model, xs, t, n)
        assert (got.double() - want).abs().max().item() < 1e-5
        del xs


def test_networkx_graph_is_uploaded_once_and_mirrored(monkeypatch):
    """The reference's improvement loop on a networkx.Graph (sdrf_no_cuda.py:41-46: before, add_edge, after, remove_edge,
    per candidate): one upload, the edits mirrored onto the device copy, values equal to the oracle replaying the same
    edits; a mutation the mirror does not follow (add_node) drops the copy; pickles and deep copies are plain graphs."""
    import copy
    import pickle
    import networkx as nx
    from curvature import classical_curvatures as cc
    from curvature.classical_curvatures import compute_curvature_edge, compute_curvature_graph
    from oracle import sdrf_oracle as so
    G, O = _edited_networkx_graph()
    uploads = []
    real = cc._upload
    monkeypatch.setattr(cc, '_upload', lambda g, device=0: (uploads.append(1), real(g, device))[1])
    x, y = next(iter(G.edges))
    cands = [(i, j) for i in list(G[x]) + [x] for j in list(G[y]) + [y] if i != j and not G.has_edge(i, j)][:40]
    assert cands
    for ct in ('bfc', 'augmented'):
        for (i, j) in cands:
            before = compute_curvature_edge(G, (x, y), ct)
            G.add_edge(i, j)
            O.add_edge(i, j)
            after = compute_curvature_edge(G, (x, y), ct)
            assert after == so.curvature_edge(O, x, y, ct) and (ct == 'bfc' or isinstance(after, int))
            G.remove_edge(i, j)
            O.remove_edge(i, j)
            assert compute_curvature_edge(G, (x, y), ct) == before
    assert len(uploads) == 1 and isinstance(G, nx.Graph)
    # the whole-graph entry on the mirrored copy, after all those edits: networkx's G.edges order and the oracle's values
    d = compute_curvature_graph(G, 'bfc')
    assert [(u, v) for u in d for v in d[u]] == list(G.edges)
    assert all(d[u][v] == so.curvature_edge(O, u, v, 'bfc') for u, v in G.edges)
    assert len(uploads) == 1
    G.add_node(G.number_of_nodes())           # not followed by the mirror: the copy is dropped and rebuilt
    compute_curvature_edge(G, (x, y), 'bfc')
    assert len(uploads) == 2
    assert type(pickle.loads(pickle.dumps(G))) is nx.Graph and type(copy.deepcopy(G)) is nx.Graph
    with pytest.raises(nx.NetworkXError):
        G.remove_edge(x, x + 10**6)
    # an edit written straight into the adjacency (behind the mirroring methods): the endpoint's degree no longer matches the
    # device copy, which is uploaded again instead of answering for the old graph (advisor, round 3)
    i, j = cands[0]
    G._adj[i][j] = {}
    G._adj[j][i] = {}
    O.add_edge(i, j)
    assert compute_curvature_edge(G, (i, j), 'bfc') == so.curvature_edge(O, i, j, 'bfc') and len(uploads) == 3
    del G._adj[i][j], G._adj[j][i]
    O.remove_edge(i, j)
    d = compute_curvature_graph(G, 'bfc')                     # (the whole-graph entry compares the edge count)
    assert len(uploads) == 4 and all(d[u][v] == so.curvature_edge(O, u, v, 'bfc') for u, v in G.edges)
    # DCR_MIRROR=0: no mirror, no change of class
    monkeypatch.setenv('DCR_MIRROR', '0')
    P = nx.Graph(G)
    compute_curvature_edge(P, (x, y), 'bfc')
    compute_curvature_edge(P, (x, y), 'bfc')
    assert type(P) is nx.Graph and len(uploads) == 6


def test_adjacency_that_is_no_undirected_graph_is_refused():
    """A DiGraph's successor rows are not the projections of one undirected edge sequence: ValueError, not IndexError."""
    import networkx as nx
    from curvature.classical_curvatures import as_dcr_graph
    D = nx.DiGraph()
    D.add_nodes_from(range(4))
    D.add_edges_from([(0, 1), (1, 2), (2, 3), (0, 3)])
    with pytest.raises(ValueError):
        as_dcr_graph(D)
