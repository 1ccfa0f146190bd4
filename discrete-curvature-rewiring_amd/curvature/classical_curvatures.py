"""Classical discrete curvatures on the MI355X graph container.

Call surface of the reference's curvature/classical_curvatures.py:6-46
(``compute_curvature_edge(G, e, curv_type)``, ``compute_curvature_graph(G,
curv_type)``), extended with ``'bfc'`` so that rewiring/sdrf_no_cuda.py can run
Balanced Forman curvature through the same two functions.  ``G`` is a
``dcr.DcrGraph`` (device-resident); a ``networkx.Graph`` is accepted and
uploaded first.
"""
from dcr.graph import DcrGraph, curv_code

CURV_TYPES = ('1d', 'augmented', 'haantjes', 'bfc')


def as_dcr_graph(G, device=0):
    """``G`` as a device-resident graph whose adjacency rows are in the SAME order as ``G.adj[u]`` for every node, so
    that ``G.edges`` order, and with it every first-extremum tie-break (sdrf_no_cuda.py:27,59,61), is networkx's.

    networkx appends v to adj[u] and u to adj[v] in one ``add_edge``, so the rows of any real graph are the projections
    of one edge-insertion sequence; that sequence is recovered here by merging the rows (an edge is emitted once it is
    the next one in the rows of both its endpoints) and replayed into the container.  Node labels must be 0..n-1 in
    insertion order, as curvature/bfc_naive.py:34-37 (positional rows of ``nx.adj_matrix``) needs them too.
    """
    if isinstance(G, DcrGraph):
        return G
    import numpy as np
    n = G.number_of_nodes()
    if list(G.nodes) != list(range(n)):
        raise ValueError('node labels must be 0..n-1 in insertion order')
    rows = [[v for v in G.adj[u] if v != u] for u in range(n)]  # (self-loops carry no curvature; none in the reference's use)
    nxt = [0] * n
    src, dst = [], []
    stack = list(range(n - 1, -1, -1))
    while stack:
        u = stack.pop()
        while nxt[u] < len(rows[u]):
            v = rows[u][nxt[u]]
            if rows[v][nxt[v]] != u:
                break           # (u, v) waits for earlier edges of v; v's turn will come back to it
            nxt[u] += 1
            nxt[v] += 1
            src.append(max(u, v))
            dst.append(min(u, v))
            stack.append(v)     # v's next edge may now be ready
    if any(nxt[u] != len(rows[u]) for u in range(n)):
        raise ValueError('adjacency rows are not the projections of one edge sequence (not a networkx.Graph?)')
    return DcrGraph(np.array([src, dst], dtype=np.int64).reshape(2, -1), n, device=device)


def _num(v, curv_type):
    # the reference returns Python ints for the classical kinds (classical_curvatures.py:16-27)
    return v if curv_type == 'bfc' else int(v)


def compute_curvature_edge(G, e, curv_type):
    """classical_curvatures.py:6-28."""
    if curv_type not in CURV_TYPES:
        raise Exception(f'Method {curv_type} not available.')
    v1, v2 = e
    return _num(as_dcr_graph(G).curvature_edge(v1, v2, curv_type), curv_type)


def compute_curvature_graph(G, curv_type):
    """classical_curvatures.py:31-46: dict of dicts keyed [v1][v2] in ``G.edges`` orientation."""
    if curv_type not in CURV_TYPES:
        return None  # the reference falls through an ``assert True`` and returns None
    g = as_dcr_graph(G)
    eu, ev, cv = g.curvature_all(curv_type)
    curv_dict = {}
    for u, v, c in zip(eu.tolist(), ev.tolist(), cv.tolist()):
        curv_dict.setdefault(u, {})[v] = _num(c, curv_type)
    return curv_dict
