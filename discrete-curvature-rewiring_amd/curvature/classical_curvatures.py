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
    if isinstance(G, DcrGraph):
        return G
    # networkx-like: nodes must be 0..n-1; adjacency order is taken as is
    import numpy as np
    n = G.number_of_nodes()
    src, dst = [], []
    seen = set()
    for u in G.nodes:
        for v in G.adj[u]:
            if v not in seen:
                # (max, min) so that the row order of both endpoints follows G's insertion order
                src.append(max(u, v))
                dst.append(min(u, v))
        seen.add(u)
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
