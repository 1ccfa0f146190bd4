"""Balanced Forman curvature — call surface of the reference's curvature/bfc_naive.py.

``bfc_edge(G, v1, v2)`` (bfc_naive.py:7-40) and ``bfc(G)`` (bfc_naive.py:43-52)
with the reference's semantics and float64 results, computed by the HIP
kernels in csrc/dcr_bfc.hip on a device-resident graph.
"""
from curvature.classical_curvatures import as_dcr_graph
from dcr.graph import DcrGraph


def bfc_edge(G, v1, v2):
    g = as_dcr_graph(G, check=(v1, v2))
    if isinstance(G, DcrGraph):
        d1, d2 = g.degree(v1), g.degree(v2)
    else:   # (the device copy has just been checked against these very degrees: no round trip for them)
        a1, a2 = G._adj[v1], G._adj[v2]
        d1, d2 = len(a1) - (v1 in a1), len(a2) - (v2 in a2)
    if min(d1, d2) == 1:
        return 0  # bfc_naive.py:18-19 returns the int 0
    return g.curvature_edge(v1, v2, 'bfc')


def bfc(G):
    """Curvature of every edge.  For a networkx graph the values are written to
    ``G[v1][v2]['bfc']`` as the reference does; a DcrGraph gets a ``.bfc`` dict
    keyed (v1, v2) in ``G.edges`` order."""
    g = as_dcr_graph(G, check='all')
    eu, ev, cv = g.curvature_all('bfc')
    if isinstance(G, DcrGraph):
        G.bfc = {(u, v): c for u, v, c in zip(eu.tolist(), ev.tolist(), cv.tolist())}
    else:
        for u, v, c in zip(eu.tolist(), ev.tolist(), cv.tolist()):
            G[u][v]['bfc'] = c
    return G
