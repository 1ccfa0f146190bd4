"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-Python restatement of the reference's Balanced Forman curvature and
SDRF rewiring loop, used only as the checker by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing
under ``discrete-curvature-rewiring_amd/`` may import it.

Parity pin: this file is checked bit-for-bit against fixtures produced by
running the reference itself (``tools/make_golden.py`` → ``tests/golden/``):
known-answer curvatures, whole-graph passes, and per-iteration SDRF traces.

What it restates (file:line into the reference):
  * ``bfc_edge``                 curvature/bfc_naive.py:7-40
  * ``classical_edge``           curvature/classical_curvatures.py:6-28
  * ``softmax``                  utils/softmax.py:4-10
  * ``sdrf``                     rewiring/sdrf_no_cuda.py:9-68
  * ``OGraph``                   the slice of networkx.Graph semantics the loop
                                 relies on: insertion-
                                 ordered adjacency, ``edges`` enumeration with
                                 a seen-set, append on add, in-place delete
  * ``OGraph.from_edge_index`` / ``to_edge_index``
                                 torch_geometric 2.0.3 ``to_networkx(...,
                                 to_undirected=True)`` / ``from_networkx``
                                 (third-party, not vendored in the reference;
                                 call sites rewiring/sdrf_no_cuda.py:20,68)
"""
import numpy as np


class OGraph:
    """Undirected simple graph, nodes 0..n-1, adjacency in insertion order."""

    def __init__(self, n):
        self.n = int(n)
        self.adj = [dict() for _ in range(self.n)]

    @classmethod
    def from_edge_index(cls, edge_index, num_nodes):
        # to_networkx(to_undirected=True): walk edge_index in order and keep
        # only pairs with v <= u (PyG 2.0.3 skips ``v > u``).
        g = cls(num_nodes)
        ei = np.asarray(edge_index)
        for u, v in zip(ei[0].tolist(), ei[1].tolist()):
            if v > u:
                continue
            if u == v:
                raise ValueError("self-loops are outside the SDRF boundary contract")
            g.add_edge(u, v)
        return g

    def add_edge(self, u, v):
        if v not in self.adj[u]:
            self.adj[u][v] = None
            self.adj[v][u] = None

    def remove_edge(self, u, v):
        del self.adj[u][v]
        del self.adj[v][u]

    def has_edge(self, u, v):
        return v in self.adj[u]

    def degree(self, u):
        return len(self.adj[u])

    def neighbors(self, u):
        return list(self.adj[u])

    def edges(self):
        # networkx EdgeView over an undirected graph: for n in node order, for
        # nbr in adj[n] order, yield (n, nbr) unless nbr was already an outer node.
        seen = set()
        for n in range(self.n):
            for nbr in self.adj[n]:
                if nbr not in seen:
                    yield (n, nbr)
            seen.add(n)

    def num_edges(self):
        return sum(len(a) for a in self.adj) // 2

    def to_edge_index(self):
        # from_networkx: convert_node_labels_to_integers rebuilds the graph by adding edges in
        # G.edges order, so row u holds first its smaller neighbours (ascending: one edge per
        # earlier outer node), then its larger ones in adj[u] order; to_directed().edges then
        # walks rows in node order.
        src, dst = [], []
        for u in range(self.n):
            row = sorted(v for v in self.adj[u] if v < u) + [v for v in self.adj[u] if v > u]
            for v in row:
                src.append(u)
                dst.append(v)
        return np.array([src, dst], dtype=np.int64).reshape(2, -1)


def bfc_formula(d1, d2, T, s1, s2, gamma):
    """The closing float64 expression of bfc_naive.py:31-32 / :39-40, in the
    reference's left-to-right order.  ``s1 == 0 or s2 == 0`` selects the
    no-4-cycle branch (bfc_naive.py:30)."""
    dmax, dmin = max(d1, d2), min(d1, d2)
    if s1 == 0 or s2 == 0:
        return 2 / d1 + 2 / d2 - 2 + 2 * T / dmax + T / dmin
    return 2 / d1 + 2 / d2 - 2 + 2 * T / dmax + T / dmin + 1 / np.int64(gamma) / dmax * (s1 + s2)


def bfc_ingredients(G, v1, v2):
    """Integer ingredients of bfc_edge: (d1, d2, T, |sq1|, |sq2|, gamma)."""
    d1, d2 = G.degree(v1), G.degree(v2)
    S1, S2 = set(G.adj[v1]), set(G.adj[v2])
    T = len(S1 & S2)
    # bfc_naive.py:26-29
    sq1 = [k for k in S1 - S2 if k != v2 and (set(G.adj[k]) & S2) - (S1 | {v1})]
    sq2 = [k for k in S2 - S1 if k != v1 and (set(G.adj[k]) & S1) - (S2 | {v2})]
    if len(sq1) == 0 or len(sq2) == 0:
        return d1, d2, T, len(sq1), len(sq2), 0
    # bfc_naive.py:36-37: A[k] @ (A[v2] - A[v1]*A[v2]) = |N(k) & (N(v2) - N(v1))|, minus one for v1
    only2, only1 = S2 - S1, S1 - S2
    gamma = max(max(len(set(G.adj[k]) & only2) - 1 for k in sq1),
                max(len(set(G.adj[k]) & only1) - 1 for k in sq2))
    return d1, d2, T, len(sq1), len(sq2), gamma


def bfc_edge(G, v1, v2):
    """curvature/bfc_naive.py:7-40."""
    if min(G.degree(v1), G.degree(v2)) == 1:
        return 0  # bfc_naive.py:18-19 (an int, as in the reference)
    return bfc_formula(*bfc_ingredients(G, v1, v2))


def classical_edge(G, v1, v2, curv_type):
    """curvature/classical_curvatures.py:14-28."""
    if curv_type == '1d':
        return 4 - G.degree(v1) - G.degree(v2)
    T = len(set(G.adj[v1]) & set(G.adj[v2]))
    if curv_type == 'augmented':
        return 4 - G.degree(v1) - G.degree(v2) + 3 * T
    if curv_type == 'haantjes':
        return T
    raise Exception(f'Method {curv_type} not available.')


def curvature_edge(G, v1, v2, curv_type):
    if curv_type == 'bfc':
        return bfc_edge(G, v1, v2)
    return classical_edge(G, v1, v2, curv_type)


def curvature_graph(G, curv_type):
    """classical_curvatures.py:38-46 (plus the 'bfc' composition): values in G.edges order."""
    return {(v1, v2): curvature_edge(G, v1, v2, curv_type) for (v1, v2) in G.edges()}


def softmax(a, tau=1):
    """Restates utils/softmax.py:4-10: for tau = inf the indicator of the first arg-max, else exp(a * tau) over its
    (pairwise, numpy) sum; no max-subtraction, as in the reference."""
    if tau == float('inf'):
        indicator = np.zeros(len(a))
        indicator[np.argmax(a)] = 1
        return indicator
    weights = np.exp(a * tau)
    return weights / weights.sum()


def sdrf(edge_index, num_nodes, curv_type, loops, remove_edges, removal_bound, tau, trace=None):
    """rewiring/sdrf_no_cuda.py:9-68.  Consumes the global legacy numpy RNG
    exactly as the reference does (one ``np.random.choice`` per iteration that
    has candidates).  ``trace`` (a list) receives one dict per iteration."""
    G = OGraph.from_edge_index(edge_index, num_nodes)
    for _ in range(loops):
        can_add = True
        curv = curvature_graph(G, curv_type)
        rec = {}
        # first minimum in G.edges order (sdrf_no_cuda.py:27)
        x, y = min(G.edges(), key=lambda e: curv[e])
        rec['argmin'] = [x, y]
        x_nb = G.neighbors(x) + [x]
        y_nb = G.neighbors(y) + [y]
        candidates = []
        for i in x_nb:
            for j in y_nb:
                if i != j and not G.has_edge(i, j):
                    candidates.append(sorted((i, j)))
        rec['candidates'] = [list(c) for c in candidates]
        k = l = None
        if len(candidates):
            improvements = []
            for (i, j) in candidates:
                before = curvature_edge(G, x, y, curv_type)
                G.add_edge(i, j)
                after = curvature_edge(G, x, y, curv_type)
                improvements.append(after - before)
                G.remove_edge(i, j)
            rec['improvements'] = [float(v) for v in improvements]
            idx = np.random.choice(range(len(candidates)), p=softmax(np.array(improvements), tau=tau))
            rec['choice'] = int(idx)
            k, l = sorted(candidates[idx])
            G.add_edge(k, l)
            rec['added'] = [k, l]
        else:
            rec['improvements'] = []
            rec['choice'] = None
            rec['added'] = None
            can_add = False
            if not remove_edges:
                rec['removed'] = None
                if trace is not None:
                    trace.append(rec)
                break
        rec['removed'] = None
        stop = False
        if remove_edges:
            # stale curvatures, first maximum, the new edge excluded (sdrf_no_cuda.py:57-61)
            if len(candidates):
                x, y = max([e for e in G.edges() if e != (k, l)], key=lambda e: curv[e])
            else:
                x, y = max(list(G.edges()), key=lambda e: curv[e])
            if curv[(x, y)] > removal_bound:
                G.remove_edge(x, y)
                rec['removed'] = [x, y]
            elif can_add is False:
                stop = True
        if trace is not None:
            trace.append(rec)
        if stop:
            break
    return G.to_edge_index()
