"""Host-side adjacency with networkx's ordering semantics, for the loops that keep their graph on the host
(rewiring/sdrf_cuda_bfc.py:31-33,44-54,69,85,93 use an ``nx.Graph`` / ``nx.DiGraph`` next to the dense device matrices).

Only what that loop touches: nodes 0..n-1, insertion-ordered adjacency, ``neighbors`` / ``successors`` /
``predecessors``, ``has_edge``, ``add_edge``, ``remove_edge``, ``DiGraph.to_undirected()`` and the edge order of
``torch_geometric.utils.from_networkx`` (PyG 2.0.3, third-party: ``convert_node_labels_to_integers`` re-adds the edges in
``G.edges`` order before ``to_directed().edges`` is listed)."""
import numpy as np


class OrderedGraph:
    directed = False

    def __init__(self, num_nodes):
        self.n = int(num_nodes)
        self.adj = [dict() for _ in range(self.n)]

    def add_edge(self, u, v):
        self.adj[u].setdefault(v, None)
        self.adj[v].setdefault(u, None)

    def remove_edge(self, u, v):
        del self.adj[u][v]
        if u != v:
            del self.adj[v][u]

    def has_edge(self, u, v):
        return v in self.adj[u]

    def neighbors(self, u):
        return list(self.adj[u])

    def number_of_edges(self):
        return sum(len(a) for a in self.adj) // 2

    def to_edge_index(self):
        relabelled = OrderedGraph(self.n)
        done = set()
        for u in range(self.n):
            for v in self.adj[u]:
                if v not in done:
                    relabelled.add_edge(u, v)
            done.add(u)
        pairs = [(u, v) for u in range(self.n) for v in relabelled.adj[u]]
        return np.array(pairs, dtype=np.int64).reshape(-1, 2).T.copy()


class OrderedDiGraph:
    directed = True

    def __init__(self, num_nodes):
        self.n = int(num_nodes)
        self.succ = [dict() for _ in range(self.n)]
        self.pred = [dict() for _ in range(self.n)]

    def add_edge(self, u, v):
        self.succ[u].setdefault(v, None)
        self.pred[v].setdefault(u, None)

    def remove_edge(self, u, v):
        del self.succ[u][v]
        del self.pred[v][u]

    def has_edge(self, u, v):
        return v in self.succ[u]

    def successors(self, u):
        return list(self.succ[u])

    def predecessors(self, u):
        return list(self.pred[u])

    def to_undirected(self):
        G = OrderedGraph(self.n)
        for u in range(self.n):
            for v in self.succ[u]:
                G.add_edge(u, v)
        return G

    def to_edge_index(self):
        pairs = [(u, v) for u in range(self.n) for v in self.succ[u]]
        return np.array(pairs, dtype=np.int64).reshape(-1, 2).T.copy()


def digraph_from_edge_index(edge_index, num_nodes):
    """``to_networkx(data)``: a DiGraph holding every (u, v) of ``edge_index`` in order (sdrf_cuda_bfc.py:31)."""
    G = OrderedDiGraph(num_nodes)
    ei = np.asarray(edge_index)
    for u, v in zip(ei[0].tolist(), ei[1].tolist()):
        G.add_edge(u, v)
    return G
