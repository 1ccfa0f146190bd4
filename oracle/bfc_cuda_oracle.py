"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates the reference's DENSE float32 Balanced Forman path — the numerics ``rewire('bfc')`` runs in the reference
(rewiring/rewire.py:8-10) and that differ from curvature/bfc_naive.py:

  * ``balanced_forman_curvature``      curvature/bfc_cuda.py:11-48 (kernel) and :51-65 (host wrapper)
  * ``balanced_forman_post_delta``     curvature/bfc_cuda.py:68-141 and :144-159
  * ``sdrf_cuda_bfc``                  rewiring/sdrf_cuda_bfc.py:14-93, undirected and directed
  * ``OrderedGraph`` / ``OrderedDiGraph``  the slice of networkx the loop uses (insertion-ordered adjacency,
                                       successors / predecessors, ``edges`` order) and PyG 2.0.3 ``to_networkx`` /
                                       ``from_networkx`` / ``to_undirected`` as called at sdrf_cuda_bfc.py:25-33,93

Arithmetic: numba types these kernels with float32 arrays and int64 literals, which unify to float64: every expression
is evaluated in float64 on float32-valued inputs and rounded to float32 where it is stored into C / D (the base
expression, then ``+=`` of the 4-cycle term).  Plain Python floats do exactly that here.

Parity pin: bit-for-bit against tests/golden/bfc_cuda_{curvature,sdrf}.json, which tools/make_golden_cuda_compat.py
recorded by executing the reference's two files unmodified through a numba.cuda stand-in (tests/test_oracle_golden.py).
Only tests/ may import this module.
"""
import numpy as np


def _f32(v):
    return float(np.float32(v))


def _closing(d_max, d_min, a2_xy, a_xy, sharp, lam):
    # bfc_cuda.py:46-48 / :139-141, Python precedence, float64, two float32 stores
    c = _f32((2 / d_max) + (2 / d_min) - 2 + (2 / d_max + 1 / d_min) * a2_xy * a_xy)
    if lam > 0:
        c = _f32(c + sharp / (d_max * lam))
    return c


def balanced_forman_curvature(A):
    """A: [N, N] array of 0/1 (float32 semantics).  Returns C float32 [N, N] (bfc_cuda.py:11-65)."""
    A = np.asarray(A, dtype=np.float64)
    N = A.shape[0]
    A2 = A @ A
    d_in, d_out = A.sum(axis=0), A.sum(axis=1)
    C = np.zeros((N, N), dtype=np.float32)
    for i, j in zip(*np.nonzero(A)):
        if d_in[i] > d_out[j]:
            d_max, d_min = float(d_in[i]), float(d_out[j])
        else:
            d_max, d_min = float(d_out[j]), float(d_in[i])
        if d_max * d_min == 0:
            continue
        t1 = A[:, j] * (A2[i, :] - A[i, :]) * A[i, j]
        t2 = A[i, :] * (A2[:, j] - A[:, j]) * A[i, j]
        sharp = int((t1 > 0).sum() + (t2 > 0).sum())
        lam = max(0.0, float(t1.max(initial=0.0)), float(t2.max(initial=0.0)))
        C[i, j] = _closing(d_max, d_min, float(A2[i, j]), float(A[i, j]), sharp, lam)
    return C


def balanced_forman_post_delta(A, x, y, i_neighbors, j_neighbors):
    """D[I, J] = curvature of (x, y) after adding (i_I, j_J); -1000 where i == j or the pair is an edge
    (bfc_cuda.py:68-159)."""
    A = np.asarray(A, dtype=np.float64)
    N = A.shape[0]
    A2 = A @ A
    d_in_x0, d_out_y0 = float(A[:, x].sum()), float(A[y].sum())
    D = np.zeros((len(i_neighbors), len(j_neighbors)), dtype=np.float32)
    z = np.arange(N)
    for I, i in enumerate(i_neighbors):
        for J, j in enumerate(j_neighbors):
            if i == j or A[i, j] != 0:
                D[I, J] = -1000
                continue
            d_in_x, d_out_y = d_in_x0, d_out_y0
            if j == x:
                d_in_x += 1
            elif i == y:
                d_out_y += 1
            if d_in_x * d_out_y == 0:
                D[I, J] = 0
                continue
            if d_in_x > d_out_y:
                d_max, d_min = d_in_x, d_out_y
            else:
                d_max, d_min = d_out_y, d_in_x
            a2_xy = float(A2[x, y])
            if x == i and A[j, y] != 0:
                a2_xy += A[j, y]
            elif y == j and A[x, i] != 0:
                a2_xy += A[x, i]
            A_z_y, A_x_z = A[:, y].copy(), A[x, :].copy()
            A2_z_y, A2_x_z = A2[:, y].copy(), A2[x, :].copy()
            if y == j:
                A_z_y[i] += 1
            if x == i:
                A_x_z[j] += 1
            if A[j, y] != 0:
                A2_z_y[i] += A[j, y]
            if x == i:
                A2_x_z += np.where(A[j, :] != 0, A[j, :], 0.0)
            if y == j:
                A2_z_y += np.where(A[:, i] != 0, A[:, i], 0.0)
            if A[x, i] != 0:
                A2_x_z[j] += A[x, i]
            t1 = A_z_y * (A2_x_z - A_x_z) * A[x, y]
            t2 = A_x_z * (A2_z_y - A_z_y) * A[x, y]
            sharp = int((t1 > 0).sum() + (t2 > 0).sum())
            lam = max(0.0, float(t1.max(initial=0.0)), float(t2.max(initial=0.0)))
            D[I, J] = _closing(d_max, d_min, a2_xy, float(A[x, y]), sharp, lam)
    del z
    return D


# ---- the slice of networkx / PyG the loop relies on -------------------------------------------------------------------
class OrderedGraph:
    """nx.Graph on nodes 0..n-1: adjacency dicts in insertion order."""
    directed = False

    def __init__(self, n):
        self.n = n
        self.adj = [dict() for _ in range(n)]

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

    def edge_index(self):
        """from_networkx: convert_node_labels_to_integers re-adds the edges in G.edges order, then to_directed().edges."""
        H = OrderedGraph(self.n)
        seen = set()
        for u in range(self.n):
            for v in self.adj[u]:
                if v not in seen:
                    H.add_edge(u, v)
            seen.add(u)
        return np.array([[u, v] for u in range(self.n) for v in H.adj[u]], dtype=np.int64).reshape(-1, 2).T


class OrderedDiGraph:
    directed = True

    def __init__(self, n):
        self.n = n
        self.succ = [dict() for _ in range(n)]
        self.pred = [dict() for _ in range(n)]

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
        """nx.DiGraph.to_undirected: edges re-added for u in node order, for v in succ[u]."""
        G = OrderedGraph(self.n)
        for u in range(self.n):
            for v in self.succ[u]:
                G.add_edge(u, v)
        return G

    def edge_index(self):
        return np.array([[u, v] for u in range(self.n) for v in self.succ[u]], dtype=np.int64).reshape(-1, 2).T


def to_undirected(edge_index):
    """PyG to_undirected: symmetrise and coalesce (sorted by row, then column)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    row = np.concatenate([ei[0], ei[1]])
    col = np.concatenate([ei[1], ei[0]])
    n = int(max(row.max(initial=-1), col.max(initial=-1))) + 1
    key = np.unique(row * n + col)
    return np.stack([key // n, key % n])


def softmax(a, tau=1):
    if tau == float('inf'):
        r = np.zeros(len(a))
        r[np.argmax(a)] = 1
        return r
    e = np.exp(a * tau)
    return e / e.sum()


def sdrf_cuda_bfc(edge_index, num_nodes, loops, remove_edges, removal_bound, tau, is_undirected, trace=None):
    """rewiring/sdrf_cuda_bfc.py:14-93.  Returns the int64 [2, M] edge_index of from_networkx(G)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    dense_ei = to_undirected(ei) if is_undirected else ei
    dense_ei = dense_ei[:, dense_ei[0] != dense_ei[1]]            # remove_self_loops (:29)
    N = int(dense_ei.max()) + 1 if dense_ei.size else 0           # to_dense_adj sizes by the largest id
    A = np.zeros((N, N), dtype=np.float64)
    np.add.at(A, (dense_ei[0], dense_ei[1]), 1.0)
    G = OrderedDiGraph(num_nodes)                                  # to_networkx(data): every directed edge, in order
    for u, v in zip(ei[0].tolist(), ei[1].tolist()):
        G.add_edge(u, v)
    if is_undirected:
        G = G.to_undirected()
    C = np.zeros((N, N), dtype=np.float32)
    for _ in range(loops):
        can_add = True
        C = balanced_forman_curvature(A)
        ix_min = int(np.argmin(C))                                 # dense arg-min: zeros of non-edges take part (:40)
        x, y = ix_min // N, ix_min % N
        if is_undirected:
            x_nb, y_nb = G.neighbors(x) + [x], G.neighbors(y) + [y]
        else:
            x_nb, y_nb = G.successors(x) + [x], G.predecessors(y) + [y]
        cand = [(i, j) for i in x_nb for j in y_nb if i != j and not G.has_edge(i, j)]
        rec = {'argmin': [x, y], 'x_neighbors': x_nb, 'y_neighbors': y_nb, 'n_candidates': len(cand), 'improvements': None,
               'choice': None, 'events': []}
        if cand:
            D = balanced_forman_post_delta(A, x, y, x_nb, y_nb)
            cxy = C[x, y]
            imp = [float(np.float32(D[x_nb.index(i), y_nb.index(j)] - cxy)) for i, j in cand]  # float32 subtraction (:61)
            rec['improvements'] = imp
            idx = np.random.choice(range(len(cand)), p=softmax(np.array(imp), tau=tau))
            rec['choice'] = int(idx)
            k, l = cand[idx]
            G.add_edge(k, l)
            rec['events'].append(['add', k, l])
            A[k, l] = 1
            if is_undirected:
                A[l, k] = 1
        else:
            can_add = False
            if not remove_edges:
                if trace is not None:
                    trace.append(rec)
                break
        stop = False
        if remove_edges:
            ix_max = int(np.argmax(C))                             # stale C, nothing excluded (:80)
            x, y = ix_max // N, ix_max % N
            if C[x, y] > removal_bound:
                G.remove_edge(x, y)
                rec['events'].append(['rm', x, y])
                A[x, y] = 0
                if is_undirected:
                    A[y, x] = 0
            elif can_add is False:
                stop = True
        if trace is not None:
            trace.append(rec)
        if stop:
            break
    return G.edge_index()
