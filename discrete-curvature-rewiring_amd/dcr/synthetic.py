"""Synthetic graph generators owned by this build.

The bench and the parity fixtures must not depend on networkx's generators
(their streams drift between versions), so the power-law graph comes from a
preferential-attachment generator written here on top of numpy's PCG64.
"""
import numpy as np


def coalesced_edge_index(src, dst, num_nodes=None):
    """Symmetrise, drop duplicates and self-loops, sort by (row, col).

    Returns int64 [2, 2E]: the layout PyG's ``coalesce`` gives and the one the
    SDRF boundary contract assumes (initial adjacency lists ascending by id).
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if num_nodes is None:
        num_nodes = int(max(src.max(initial=-1), dst.max(initial=-1))) + 1
    a = np.concatenate([src, dst])
    b = np.concatenate([dst, src])
    key = np.unique(a * np.int64(num_nodes) + b)
    return np.stack([key // num_nodes, key % num_nodes]).astype(np.int64)


def powerlaw_graph(n, m, seed=12345):
    """Preferential attachment: node s >= m attaches to m distinct earlier
    nodes drawn in proportion to degree.  E = m * (n - m) exactly.

    Returns (edge_index int64 [2, 2E] coalesced+sorted, num_nodes).
    """
    if not (1 <= m < n):
        raise ValueError("need 1 <= m < n")
    rng = np.random.Generator(np.random.PCG64(seed))
    n_e = m * (n - m)
    src = np.empty(n_e, dtype=np.int64)
    dst = np.empty(n_e, dtype=np.int64)
    rep = np.empty(2 * n_e, dtype=np.int64)  # every endpoint once per incident edge
    fill = 0
    e = 0
    targets = np.arange(m, dtype=np.int64)
    for s in range(m, n):
        src[e:e + m] = s
        dst[e:e + m] = targets
        e += m
        rep[fill:fill + m] = targets
        rep[fill + m:fill + 2 * m] = s
        fill += 2 * m
        if s + 1 == n:
            break
        chosen = []
        seen = set()
        while len(chosen) < m:
            for t in rep[rng.integers(0, fill, size=2 * m)].tolist():
                if t not in seen:
                    seen.add(t)
                    chosen.append(t)
                    if len(chosen) == m:
                        break
        targets = np.asarray(chosen, dtype=np.int64)
    return coalesced_edge_index(src, dst, n), n


def grid_graph(rows, cols):
    """rows x cols lattice, nodes numbered row-major."""
    idx = np.arange(rows * cols).reshape(rows, cols)
    src = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    dst = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    return coalesced_edge_index(src, dst, rows * cols), rows * cols


def erdos_renyi_graph(n, p, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    iu, ju = np.triu_indices(n, k=1)
    keep = rng.random(iu.shape[0]) < p
    return coalesced_edge_index(iu[keep], ju[keep], n), n
