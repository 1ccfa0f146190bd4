"""Dense-tensor entry points — call surface of the reference's curvature/bfc_cuda.py:51-65,144-159.

The reference launches two numba kernels on a dense N x N float32 adjacency.  Their float32 formula gives
different numbers from curvature/bfc_naive.py (C5 0.0 vs 1.0, K4 1.3333 vs 2.0, …) and
BASELINE.json names bfc_naive as the parity target, so these wrappers keep the reference's NAMES, ARGUMENTS and
OUTPUT SHAPES and compute Balanced Forman curvature with bfc_naive semantics on the MI355X CSR kernels:

  balanced_forman_curvature(A, C=None) -> C      C[i, j] = BFC(i, j) where A[i, j] != 0, else 0   (float32, as A)
  balanced_forman_post_delta(A, x, y, i_neighbors, j_neighbors, D=None) -> D
        D[I, J] = BFC(x, y) on G + (i_I, j_J); -1000 where i == j or the edge exists (bfc_cuda.py:77-79)
"""
import numpy as np
import torch

from dcr.graph import DcrGraph


def _graph_from_dense(A):
    N = A.shape[0]
    nz = torch.nonzero((A != 0) & ~torch.eye(N, dtype=torch.bool, device=A.device))
    sym = torch.cat([nz, nz.flip(1)], 0).unique(dim=0)            # undirected, coalesced, sorted
    ei = sym.t().contiguous().cpu().numpy()
    return DcrGraph(ei, N, device=A.device.index or 0 if A.is_cuda else 0)


def balanced_forman_curvature(A, C=None):
    N = A.shape[0]
    G = _graph_from_dense(A)
    eu, ev, cv = G.curvature_all('bfc')
    if C is None:
        C = torch.zeros(N, N, dtype=A.dtype, device=A.device)
    else:
        C.zero_()
    u = torch.from_numpy(eu.astype(np.int64)).to(A.device)
    v = torch.from_numpy(ev.astype(np.int64)).to(A.device)
    val = torch.from_numpy(cv).to(device=A.device, dtype=C.dtype)
    C[u, v] = val
    C[v, u] = val
    return C


def balanced_forman_post_delta(A, x, y, i_neighbors, j_neighbors, D=None):
    G = _graph_from_dense(A)
    before = G.curvature_edge(x, y, 'bfc') if min(G.degree(x), G.degree(y)) > 1 else 0.0
    imp, ci, cj = G.improvements(x, y, 'bfc', want_candidates=True)
    if D is None:
        D = torch.zeros(len(i_neighbors), len(j_neighbors), dtype=A.dtype, device=A.device)
    out = np.full((len(i_neighbors), len(j_neighbors)), -1000.0)
    # the device emits admissible pairs in the caller's nested-loop order when the neighbour lists are the
    # reference's (x_neighbors = list(G.neighbors(x)) + [x]); map by value so any ordering of the lists works
    best = {}
    for c in range(imp.shape[0]):
        best.setdefault((int(ci[c]), int(cj[c])), float(imp[c]))
    for I, i in enumerate(i_neighbors):
        for J, j in enumerate(j_neighbors):
            key = (min(i, j), max(i, j))
            if i != j and key in best:
                out[I, J] = before + best[key]
    D.copy_(torch.from_numpy(out).to(device=A.device, dtype=D.dtype))
    return D
