"""Dense-tensor entry points — call surface of the reference's curvature/bfc_cuda.py:51-65,144-159.

  balanced_forman_curvature(A, C=None) -> C
  balanced_forman_post_delta(A, x, y, i_neighbors, j_neighbors, D=None) -> D      (-1000 where i == j or the edge exists)

Two numerics, chosen per call (``numerics=``) or for the process (``set_numerics`` / environment ``DCR_BFC_NUMERICS``):

``'bfc_naive'`` (default): Balanced Forman curvature as curvature/bfc_naive.py:7-40 defines it, float64 inside, computed
    by the CSR kernels (csrc/dcr_bfc_nc.hip, dcr_sdrf.hip).  BASELINE.json names bfc_naive + sdrf_no_cuda as the parity
    target of this build, so this is what ``rewire('bfc')`` runs unless told otherwise.

``'bfc_cuda'``: the reference's own numba kernels restated (curvature/bfc_cuda.py:11-48, :68-141) — a float32 dense formula
    with another 4-cycle term and no degree-1 rule, which gives different numbers (C5 1.0 vs 0.0,
    K4 2.0 vs 1.3333, ...) and is what ``rewire('bfc')`` runs in the reference.  Results obtained with the reference
    reproduce only in this mode.  ``A`` must be a float32 tensor on the MI355X; A2 = A @ A goes through the ROCm GEMM
    library (exact on 0/1 entries) and the per-pair loop and the float64 -> float32 closing expression run in
    csrc/dcr_bfc_dense.hip.  Directed adjacency matrices are supported (d_in = column sums, d_out = row sums).
"""
import ctypes
import os

import numpy as np
import torch

from dcr.graph import DcrGraph

_NUMERICS = os.environ.get('DCR_BFC_NUMERICS', 'bfc_naive')


def set_numerics(name):
    global _NUMERICS
    if name not in ('bfc_naive', 'bfc_cuda'):
        raise ValueError(name)
    _NUMERICS = name


def get_numerics():
    return _NUMERICS


def _pick(numerics):
    name = _NUMERICS if numerics is None else numerics
    if name not in ('bfc_naive', 'bfc_cuda'):
        raise ValueError(f'unknown numerics {name!r}')
    return name


def _require_device_f32(A):
    if not (isinstance(A, torch.Tensor) and A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.shape[0] == A.shape[1]):
        raise RuntimeError("numerics='bfc_cuda' needs a square float32 adjacency tensor on the MI355X (the reference's "
                           'kernels take a CUDA tensor, curvature/bfc_cuda.py:57; there is no CPU fallback)')
    return A.contiguous()


# ---- 'bfc_cuda': the reference's dense float32 formula ------------------------------------------------------------------
def _dense_curvature(A, C):
    from dcr import _lib
    A = _require_device_f32(A)
    N = A.shape[0]
    A2 = torch.matmul(A, A)                      # bfc_cuda.py:53
    d_in, d_out = A.sum(dim=0), A.sum(dim=1)     # :54-55
    if C is None:
        C = torch.zeros(N, N, dtype=torch.float32, device=A.device)
    else:
        C.zero_()                                # the kernel writes 0 wherever A is 0 (:15-17)
    pairs = torch.nonzero(A).contiguous()        # int64 [nnz, 2]
    stream = torch.cuda.current_stream(A.device).cuda_stream
    _lib.check(_lib.lib().dcr_bfc_dense_f32_dev(A.data_ptr(), A2.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), N,
                                                pairs.data_ptr(), pairs.shape[0], C.data_ptr(), ctypes.c_void_p(stream)))
    return C


def _dense_post_delta(A, x, y, i_neighbors, j_neighbors, D):
    from dcr import _lib
    A = _require_device_f32(A)
    N = A.shape[0]
    A2 = torch.matmul(A, A)                      # bfc_cuda.py:146
    sums = torch.stack([A[:, x].sum(), A[y].sum()]).cpu()   # :147-148, one host round trip for both
    dim_i, dim_j = len(i_neighbors), len(j_neighbors)
    if D is None:
        D = torch.zeros(dim_i, dim_j, dtype=torch.float32, device=A.device)
    nb = torch.tensor(list(i_neighbors) + list(j_neighbors), dtype=torch.int32, device=A.device)
    stream = torch.cuda.current_stream(A.device).cuda_stream
    _lib.check(_lib.lib().dcr_bfc_dense_post_delta_f32_dev(
        A.data_ptr(), A2.data_ptr(), float(sums[0]), float(sums[1]), N, D.data_ptr(), int(x), int(y), nb.data_ptr(),
        nb.data_ptr() + 4 * dim_i, dim_i, dim_j, ctypes.c_void_p(stream)))
    return D


# ---- 'bfc_naive': Balanced Forman curvature proper, on the CSR kernels --------------------------------------------------
def _graph_from_dense(A):
    N = A.shape[0]
    nz = torch.nonzero((A != 0) & ~torch.eye(N, dtype=torch.bool, device=A.device))
    sym = torch.cat([nz, nz.flip(1)], 0).unique(dim=0)            # undirected, coalesced, sorted
    ei = sym.t().contiguous().cpu().numpy()
    return DcrGraph(ei, N, device=A.device.index or 0 if A.is_cuda else 0)


def _naive_curvature(A, C):
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


def _naive_post_delta(A, x, y, i_neighbors, j_neighbors, D):
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


def balanced_forman_curvature(A, C=None, numerics=None):
    if _pick(numerics) == 'bfc_cuda':
        return _dense_curvature(A, C)
    return _naive_curvature(A, C)


def balanced_forman_post_delta(A, x, y, i_neighbors, j_neighbors, D=None, numerics=None):
    if _pick(numerics) == 'bfc_cuda':
        return _dense_post_delta(A, x, y, i_neighbors, j_neighbors, D)
    return _naive_post_delta(A, x, y, i_neighbors, j_neighbors, D)
