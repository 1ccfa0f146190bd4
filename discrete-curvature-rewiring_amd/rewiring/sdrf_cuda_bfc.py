"""BFC SDRF entry point — call surface of the reference's rewiring/sdrf_cuda_bfc.py:14-93; ``rewire('bfc')`` dispatches
here (rewiring/rewire.py:8-10).

``numerics='bfc_naive'`` (default, or whatever ``curvature.bfc_cuda.set_numerics`` / ``DCR_BFC_NUMERICS`` selected): the
loop of rewiring/sdrf_no_cuda.py:22-66 with curvature/bfc_naive.py as the curvature, on the CSR kernels — the parity
target BASELINE.json names.  Undirected graphs only (sdrf_no_cuda.py has no directed mode).

``numerics='bfc_cuda'``: the reference's own loop, line for line in behaviour (sdrf_cuda_bfc.py:25-93): dense float32
adjacency on the device, curvature and post-delta by the kernels of csrc/dcr_bfc_dense.hip (the reference's float32
formula), arg-min / arg-max over the DENSE matrix with the zeros of non-edges taking part (:40, :80), candidates as
unsorted (i, j) pairs (:54), no exclusion of the edge just added at the removal step, float32 improvements (:61), and
``is_undirected=False`` through successors / predecessors with one-sided updates of ``A`` (:47-49, :73, :88).  What is
not kept is the reference's host synchronisation per candidate (:59-62): the improvements come back in one copy.
"""
import numpy as np
import torch

from curvature import bfc_cuda
from dcr.data import Data
from dcr.ordered_graph import digraph_from_edge_index
from rewiring.sdrf_no_cuda import sdrf_no_cuda
from utils.softmax import softmax


def _to_undirected(edge_index):
    """torch_geometric.utils.to_undirected (PyG 2.0.3): both directions, coalesced, sorted by (row, col)."""
    row, col = edge_index[0], edge_index[1]
    row, col = torch.cat([row, col]), torch.cat([col, row])
    n = int(torch.max(row.max(), col.max())) + 1 if row.numel() else 0
    key = torch.unique(row * n + col)
    return torch.stack([key // n, key % n])


def _dense_loop(data, loops, remove_edges, removal_bound, tau, is_undirected, trace, device):
    dev = torch.device('cuda', device) if isinstance(device, int) else torch.device(device)
    edge_index = torch.as_tensor(data.edge_index).cpu()
    dense_ei = _to_undirected(edge_index) if is_undirected else edge_index
    dense_ei = dense_ei[:, dense_ei[0] != dense_ei[1]]                       # remove_self_loops (:29)
    N = int(dense_ei.max()) + 1 if dense_ei.numel() else 0                    # to_dense_adj sizes by the largest id
    A = torch.zeros(N, N, dtype=torch.float32, device=dev)
    if dense_ei.numel():
        A.index_put_((dense_ei[0].to(dev), dense_ei[1].to(dev)), torch.ones(dense_ei.shape[1], device=dev), accumulate=True)
    G = digraph_from_edge_index(edge_index.numpy(), data.num_nodes)           # to_networkx(data) (:31)
    if is_undirected:
        G = G.to_undirected()
    C = torch.zeros(N, N, dtype=torch.float32, device=dev)
    for _ in range(loops):
        can_add = True
        bfc_cuda.balanced_forman_curvature(A, C=C, numerics='bfc_cuda')
        ext = torch.stack([C.argmin(), C.argmax()]).cpu()                     # one host round trip for :40 and :80
        ix_min, ix_max = int(ext[0]), int(ext[1])
        x, y = ix_min // N, ix_min % N
        if is_undirected:
            x_nb, y_nb = G.neighbors(x) + [x], G.neighbors(y) + [y]
        else:
            x_nb, y_nb = G.successors(x) + [x], G.predecessors(y) + [y]
        candidates = [(i, j) for i in x_nb for j in y_nb if i != j and not G.has_edge(i, j)]
        rec = {'argmin': [x, y], 'x_neighbors': x_nb, 'y_neighbors': y_nb, 'n_candidates': len(candidates),
               'improvements': None, 'choice': None, 'events': []}
        if candidates:
            D = bfc_cuda.balanced_forman_post_delta(A, x, y, x_nb, y_nb, numerics='bfc_cuda')
            first_i, first_j = {}, {}
            for p, i in enumerate(x_nb):                                      # list.index: the first occurrence
                first_i.setdefault(i, p)
            for p, j in enumerate(y_nb):
                first_j.setdefault(j, p)
            rows = torch.tensor([first_i[i] for i, _ in candidates], device=dev)
            cols = torch.tensor([first_j[j] for _, j in candidates], device=dev)
            improvements = (D - C[x, y])[rows, cols].cpu().tolist()           # float32 subtraction (:61), one copy
            rec['improvements'] = improvements
            idx = np.random.choice(range(len(candidates)), p=softmax(np.array(improvements), tau=tau))
            rec['choice'] = int(idx)
            k, l = candidates[idx]
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
            x, y = ix_max // N, ix_max % N
            if bool(C[x, y] > removal_bound):                                 # stale C, nothing excluded (:80-83)
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
    return Data(edge_index=torch.from_numpy(G.to_edge_index()), num_nodes=G.n)


def sdrf_cuda_bfc(data, loops, remove_edges, removal_bound, tau, is_undirected, trace=None, device=0, numerics=None):
    name = bfc_cuda.get_numerics() if numerics is None else numerics
    if name == 'bfc_cuda':
        return _dense_loop(data, loops, remove_edges, removal_bound, tau, is_undirected, trace, device)
    if name != 'bfc_naive':
        raise ValueError(f'unknown numerics {name!r}')
    if not is_undirected:
        raise ValueError("directed graphs are defined for numerics='bfc_cuda' only: rewiring/sdrf_no_cuda.py, whose loop "
                         "'bfc_naive' follows, has no directed mode (pass numerics='bfc_cuda' or set DCR_BFC_NUMERICS)")
    return sdrf_no_cuda(data, 'bfc', loops, remove_edges, removal_bound, tau, trace=trace, device=device)
