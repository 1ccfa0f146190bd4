"""BFC SDRF entry point — call surface of the reference's rewiring/sdrf_cuda_bfc.py:14-93.

``rewire('bfc')`` dispatches here (rewiring/rewire.py:8-10).  The reference's
numba kernels compute a float32 dense formula that differs numerically from
curvature/bfc_naive.py; BASELINE.json names
``bfc_naive.py + sdrf_no_cuda.py`` as the parity target, so this entry point
runs the same device pipeline as ``sdrf_no_cuda(data, 'bfc', ...)``.
"""
from rewiring.sdrf_no_cuda import sdrf_no_cuda


def sdrf_cuda_bfc(data, loops, remove_edges, removal_bound, tau, is_undirected, trace=None, device=0):
    if not is_undirected:
        raise NotImplementedError('directed SDRF (sdrf_cuda_bfc.py:47-49) is not built yet')
    return sdrf_no_cuda(data, 'bfc', loops, remove_edges, removal_bound, tau, trace=trace, device=device)
