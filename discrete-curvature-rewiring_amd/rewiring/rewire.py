"""Rewiring dispatcher — call surface and behaviour of the reference's rewiring/rewire.py:7-14: ``'bfc'`` goes to the
``sdrf_cuda_bfc`` entry point, any other curvature name to ``sdrf_no_cuda``, ``None`` leaves the graph alone; removal is
always on and the graph is treated as undirected.  Returns the (rewired) ``edge_index``."""
from rewiring.sdrf_cuda_bfc import sdrf_cuda_bfc
from rewiring.sdrf_no_cuda import sdrf_no_cuda


def rewire(dt, curv_type, max_iterations, removal_bound, tau):
    if curv_type is None:
        return dt.edge_index
    loop_args = dict(loops=max_iterations, remove_edges=True, removal_bound=removal_bound, tau=tau)
    if curv_type == 'bfc':
        rewired = sdrf_cuda_bfc(dt, is_undirected=True, **loop_args)
    else:
        rewired = sdrf_no_cuda(dt, curv_type, **loop_args)
    return rewired.edge_index
