"""Device-resident graph handle: the host-side mirror of the slice of
``networkx.Graph`` the reference's curvature / SDRF code uses
(rewiring/sdrf_no_cuda.py:20-68, curvature/bfc_naive.py:7-52)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib

CURV = {'bfc': 0, '1d': 1, 'augmented': 2, 'haantjes': 3}


def curv_code(curv_type):
    try:
        return CURV[curv_type]
    except KeyError:
        # classical_curvatures.py:28 raises a bare Exception with this text
        raise Exception(f'Method {curv_type} not available.')


def _as_numpy_edge_index(edge_index):
    if hasattr(edge_index, 'detach'):
        edge_index = edge_index.detach().cpu().numpy()
    ei = np.ascontiguousarray(np.asarray(edge_index), dtype=np.int64)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError('edge_index must have shape [2, M]')
    return ei



class _HostBlock:
    """Carrier of an ``__array_interface__`` for a library-owned host buffer."""
    __slots__ = ('__array_interface__',)


def _host_view(ptr, n, typestr):
    """numpy view of n elements at a ctypes pointer.  (``np.ctypeslib.as_array(ptr, shape)`` builds a new ctypes array
    type for every distinct n, ~0.3 ms per call: more than the improvement kernels take.)"""
    blk = _HostBlock()
    blk.__array_interface__ = {'data': (ctypes.cast(ptr, ctypes.c_void_p).value, False), 'shape': (int(n),),
                               'typestr': typestr, 'version': 3}
    return np.asarray(blk)

class DcrGraph:
    """Undirected simple graph on nodes 0..n-1 living in HBM.

    Construction follows ``to_networkx(data, to_undirected=True)``
    (sdrf_no_cuda.py:20): pairs with dst <= src are kept, in order; adjacency
    rows keep insertion order, which fixes ``G.edges`` order and every
    first-extremum tie-break of the SDRF loop.
    """

    def __init__(self, edge_index, num_nodes, device=0):
        ei = _as_numpy_edge_index(edge_index)
        src = np.ascontiguousarray(ei[0])
        dst = np.ascontiguousarray(ei[1])
        self._h = ctypes.c_void_p()
        self.num_nodes = int(num_nodes)
        rc = lib().dcr_graph_create(int(device), self.num_nodes, src.shape[0],
                                    src.ctypes.data_as(_lib._i64p), dst.ctypes.data_as(_lib._i64p),
                                    ctypes.byref(self._h))
        if rc != 0:
            h, self._h = self._h, ctypes.c_void_p()
            msg = lib().dcr_last_error().decode()
            if h:
                lib().dcr_graph_destroy(h)
            if rc == -1:
                raise ValueError(msg)
            raise _lib.DcrError(f'libdcr_hip error {rc}: {msg}')

    @classmethod
    def from_data(cls, data, device=0):
        return cls(data.edge_index, data.num_nodes, device=device)

    def close(self):
        if getattr(self, '_h', None):
            lib().dcr_graph_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- container -----------------------------------------------------------------
    def number_of_nodes(self):
        return self.num_nodes

    def number_of_edges(self):
        out = ctypes.c_int64()
        check(lib().dcr_graph_num_edges(self._h, ctypes.byref(out)))
        return out.value

    def add_edge(self, u, v):
        check(lib().dcr_graph_add_edge(self._h, int(u), int(v)))

    def remove_edge(self, u, v):
        check(lib().dcr_graph_remove_edge(self._h, int(u), int(v)))

    def has_edge(self, u, v):
        out = ctypes.c_int()
        check(lib().dcr_graph_has_edge(self._h, int(u), int(v), ctypes.byref(out)))
        return bool(out.value)

    def degree(self, u):
        out = ctypes.c_int32()
        check(lib().dcr_graph_degree(self._h, int(u), ctypes.byref(out)))
        return out.value

    def neighbors(self, u):
        d = self.degree(u)
        buf = np.empty(max(d, 1), dtype=np.int32)
        n = ctypes.c_int64()
        check(lib().dcr_graph_neighbors(self._h, int(u), buf.shape[0], buf.ctypes.data_as(_lib._i32p),
                                        ctypes.byref(n)))
        return buf[:n.value].tolist()

    def edges(self):
        """(u, v) arrays in ``G.edges`` order."""
        ne = self.number_of_edges()
        eu = np.empty(ne, dtype=np.int32)
        ev = np.empty(ne, dtype=np.int32)
        check(lib().dcr_graph_edges(self._h, eu.ctypes.data_as(_lib._i32p), ev.ctypes.data_as(_lib._i32p)))
        return eu, ev

    def to_edge_index(self):
        """``from_networkx(G).edge_index`` as int64 numpy [2, 2E] (sdrf_no_cuda.py:68)."""
        out = np.empty((2, 2 * self.number_of_edges()), dtype=np.int64)
        check(lib().dcr_graph_export_edge_index(self._h, out.ctypes.data_as(_lib._i64p)))
        return out

    # ---- curvature -------------------------------------------------------------------
    def curvature_pass(self, curv_type='bfc', incremental=False):
        fn = lib().dcr_curvature_pass_incremental if incremental else lib().dcr_curvature_pass
        check(fn(self._h, curv_code(curv_type)))

    def curvature_pass_argmin(self, curv_type='bfc', incremental=False):
        """One pass, then the first minimum in ``G.edges`` order: (u, v, value), one host synchronisation."""
        u, v, val = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        check(lib().dcr_curvature_pass_argmin(self._h, curv_code(curv_type), int(bool(incremental)), ctypes.byref(u),
                                              ctypes.byref(v), ctypes.byref(val)))
        return u.value, v.value, val.value

    def curvature_read(self):
        ne = self.number_of_edges()
        cv = np.empty(ne, dtype=np.float64)
        eu = np.empty(ne, dtype=np.int32)
        ev = np.empty(ne, dtype=np.int32)
        check(lib().dcr_curvature_read(self._h, cv.ctypes.data_as(_lib._f64p), eu.ctypes.data_as(_lib._i32p),
                                       ev.ctypes.data_as(_lib._i32p)))
        return eu, ev, cv

    def curvature_all(self, curv_type='bfc'):
        self.curvature_pass(curv_type)
        return self.curvature_read()

    def curvature_edge(self, u, v, curv_type='bfc'):
        out = ctypes.c_double()
        check(lib().dcr_curvature_edge(self._h, int(u), int(v), curv_code(curv_type), ctypes.byref(out)))
        return out.value

    def bfc_ingredients(self, u, v):
        out = np.empty(6, dtype=np.int64)
        check(lib().dcr_bfc_ingredients(self._h, int(u), int(v), out.ctypes.data_as(_lib._i64p)))
        return out

    def argext(self, want_max, exclude=None):
        u, v, val = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        eu, ev = (-1, -1) if exclude is None else (int(exclude[0]), int(exclude[1]))
        check(lib().dcr_argext(self._h, int(bool(want_max)), eu, ev, ctypes.byref(u), ctypes.byref(v),
                               ctypes.byref(val)))
        return u.value, v.value, val.value

    def improvements(self, x, y, curv_type='bfc', want_candidates=False):
        """Returns (improvements view, ci, cj); the arrays are views of library-owned pinned
        buffers, valid until the next call on this graph."""
        n = ctypes.c_int64()
        pi, pci, pcj = _lib._f64p(), _lib._i32p(), _lib._i32p()
        check(lib().dcr_improvements(self._h, int(x), int(y), curv_code(curv_type), int(bool(want_candidates)),
                                     ctypes.byref(n), ctypes.byref(pi), ctypes.byref(pci), ctypes.byref(pcj)))
        if n.value == 0:
            e = np.empty(0, dtype=np.float64)
            return e, np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)
        imp = _host_view(pi, n.value, '<f8')
        ci = cj = None
        if want_candidates:
            ci = _host_view(pci, n.value, '<i4')
            cj = _host_view(pcj, n.value, '<i4')
        return imp, ci, cj

    def improvements_count(self, x, y, curv_type='bfc'):
        """Run the improvement pipeline but leave the values on the device (tau = inf path)."""
        n = ctypes.c_int64()
        check(lib().dcr_improvements(self._h, int(x), int(y), curv_code(curv_type), 0, ctypes.byref(n), None, None,
                                     None))
        return n.value

    def improvements_argmax(self):
        out = ctypes.c_int64()
        check(lib().dcr_improvements_argmax(self._h, ctypes.byref(out)))
        return out.value

    def candidate_at(self, index):
        i, j = ctypes.c_int32(), ctypes.c_int32()
        check(lib().dcr_candidate_at(self._h, int(index), ctypes.byref(i), ctypes.byref(j)))
        return i.value, j.value

    def sdrf_tail(self, add, do_remove, removal_bound):
        k, l = (-1, -1) if add is None else (int(add[0]), int(add[1]))
        removed = (ctypes.c_int32 * 2)(-1, -1)
        mx = ctypes.c_double()
        check(lib().dcr_sdrf_tail(self._h, k, l, int(bool(do_remove)), float(removal_bound), removed,
                                  ctypes.byref(mx)))
        rem = None if removed[0] < 0 else (removed[0], removed[1])
        return rem, mx.value

    def sdrf_tail_at(self, cand_index, do_remove, removal_bound):
        """``sdrf_tail`` with the edge to add given by its index in the last candidate list; returns
        (added pair, removed pair or None, stale maximum)."""
        added = (ctypes.c_int32 * 2)(-1, -1)
        removed = (ctypes.c_int32 * 2)(-1, -1)
        mx = ctypes.c_double()
        check(lib().dcr_sdrf_tail_at(self._h, int(cand_index), int(bool(do_remove)), float(removal_bound), added, removed,
                                     ctypes.byref(mx)))
        rem = None if removed[0] < 0 else (removed[0], removed[1])
        return (added[0], added[1]), rem, mx.value

    def sdrf_tail_at_pass_argmin(self, cand_index, do_remove, removal_bound, curv_type='bfc', incremental=False):
        """``sdrf_tail_at`` and the next iteration's ``curvature_pass_argmin`` with one host synchronisation; returns
        (added pair, removed pair or None, (u, v, value) of the next first minimum)."""
        added = (ctypes.c_int32 * 2)(-1, -1)
        removed = (ctypes.c_int32 * 2)(-1, -1)
        u, v, val = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        check(lib().dcr_sdrf_tail_at_pass_argmin(self._h, int(cand_index), int(bool(do_remove)), float(removal_bound),
                                                 curv_code(curv_type), int(bool(incremental)), added, removed,
                                                 ctypes.byref(u), ctypes.byref(v), ctypes.byref(val)))
        rem = None if removed[0] < 0 else (removed[0], removed[1])
        return (added[0], added[1]), rem, (u.value, v.value, val.value)

    def sdrf_iteration_device_draw(self, x, y, curv_type, tau, uniform, do_remove, removal_bound, incremental=False):
        """One loop iteration for the edge (x, y) with the draw on the device and one host synchronisation
        (``dcr_sdrf_iteration_device_draw``); returns (status, candidates, added pair, removed pair or None, (u, v, value) of the
        next first minimum).  status != 0: nothing was edited (1: the draw was left undecided, 2: no candidates)."""
        status, n_cand = ctypes.c_int(), ctypes.c_int64()
        added = (ctypes.c_int32 * 2)(-1, -1)
        removed = (ctypes.c_int32 * 2)(-1, -1)
        u, v, val = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        check(lib().dcr_sdrf_iteration_device_draw(self._h, int(x), int(y), curv_code(curv_type), float(tau), float(uniform),
                                                   int(bool(do_remove)), float(removal_bound), int(bool(incremental)),
                                                   ctypes.byref(status), ctypes.byref(n_cand), added, removed,
                                                   ctypes.byref(u), ctypes.byref(v), ctypes.byref(val)))
        rem = None if removed[0] < 0 else (removed[0], removed[1])
        return status.value, n_cand.value, (added[0], added[1]), rem, (u.value, v.value, val.value)

    # ---- measurement hooks ------------------------------------------------------------
    def profile_reset(self):
        check(lib().dcr_profile_reset(self._h))

    def profile_read(self):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        check(lib().dcr_profile_read(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def pass_engine(self):
        """Which kernels ran the last curvature pass (all produce the same bits)."""
        out = ctypes.c_int()
        check(lib().dcr_pass_engine(self._h, ctypes.byref(out)))
        return {0: 'two-hop', 1: 'edge-centric', 2: 'node-centric'}.get(out.value, 'none')

    def bfc_algorithmic_bytes(self, one_sided=False):
        """SURVEY §8(d) bytes of one BFC pass; ``one_sided``: only the cheaper difference set's rows per edge."""
        out = ctypes.c_double()
        fn = lib().dcr_bfc_algorithmic_bytes_one_sided if one_sided else lib().dcr_bfc_algorithmic_bytes
        check(fn(self._h, ctypes.byref(out)))
        return out.value
