"""ctypes driver for oracle/libdcr_oracle.so — TEST INFRASTRUCTURE ONLY.

Wraps the C restatement (oracle/dcr_oracle.c) and composes the SDRF loop
around it the way rewiring/sdrf_no_cuda.py:22-66 does, with the softmax and
the legacy-numpy draw on the host exactly as in the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CURV = {'bfc': 0, '1d': 1, 'augmented': 2, 'haantjes': 3}

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libdcr_oracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.dcro_graph_create.argtypes = [ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, ctypes.POINTER(ctypes.c_void_p)]
        L.dcro_graph_destroy.argtypes = [ctypes.c_void_p]
        L.dcro_graph_destroy.restype = None
        L.dcro_num_edges.argtypes = [ctypes.c_void_p]
        L.dcro_num_edges.restype = ctypes.c_int64
        L.dcro_degree.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.dcro_degree.restype = ctypes.c_int32
        for f in (L.dcro_add_edge, L.dcro_remove_edge, L.dcro_has_edge):
            f.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
        L.dcro_export_edge_index.argtypes = [ctypes.c_void_p, _i64p]
        L.dcro_edges.argtypes = [ctypes.c_void_p, _i32p, _i32p]
        L.dcro_bfc_formula.argtypes = [ctypes.c_int64] * 6
        L.dcro_bfc_formula.restype = ctypes.c_double
        L.dcro_bfc_ingredients.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _i64p]
        L.dcro_curv_edge.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, _f64p]
        L.dcro_curv_edges.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _i32p, _i32p, _f64p]
        L.dcro_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _i32p, _i32p,
                                      _i64p]
        L.dcro_improvements.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int64,
                                        _i32p, _i32p, _f64p]
        L.dcro_improvements_mt.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int64,
                                           _i32p, _i32p, _f64p, ctypes.c_int]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class CGraph:
    def __init__(self, edge_index, num_nodes):
        ei = np.ascontiguousarray(np.asarray(edge_index), dtype=np.int64)
        src = np.ascontiguousarray(ei[0])
        dst = np.ascontiguousarray(ei[1])
        self.n = int(num_nodes)
        h = ctypes.c_void_p()
        rc = lib().dcro_graph_create(self.n, src.shape[0], _p(src, _i64p), _p(dst, _i64p), ctypes.byref(h))
        self.h = h
        if rc:
            raise ValueError(f'dcro_graph_create failed ({rc})')

    def __del__(self):
        if getattr(self, 'h', None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().dcro_graph_destroy(self.h)
            self.h = None

    def num_edges(self):
        return int(lib().dcro_num_edges(self.h))

    def degree(self, u):
        return int(lib().dcro_degree(self.h, u))

    def add_edge(self, u, v):
        return lib().dcro_add_edge(self.h, u, v)

    def remove_edge(self, u, v):
        return lib().dcro_remove_edge(self.h, u, v)

    def has_edge(self, u, v):
        return bool(lib().dcro_has_edge(self.h, u, v))

    def edges(self):
        ne = self.num_edges()
        eu = np.empty(ne, dtype=np.int32)
        ev = np.empty(ne, dtype=np.int32)
        lib().dcro_edges(self.h, _p(eu, _i32p), _p(ev, _i32p))
        return eu, ev

    def to_edge_index(self):
        out = np.empty((2, 2 * self.num_edges()), dtype=np.int64)
        lib().dcro_export_edge_index(self.h, _p(out, _i64p))
        return out

    def ingredients(self, u, v):
        out = np.empty(6, dtype=np.int64)
        lib().dcro_bfc_ingredients(self.h, u, v, _p(out, _i64p))
        return out

    def curv_edge(self, u, v, curv_type='bfc'):
        out = ctypes.c_double()
        lib().dcro_curv_edge(self.h, u, v, CURV[curv_type], ctypes.byref(out))
        return out.value

    def curv_edges(self, eu, ev, curv_type='bfc', nthreads=1):
        eu = np.ascontiguousarray(eu, dtype=np.int32)
        ev = np.ascontiguousarray(ev, dtype=np.int32)
        out = np.empty(eu.shape[0], dtype=np.float64)
        rc = lib().dcro_curv_edges(self.h, CURV[curv_type], nthreads, eu.shape[0], _p(eu, _i32p), _p(ev, _i32p),
                                   _p(out, _f64p))
        if rc != 0:
            raise MemoryError(f'dcro_curv_edges failed ({rc})')
        return out

    def curv_all(self, curv_type='bfc', nthreads=1):
        eu, ev = self.edges()
        return eu, ev, self.curv_edges(eu, ev, curv_type, nthreads)

    def candidates(self, x, y):
        n = ctypes.c_int64()
        cap = (self.degree(x) + 1) * (self.degree(y) + 1)
        ci = np.empty(cap, dtype=np.int32)
        cj = np.empty(cap, dtype=np.int32)
        lib().dcro_candidates(self.h, x, y, cap, _p(ci, _i32p), _p(cj, _i32p), ctypes.byref(n))
        return ci[:n.value].copy(), cj[:n.value].copy()

    def improvements(self, x, y, ci, cj, curv_type='bfc', nthreads=1):
        """sdrf_no_cuda.py:41-46 per candidate; ``nthreads`` > 1: the same literal add / recompute / remove, each worker on
        a private copy of the graph (the candidates are independent of one another)."""
        ci = np.ascontiguousarray(ci, dtype=np.int32)
        cj = np.ascontiguousarray(cj, dtype=np.int32)
        out = np.empty(ci.shape[0], dtype=np.float64)
        if nthreads > 1:
            rc = lib().dcro_improvements_mt(self.h, x, y, CURV[curv_type], ci.shape[0], _p(ci, _i32p), _p(cj, _i32p),
                                            _p(out, _f64p), nthreads)
            if rc != 0:
                raise MemoryError(f'dcro_improvements_mt failed ({rc})')
            return out
        lib().dcro_improvements(self.h, x, y, CURV[curv_type], ci.shape[0], _p(ci, _i32p), _p(cj, _i32p),
                                _p(out, _f64p))
        return out


def bfc_formula(d1, d2, T, s1, s2, gamma):
    return lib().dcro_bfc_formula(d1, d2, T, s1, s2, gamma)


def softmax(a, tau=1):
    """Restates utils/softmax.py:4-10: for tau = inf the indicator of the first arg-max, else exp(a * tau) over its
    (pairwise, numpy) sum; no max-subtraction, as in the reference."""
    if tau == float('inf'):
        indicator = np.zeros(len(a))
        indicator[np.argmax(a)] = 1
        return indicator
    weights = np.exp(a * tau)
    return weights / weights.sum()


def sdrf(edge_index, num_nodes, curv_type, loops, remove_edges, removal_bound, tau, trace=None, nthreads=1, compact=False,
         progress=None):
    """rewiring/sdrf_no_cuda.py:9-68 composed over the C restatement.  ``compact``: the trace keeps the number of candidates
    and a SHA-256 of the improvement vector's float64 bytes instead of the two lists (runs at the bench sizes: 10^5-10^6
    candidates per iteration); ``progress(iteration, record)`` is called after every iteration."""
    import hashlib
    G = CGraph(edge_index, num_nodes)
    for _ in range(loops):
        can_add = True
        eu, ev, curv = G.curv_all(curv_type, nthreads)
        m = int(np.argmin(curv))  # first minimum in G.edges order (np.argmin returns the first)
        x, y = int(eu[m]), int(ev[m])
        rec = {'argmin': [x, y]}
        ci, cj = G.candidates(x, y)
        if compact:
            rec['n_candidates'] = int(len(ci))
            rec['candidates_sha256'] = hashlib.sha256(np.stack([ci, cj], 1).astype(np.int32).tobytes()).hexdigest()
        else:
            rec['candidates'] = np.stack([ci, cj], 1).tolist()
        k = l = None
        if len(ci):
            imp = G.improvements(x, y, ci, cj, curv_type, nthreads)
            if compact:
                rec['improvements_sha256'] = hashlib.sha256(np.ascontiguousarray(imp, dtype=np.float64).tobytes()).hexdigest()
                rec['argmin_value_hex'] = float(curv[m]).hex()
            else:
                rec['improvements'] = imp.tolist()
            idx = np.random.choice(range(len(ci)), p=softmax(imp, tau=tau))
            rec['choice'] = int(idx)
            k, l = int(ci[idx]), int(cj[idx])
            G.add_edge(k, l)
            rec['added'] = [k, l]
        else:
            rec.update(choice=None, added=None) if compact else rec.update(improvements=[], choice=None, added=None)
            can_add = False
            if not remove_edges:
                rec['removed'] = None
                if trace is not None:
                    trace.append(rec)
                break
        rec['removed'] = None
        stop = False
        if remove_edges:
            # stale curvatures over the *new* G.edges minus (k, l): the stale
            # edge list is the same sequence with (k, l) appended to row k, so
            # the first maximum over the stale list is the answer.
            m = int(np.argmax(curv))
            x, y = int(eu[m]), int(ev[m])
            if curv[m] > removal_bound:
                G.remove_edge(x, y)
                rec['removed'] = [x, y]
            elif can_add is False:
                stop = True
        if trace is not None:
            trace.append(rec)
        if progress is not None:
            progress(len(trace) if trace is not None else -1, rec)
        if stop:
            break
    return G.to_edge_index()
