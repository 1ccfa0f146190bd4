"""SDRF rewiring loop — call surface of the reference's rewiring/sdrf_no_cuda.py:9-68.

Same signature, same loop, same results (edge list bit-identical for a given
``np.random.seed``); every step that touches the graph runs on the MI355X:

  reference line                       here
  :20  to_networkx                     DcrGraph.from_data  (HBM-resident rows, insertion order)
  :24  compute_curvature_graph  \       G.curvature_pass_argmin   (csrc/dcr_bfc_nc.hip; wavefront reduction for
  :27  min(G.edges, key=...)    /                                  the first minimum; one host sync for both)
  :29-46 candidates + improvements     G.improvements      (csrc/dcr_sdrf.hip)
  :49-50 softmax + np.random.choice    draw_index: numpy's exp and sum, the rest fused on the host (bit-exact draw)
  :51,57-66 add / stale arg-max / remove   G.sdrf_tail     (one fused device step)
  :68  from_networkx                   G.to_edge_index

``curv_type='bfc'`` is accepted in addition to the reference's classical kinds
(the reference raises there; BASELINE.json names this composition as the parity
target).  ``trace`` (optional list) receives one dict per iteration.
"""
import os

import numpy as np

from dcr.graph import DcrGraph, curv_code
from utils.softmax import softmax


def _make_data(data, edge_index):
    import torch
    ei = torch.from_numpy(edge_index)
    try:
        from torch_geometric.data import Data as PygData  # real PyG when installed
        return PygData(x=getattr(data, 'x', None), edge_index=ei)
    except Exception:
        from dcr.data import Data
        return Data(x=getattr(data, 'x', None), edge_index=ei, num_nodes=data.num_nodes)


_ATOL = float(np.sqrt(np.finfo(np.float64).eps))


def choice_index(p):
    """``np.random.choice(len(p), p=p)`` (sdrf_no_cuda.py:49-50) for a vector produced by ``utils.softmax``: the same
    index and the same consumption of the legacy global stream.  ``RandomState.choice`` computes
    ``cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(random_sample(), side='right')`` after validating
    ``p``; for ``exp(a*tau)/sum`` those checks can only trip on NaN (overflow) or an all-zero vector, which are
    raised here with numpy's messages.  Pinned against numpy in tests/test_host_cpu.py."""
    cdf = p.cumsum()
    total = cdf[-1]
    if np.isnan(total):
        raise ValueError('probabilities contain NaN')
    if abs(total - 1.0) > _ATOL:
        raise ValueError('probabilities do not sum to 1')
    # searchsorted(cdf / total, u, side='right') without dividing the whole vector: x -> x / total is monotone in
    # floating point, so the first index whose quotient exceeds u is found by bisection on the fly
    u = np.random.random_sample()
    lo, hi = 0, cdf.shape[0]
    while lo < hi:
        mid = (lo + hi) >> 1
        if cdf[mid] / total > u:
            hi = mid
        else:
            lo = mid + 1
    return lo


_cdf_scratch = np.empty(0, dtype=np.float64)
_exp_scratch = np.empty(0, dtype=np.float64)


def draw_index(improvements, tau):
    """Index drawn by ``np.random.choice(n, p=softmax(improvements, tau))`` (sdrf_no_cuda.py:49-50) for finite tau, with
    the same consumption of the legacy global stream.  ``exp`` and the pairwise ``sum`` are numpy's own (their rounding
    is part of the bit-exact contract); the division and the sequential cumsum run fused in the library's host helper
    ``dcr_host_cdf_from_exp`` with numpy's operations in numpy's order; validation, the one uniform and the search follow
    ``RandomState.choice``.  Pinned against numpy in tests/test_host_cpu.py."""
    global _cdf_scratch, _exp_scratch
    import ctypes
    from dcr import _lib
    a = np.asarray(improvements, dtype=np.float64)
    n = a.shape[0]
    if _exp_scratch.shape[0] < n:
        _exp_scratch = np.empty(n + n // 4 + 64, dtype=np.float64)
        _cdf_scratch = np.empty(n + n // 4 + 64, dtype=np.float64)
    # the same numpy operations as utils/softmax.py:9 (multiply, exp, pairwise sum), written into scratch that is kept
    # between iterations: two fresh 1.4 MB temporaries per draw cost more in page faults than the arithmetic
    exp_a = _exp_scratch[:n]
    np.multiply(a, tau, out=exp_a)
    np.exp(exp_a, out=exp_a)
    s = exp_a.sum()
    if not np.isfinite(s) or s == 0.0:
        return choice_index(exp_a / s)  # overflow / all-zero: numpy's own NaN pattern and error (utils/softmax.py:9-10)
    cdf = _cdf_scratch
    total = ctypes.c_double()
    _lib.check(_lib.lib().dcr_host_cdf_from_exp(exp_a.ctypes.data_as(_lib._f64p), n, float(s),
                                                cdf.ctypes.data_as(_lib._f64p), ctypes.byref(total)))
    t = total.value
    if np.isnan(t):
        raise ValueError('probabilities contain NaN')
    if abs(t - 1.0) > _ATOL:
        raise ValueError('probabilities do not sum to 1')
    u = np.random.random_sample()
    lo, hi = 0, n
    while lo < hi:  # searchsorted(cdf / t, u, side='right'); x -> x / t is monotone in floating point
        mid = (lo + hi) >> 1
        if cdf[mid] / t > u:
            hi = mid
        else:
            lo = mid + 1
    return lo


class _StreamMark:
    """Put ONE uniform back into numpy's global stream (the device-side draw takes it before it knows whether numpy would
    have).  ``np.random.get_state()`` / ``set_state()`` cost 34 us each — more than the whole draw on the device; the global
    ``RandomState`` sits on an ``MT19937`` bit generator whose state struct (624 key words + position: 2,500 bytes) numpy
    exposes by address for exactly this kind of interop (``BitGenerator.ctypes.state_address``), so the mark is a 2.5 KB
    ``memmove`` (0.3 us).  Checked once against ``get_state`` / ``set_state`` themselves; anything unexpected (another bit
    generator installed, the check failing) falls back to those two calls."""

    _SIZE = 624 * 4 + 4
    _checked = None  # None: not yet; True / False: the fast way is / is not usable

    def __init__(self):
        import ctypes
        self._ctypes = ctypes
        self._buf = ctypes.create_string_buffer(self._SIZE)
        self._slow = None
        self._addr = 0

    @staticmethod
    def _address():
        try:
            from numpy.random import MT19937
            bg = np.random.mtrand._rand._bit_generator
            return int(bg.ctypes.state_address) if type(bg) is MT19937 else 0
        except Exception:  # noqa: BLE001  (private attribute layout of another numpy)
            return 0

    @classmethod
    def _self_check(cls):
        addr = cls._address()
        if not addr:
            return False
        import ctypes
        before = np.random.get_state()
        try:
            buf = ctypes.create_string_buffer(cls._SIZE)
            ctypes.memmove(buf, addr, cls._SIZE)
            first = np.random.random_sample(3)
            moved = np.random.get_state()
            ctypes.memmove(addr, buf, cls._SIZE)
            back = np.random.get_state()
            again = np.random.random_sample(3)
            same = lambda a, b: np.array_equal(a[1], b[1]) and a[2] == b[2]   # (key words, position)
            ok = same(back, before) and not same(moved, before)
            return bool(ok and np.array_equal(first, again))
        finally:
            np.random.set_state(before)

    def mark(self):
        if _StreamMark._checked is None:
            _StreamMark._checked = self._self_check()
        self._addr = self._address() if _StreamMark._checked else 0
        if self._addr:
            self._ctypes.memmove(self._buf, self._addr, self._SIZE)
            self._slow = None
        else:
            self._slow = np.random.get_state()

    def rewind(self):
        if self._slow is not None:
            np.random.set_state(self._slow)
        elif self._addr and self._addr == self._address():
            self._ctypes.memmove(self._addr, self._buf, self._SIZE)
        else:
            raise RuntimeError('numpy global bit generator was replaced between mark() and rewind()')


class SdrfRun:
    """One SDRF rewiring run, steppable: ``step()`` is one iteration of the loop body
    sdrf_no_cuda.py:22-66 and returns False when the reference loop would ``break``."""

    def __init__(self, data, curv_type, remove_edges, removal_bound, tau, trace=None, device=0, incremental=None):
        curv_code(curv_type)
        # incremental=False recomputes every edge each iteration like the reference (sdrf_no_cuda.py:24);
        # True only recomputes edges near the previous iteration's edits: same values, less work
        if incremental is None:  # the reference's signature has no such switch: DCR_INCREMENTAL=1 turns it on for drop-in callers
            incremental = os.environ.get('DCR_INCREMENTAL', '0') == '1'
        self.incremental = bool(incremental)
        self.data = data
        self.curv_type = curv_type
        self.remove_edges = remove_edges
        self.removal_bound = removal_bound
        self.tau = tau
        self.trace = trace
        self.G = DcrGraph.from_data(data, device=device)
        # DCR_DEVICE_DRAW=0: always draw on the host (numpy's exp / sum / cumsum on the downloaded improvements)
        self.device_draw = os.environ.get('DCR_DEVICE_DRAW', '1') != '0'
        self.device_draws = self.host_draws = self.no_candidate_iterations = 0
        self._mark = _StreamMark()
        self._next_argmin = None  # (x, y) of the pass already run for the coming iteration (see step)
        self.last = (None, None, None)  # (x, y, candidates) of the last iteration (bench.py: bytes of the improvement step)

    def step(self, more=True):
        """One iteration.  ``more``: another iteration will follow (the default; ``sdrf_no_cuda`` passes False for the last
        one): the coming iteration's curvature pass and first minimum are then enqueued right behind this iteration's edit
        and share its host synchronisation (``dcr_sdrf_tail_at_pass_argmin``): two host round trips per iteration, not three."""
        G, curv_type, tau, trace = self.G, self.curv_type, self.tau, self.trace
        want_trace = trace is not None
        can_add = True
        # Full curvature pass, then the edge with the lowest curvature (first minimum in G.edges order).
        if self._next_argmin is not None:
            x, y = self._next_argmin
            self._next_argmin = None
        else:
            try:
                x, y, _ = G.curvature_pass_argmin(curv_type, incremental=self.incremental)
            except KeyError:
                raise ValueError('min() arg is an empty sequence')  # what the reference's min() raises
        rec = {'argmin': [x, y]} if want_trace else None

        if self.device_draw and more and not want_trace and (np.isfinite(tau) or tau == float('inf')):
            # The whole iteration without bringing the improvements to the host: the uniform np.random.choice would take is
            # taken here and the draw itself runs on the device (dcr_sdrf_iteration_device_draw), accepted only when it is
            # certain to be numpy's index.  Otherwise nothing was edited: the uniform goes back into the stream and the
            # iteration runs the long way below (numpy's own exp, sum and cumsum on the host).
            # (single-threaded contract, as the reference's own use of the global stream: nothing else may draw from numpy's
            #  legacy generator between mark() and rewind())
            self._mark.mark()
            uniform = np.random.random_sample()
            try:
                status, n_cand, _, _, nxt = G.sdrf_iteration_device_draw(x, y, curv_type, tau, uniform, self.remove_edges,
                                                                        self.removal_bound, incremental=self.incremental)
            except BaseException:
                self._mark.rewind()   # the call failed before any edit was accepted: the uniform was not numpy's to lose
                raise
            if status == 0:
                self._next_argmin = nxt[:2]
                self.last = (x, y, int(n_cand))
                self.device_draws += 1
                return True
            self._mark.rewind()
            if status == 2:
                self.no_candidate_iterations += 1   # np.random.choice is never reached (sdrf_no_cuda.py:47-52)
            else:
                self.host_draws += 1                # left undecided by the margin test: numpy draws on the host below

        k = l = idx = None
        if tau == float('inf') and not want_trace:
            # softmax is one-hot at the first arg-max (utils/softmax.py:5-8): the draw is that
            # index whatever the uniform is; consume the one double np.random.choice would.
            n_cand = G.improvements_count(x, y, curv_type)
            if n_cand:
                idx = G.improvements_argmax()
                np.random.random_sample()
        else:
            imp, ci, cj = G.improvements(x, y, curv_type, want_candidates=want_trace)
            n_cand = imp.shape[0]
            if n_cand:
                # tau = inf (only reached here when tracing): softmax is the one-hot special case of utils/softmax.py:5-8
                idx = draw_index(imp, tau) if np.isfinite(tau) else choice_index(softmax(np.array(imp), tau=tau))
                if want_trace:
                    rec['candidates'] = np.stack([ci, cj], 1).tolist()
                    rec['improvements'] = imp.tolist()
                    rec['choice'] = idx
        self.last = (x, y, int(n_cand))
        if want_trace and not n_cand:
            rec.update(candidates=[], improvements=[], choice=None)

        if not n_cand:
            can_add = False
            if not self.remove_edges:
                if want_trace:
                    rec.update(added=None, removed=None)
                    trace.append(rec)
                return False

        # add candidate idx = (k, l), looked up on the device; then the stale arg-max (excluding the new edge) is
        # removed if above the bound
        if n_cand and more:
            # (an edge was added, so the loop goes on whatever the removal step does: the next pass can follow at once)
            (k, l), removed, nxt = G.sdrf_tail_at_pass_argmin(idx, self.remove_edges, self.removal_bound, curv_type,
                                                             incremental=self.incremental)
            self._next_argmin = nxt[:2]
        elif n_cand:
            (k, l), removed, _ = G.sdrf_tail_at(idx, self.remove_edges, self.removal_bound)
        else:
            removed, _ = G.sdrf_tail(None, self.remove_edges, self.removal_bound)
        if want_trace:
            rec['added'] = [k, l] if n_cand else None
            rec['removed'] = list(removed) if removed else None
            trace.append(rec)
        if self.remove_edges and removed is None and can_add is False:
            return False
        return True

    def result(self):
        return _make_data(self.data, self.G.to_edge_index())


def sdrf_no_cuda(data, curv_type, loops, remove_edges, removal_bound, tau, trace=None, device=0, incremental=None):
    """
    Perform SDRF graph rewiring using the given discrete curvature type.
    :param data: data to be rewired (undirected by default in this work).
    :param curv_type: '1d' | 'augmented' | 'haantjes' | 'bfc'.
    :param loops: number of edge addition/deletion iterations.
    :param remove_edges: whether to delete highly curved edges each iteration to compensate for the addition.
    :param removal_bound: curvature lower bound of deleting edges (delete edges only with higher curvature).
    :param tau: softmax temperature for choosing the edge to add; if infinite, the max value is chosen.
    :return: rewired data.
    """
    run = SdrfRun(data, curv_type, remove_edges, removal_bound, tau, trace=trace, device=device,
                  incremental=incremental)
    for i in range(loops):
        if not run.step(more=i + 1 < loops):
            break
    return run.result()
