"""CPU-side checks (no GPU): the C ABI library loads and exports every declared symbol, the product path
refuses to run without a GPU, host-side helpers match the reference's semantics."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO


def test_library_exports_every_declared_symbol():
    from dcr import _lib
    header = open(os.path.join(REPO, 'include', 'dcr.h')).read()
    declared = set(re.findall(r'\b(dcr_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f'{name} declared in include/dcr.h but not exported'
    assert declared == set(_lib.SIGNATURES), 'ctypes signature table and header disagree'


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_path_fails_loudly_without_gpu():
    from dcr import _lib
    from dcr.graph import DcrGraph
    with pytest.raises(_lib.DcrError):
        DcrGraph(np.array([[1, 0], [0, 1]]), 2)
    from dcr.data import Data
    from rewiring.rewire import rewire
    with pytest.raises(_lib.DcrError):
        rewire(Data(edge_index=torch.tensor([[1, 0], [0, 1]]), num_nodes=2), 'bfc', 1, 0.5, 1.0)
    from models.gcn import spmm
    with pytest.raises(RuntimeError):
        spmm(torch.zeros(2, dtype=torch.int64), torch.zeros(0, dtype=torch.int32), torch.zeros(0), torch.zeros(1, 4), 1)


def test_bad_arguments_rejected_before_any_device_call():
    from dcr.graph import DcrGraph
    with pytest.raises(ValueError):
        DcrGraph(np.array([[0, 1], [0, 0]]), 2)          # self-loop
    with pytest.raises(ValueError):
        DcrGraph(np.array([[5, 0], [0, 5]]), 3)          # id out of range
    with pytest.raises(ValueError):
        DcrGraph(np.zeros((3, 2), dtype=np.int64), 3)    # wrong shape
    with pytest.raises(Exception):
        from dcr.graph import curv_code
        curv_code('ollivier')


def test_softmax_semantics():
    from utils.softmax import softmax
    a = np.array([0.1, 0.7, 0.7, -0.2])
    assert softmax(a, float('inf')).tolist() == [0.0, 1.0, 0.0, 0.0]      # first arg-max
    p = softmax(a, 3.0)
    e = np.exp(a * 3.0)
    assert np.array_equal(p, e / e.sum())
    with np.errstate(over='ignore', invalid='ignore'):
        bad = softmax(np.array([1.0, 2.0]), 5000)
    assert np.isnan(bad).any()
    with pytest.raises(ValueError):
        np.random.choice(2, p=bad)                                          # what the reference's loop raises


def test_tau_inf_draw_consumes_one_uniform():
    """sdrf_no_cuda's tau=inf shortcut replaces np.random.choice(p=one-hot) by one random_sample():
    the legacy stream must stay aligned."""
    onehot = np.zeros(7)
    onehot[3] = 1
    np.random.seed(5)
    idx = np.random.choice(range(7), p=onehot)
    after_choice = np.random.random_sample()
    np.random.seed(5)
    np.random.random_sample()
    after_shortcut = np.random.random_sample()
    assert idx == 3 and after_choice == after_shortcut


def test_synthetic_generator_is_deterministic():
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(500, 4, seed=12345)
    assert n == 500 and ei.shape == (2, 2 * 4 * (500 - 4))
    assert (ei[0] != ei[1]).all()
    key = ei[0] * n + ei[1]
    assert (np.diff(key) > 0).all()                                         # coalesced + sorted
    assert np.array_equal(ei, synthetic.powerlaw_graph(500, 4, seed=12345)[0])
    import hashlib
    assert hashlib.sha1(ei.tobytes()).hexdigest() == hashlib.sha1(
        synthetic.coalesced_edge_index(ei[0], ei[1], n).tobytes()).hexdigest()


def test_data_duck_typing():
    from dcr.data import Data
    d = Data(x=torch.zeros(4, 3), edge_index=torch.zeros(2, 0, dtype=torch.long), val_mask=torch.ones(4, dtype=torch.bool))
    assert d.num_nodes == 4 and d['val_mask'].all() and 'x' in d and d.edge_attr is None


def test_choice_index_matches_numpy_choice():
    from rewiring.sdrf_no_cuda import choice_index
    from utils.softmax import softmax
    rng = np.random.Generator(np.random.PCG64(3))
    for trial in range(200):
        n = int(rng.integers(1, 4000))
        a = rng.choice(np.array([0.0, 0.0, 0.0, 1e-3, -2e-3, 0.01, 0.3, -0.4]), size=n) * rng.random()
        tau = float(rng.choice([0.5, 20, 163, 1000, float('inf')]))
        p = softmax(a, tau)
        np.random.seed(trial)
        want = int(np.random.choice(range(n), p=p))
        after_want = np.random.random_sample()
        np.random.seed(trial)
        got = choice_index(p.copy())
        after_got = np.random.random_sample()
        assert got == want and after_got == after_want, (trial, n, tau)
    with np.errstate(over='ignore', invalid='ignore'):
        bad = softmax(np.array([1.0, 2.0, 3.0]), 5000)
    with pytest.raises(ValueError, match='NaN'):
        choice_index(bad)
    with pytest.raises(ValueError):
        np.random.choice(3, p=bad)


def test_draw_index_matches_numpy_choice():
    """The fused host draw (exp and sum by numpy, divide + sequential cumsum in the library's host helper) returns
    numpy's index and leaves numpy's stream in numpy's state, for vectors like the improvement vectors of a run."""
    from rewiring.sdrf_no_cuda import draw_index
    from utils.softmax import softmax
    rng = np.random.Generator(np.random.PCG64(8))
    for n, tau in ((1, 3.0), (7, 163.0), (1000, 50.0), (185000, 163.0), (4096, 0.5)):
        a = rng.choice(np.array([0.0, 1e-3, -2e-3, 0.0125, 0.0, 0.004]), size=n) + (rng.random(n) < 0.01) * rng.random(n) * 0.05
        for seed in (0, 1, 2):
            np.random.seed(seed)
            want = int(np.random.choice(n, p=softmax(a, tau)))
            state_want = np.random.get_state()[1].copy(), np.random.get_state()[2]
            np.random.seed(seed)
            got = draw_index(a, tau)
            state_got = np.random.get_state()[1], np.random.get_state()[2]
            assert got == want and state_got[1] == state_want[1] and np.array_equal(state_got[0], state_want[0])
    with pytest.raises(ValueError):
        draw_index(np.array([800.0, 1.0]), 163.0)      # exp overflows: numpy's NaN error, stream untouched


def test_vectorised_cdf_is_numpy_cumsum():
    """csrc/dcr_host_draw.cpp evaluates numpy's sequential cumsum(e / S) eight elements at a time where integer-valued
    float64 arithmetic makes that exact, and falls back to the plain loop at ties, binade crossings and unusual values.
    Bit-compared with numpy.cumsum itself (the contract of np.random.choice) and with the plain loop."""
    import ctypes
    from dcr import _lib
    L = _lib.lib()
    f64p = _lib._f64p

    def run(fn, e, S):
        e = np.ascontiguousarray(e, dtype=np.float64)
        out = np.empty_like(e)
        tot = ctypes.c_double()
        assert fn(e.ctypes.data_as(f64p), e.shape[0], float(S), out.ctypes.data_as(f64p), ctypes.byref(tot)) == 0
        assert tot.value == out[-1] or (np.isnan(tot.value) and np.isnan(out[-1]))
        return out

    def check(e, S, label):
        with np.errstate(all='ignore'):
            want = (np.asarray(e, dtype=np.float64) / S).cumsum()
        for fn in (L.dcr_host_cdf_from_exp, L.dcr_host_cdf_from_exp_plain):
            got = run(fn, e, S)
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(got), nan), label
            assert np.array_equal(got[~nan].view(np.uint64), want[~nan].view(np.uint64)), label

    rng = np.random.Generator(np.random.PCG64(1))
    for trial in range(400):
        n = int(rng.integers(64, 5000))
        kind = trial % 10
        a = (rng.normal(0, 0.01, n) * 163 if kind == 0 else rng.normal(0, 0.5, n) * 163 if kind == 1 else
             rng.choice([-0.1, 0.0, 0.02, 0.03], n) * 163 if kind == 2 else rng.uniform(-700, 700, n) if kind == 3 else
             np.sort(rng.normal(0, 0.2, n) * 50) if kind == 4 else -np.sort(rng.normal(0, 0.2, n) * 50) if kind == 5 else
             np.zeros(n) if kind == 6 else np.log(rng.integers(1, 4, n).astype(float)) if kind == 7 else
             rng.normal(0, 3, n) if kind == 8 else rng.normal(0, 1e-6, n))
        with np.errstate(all='ignore'):
            e = np.exp(a)
            check(e, e.sum(), (trial, kind))
    for n in (64, 257, 1000):                      # crafted ties and binade crossings
        for base in (1.0, 0.5, 3.0, 1e-5, 7e10):
            check(np.full(n, base), 1.0, ('const', n, base))
            check(base * 2.0 ** (-rng.integers(0, 60, n).astype(float)), 1.0, ('pow2', n, base))
            check(base * (1.0 + 2.0 ** (-rng.integers(1, 53, n).astype(float))), 3.0, ('near', n, base))
    for label, mod in (('inf', lambda e: e.__setitem__(100, np.inf)), ('nan', lambda e: e.__setitem__(7, np.nan)),
                       ('negative', lambda e: e.__setitem__(300, -0.25))):
        e = rng.random(500)
        mod(e)
        check(e, 1.0, label)
    check(np.zeros(300), 1.0, 'zeros')
    check(rng.random(300) * 1e-310, 1.0, 'subnormal')
    check(rng.random(300) * 1e300, 1e-5, 'overflow')
    # the draw built on it still agrees with numpy's choice
    np.random.seed(3)
    a = rng.normal(0, 0.01, 5000)
    want = np.random.choice(5000, p=np.exp(a * 163) / np.exp(a * 163).sum())
    np.random.seed(3)
    from rewiring.sdrf_no_cuda import draw_index
    assert draw_index(a, 163.0) == want


def test_ordered_graph_follows_networkx():
    """dcr/ordered_graph.py is the host-side graph of the dense SDRF loop (rewiring/sdrf_cuda_bfc.py:31-33,44-54,69,85,93
    use an nx.Graph / nx.DiGraph there): same neighbour / successor / predecessor order as networkx under random edits,
    DiGraph.to_undirected() order, and the edge order of PyG's from_networkx (relabel, then to_directed().edges)."""
    import networkx as nx
    from dcr.ordered_graph import OrderedDiGraph, digraph_from_edge_index
    rng = np.random.Generator(np.random.PCG64(12))
    for trial in range(30):
        n = int(rng.integers(3, 25))
        m = int(rng.integers(0, 80))
        ei = rng.integers(0, n, size=(2, m))
        ei = ei[:, ei[0] != ei[1]]
        D = nx.DiGraph()
        D.add_nodes_from(range(n))
        D.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
        O = digraph_from_edge_index(ei, n)
        assert isinstance(O, OrderedDiGraph)
        U, OU = D.to_undirected(), O.to_undirected()
        for _ in range(40):  # the same random edits on all four
            a, b = (int(t) for t in rng.integers(0, n, size=2))
            if a == b:
                continue
            if rng.random() < 0.6:
                D.add_edge(a, b); O.add_edge(a, b); U.add_edge(a, b); OU.add_edge(a, b)
            else:
                if D.has_edge(a, b):
                    D.remove_edge(a, b); O.remove_edge(a, b)
                if U.has_edge(a, b):
                    U.remove_edge(a, b); OU.remove_edge(a, b)
            assert D.has_edge(a, b) == O.has_edge(a, b) and U.has_edge(a, b) == OU.has_edge(a, b)
        for v in range(n):
            assert list(D.successors(v)) == O.successors(v) and list(D.predecessors(v)) == O.predecessors(v)
            assert list(U.neighbors(v)) == OU.neighbors(v)
        assert [list(e) for e in D.edges] == O.to_edge_index().T.tolist()
        H = nx.convert_node_labels_to_integers(U).to_directed()      # what from_networkx lists for an undirected graph
        assert [list(e) for e in H.edges] == OU.to_edge_index().T.tolist()
        assert OU.number_of_edges() == U.number_of_edges()


def test_stream_mark_puts_one_uniform_back():
    """rewiring/sdrf_no_cuda.py::_StreamMark (the device-side draw takes the uniform of np.random.choice before it knows
    whether numpy would have taken one): mark / draw / rewind leaves numpy's global stream exactly where get_state /
    set_state would, at every position of the Mersenne Twister's block including the one where a draw regenerates it, and
    the self-check it runs first leaves the stream untouched."""
    from rewiring.sdrf_no_cuda import _StreamMark
    np.random.seed(12345)
    before = np.random.get_state()
    assert _StreamMark._self_check() is True
    after = np.random.get_state()
    assert np.array_equal(before[1], after[1]) and before[2] == after[2]
    m = _StreamMark()
    for start in (0, 1, 310, 311, 312, 623, 1000):
        np.random.seed(7)
        np.random.random_sample(start)
        want_state = np.random.get_state()
        m.mark()
        u = np.random.random_sample()
        m.rewind()
        got_state = np.random.get_state()
        assert np.array_equal(want_state[1], got_state[1]) and want_state[2:] == got_state[2:], start
        assert np.random.random_sample() == u
    # a gaussian cached by the legacy stream survives (it lives outside the bit generator's state)
    np.random.seed(3)
    np.random.standard_normal()
    st = np.random.get_state()
    m.mark(); np.random.random_sample(); m.rewind()
    st2 = np.random.get_state()
    assert st[3:] == st2[3:] and np.array_equal(st[1], st2[1]) and st[2] == st2[2]
