"""numerics='bfc_cuda' (SURVEY.md §8 rows f3, f4): the reference's dense float32 Balanced Forman path on the MI355X —
csrc/dcr_bfc_dense.hip behind curvature/bfc_cuda.py, and the loop of rewiring/sdrf_cuda_bfc.py:14-93 for undirected and
directed graphs — against fixtures recorded by executing the reference's two files (tools/make_golden_cuda_compat.py)
and against the CPU oracle (oracle/bfc_cuda_oracle.py) on seeded graphs.  float32 results compared bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_golden import _dense, _f32hex, check_bfc_cuda_sdrf_case

pytestmark = pytest.mark.gpu


def test_dense_kernels_match_the_reference_kernels():
    from curvature.bfc_cuda import balanced_forman_curvature, balanced_forman_post_delta
    for case in load_golden('bfc_cuda_curvature.json')['cases']:
        A = torch.from_numpy(_dense(case)).cuda()
        C = balanced_forman_curvature(A, numerics='bfc_cuda')
        assert C.dtype == torch.float32 and C.is_cuda
        assert _f32hex(C.cpu().numpy()) == case['C'], case['graph']
        C2 = torch.full_like(C, 7.0)                       # a caller-provided C is overwritten everywhere
        assert balanced_forman_curvature(A, C=C2, numerics='bfc_cuda') is C2 and torch.equal(C2, C)
        for pd in case['post_delta']:
            D = balanced_forman_post_delta(A, pd['x'], pd['y'], pd['i_neighbors'], pd['j_neighbors'], numerics='bfc_cuda')
            assert _f32hex(D.cpu().numpy()) == pd['D'], (case['graph'], pd['x'], pd['y'])


def _run_product(ei, n, loops, remove_edges, bound, tau, undirected, trace):
    from dcr.data import Data
    from rewiring.sdrf_cuda_bfc import sdrf_cuda_bfc
    out = sdrf_cuda_bfc(Data(edge_index=torch.from_numpy(ei), num_nodes=n), loops, remove_edges, bound, tau, undirected,
                        trace=trace, numerics='bfc_cuda')
    return out.edge_index.numpy()


def test_sdrf_cuda_bfc_reproduces_the_reference_runs():
    for case in load_golden('bfc_cuda_sdrf.json')['cases']:
        check_bfc_cuda_sdrf_case(case, _run_product)


@pytest.mark.parametrize('undirected', [True, False])
def test_sdrf_cuda_bfc_vs_oracle_on_seeded_graphs(undirected):
    from oracle import bfc_cuda_oracle as bo
    rng = np.random.Generator(np.random.PCG64(23 + undirected))
    n = 70
    m = rng.random((n, n)) < 0.07
    np.fill_diagonal(m, False)
    if undirected:
        m = m | m.T
    src, dst = np.nonzero(m)
    ei = np.stack([src, dst]).astype(np.int64)
    for tau, seed in ((25.0, 0), (float('inf'), 1)):
        ta, tb = [], []
        np.random.seed(seed)
        want = bo.sdrf_cuda_bfc(ei, n, 15, True, 0.4, tau, undirected, trace=ta)
        np.random.seed(seed)
        got = _run_product(ei, n, 15, True, 0.4, tau, undirected, tb)
        assert len(ta) == len(tb)
        for it, (a, b) in enumerate(zip(ta, tb)):
            assert a['argmin'] == b['argmin'] and a['n_candidates'] == b['n_candidates'], it
            assert a['improvements'] == b['improvements'] and a['choice'] == b['choice'], it
            assert [list(e) for e in a['events']] == [list(e) for e in b['events']], it
        assert np.array_equal(want, got)


def test_numerics_switch_and_errors():
    from curvature import bfc_cuda
    from dcr.data import Data
    from rewiring.rewire import rewire
    from rewiring.sdrf_cuda_bfc import sdrf_cuda_bfc
    case = load_golden('bfc_cuda_sdrf.json')['cases'][0]
    ei = torch.tensor(case['edge_index'])
    data = Data(edge_index=ei, num_nodes=case['num_nodes'])
    with pytest.raises(ValueError):                      # directed graphs exist for the dense numerics only
        sdrf_cuda_bfc(data, 2, True, 0.5, 10.0, False)
    with pytest.raises(RuntimeError):                    # the dense kernels take a device tensor, as the reference's do
        bfc_cuda.balanced_forman_curvature(torch.zeros(4, 4), numerics='bfc_cuda')
    assert bfc_cuda.get_numerics() == 'bfc_naive'
    np.random.seed(case['seed'])
    naive = rewire(data, 'bfc', case['loops'], case['removal_bound'], case['tau'])
    bfc_cuda.set_numerics('bfc_cuda')                    # rewire('bfc') now runs the reference's own numerics
    try:
        np.random.seed(case['seed'])
        dense = rewire(data, 'bfc', case['loops'], case['removal_bound'], case['tau'])
    finally:
        bfc_cuda.set_numerics('bfc_naive')
    assert dense.tolist() == case['final_edge_index']
    assert naive.tolist() != dense.tolist()              # the two numerics really rewire differently
