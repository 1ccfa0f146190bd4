"""The randomised / long-running GPU-box checkers with a small fixed budget, so that every round's GPU test run exercises all
three curvature-pass engines (node-centric, edge-centric, two-hop), the incremental pass, SDRF runs with random parameters,
hubs beyond the LDS tables and the bfc_cuda compatibility mode against the oracles (which are pinned to the reference's own
outputs: curvature/bfc_naive.py:7-40, rewiring/sdrf_no_cuda.py:22-66, curvature/bfc_cuda.py, rewiring/sdrf_cuda_bfc.py).
The long versions stay runnable as scripts (tests/fuzz_parity.py, fuzz_engines.py, fuzz_bfc_cuda.py, check_soak.py, check_hub_sdrf.py)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('engine,seed', [('nc', 11), ('edge', 12), ('h2', 13)])
def test_fuzz_parity_small_budget(engine, seed):
    import fuzz_parity
    graphs, values, runs = fuzz_parity.run(seed=seed, seconds=90.0, graphs=14, engines=(engine,), hub_prob=0.0, verbose=False)
    assert graphs == 14 and values > 0


def test_fuzz_parity_all_engines_on_the_same_graphs():
    import fuzz_parity
    graphs, values, runs = fuzz_parity.run(seed=5, seconds=90.0, graphs=8, engines=('nc', 'edge', 'h2'), hub_prob=0.0, verbose=False)
    assert graphs == 8 and values > 0


def test_hub_graph_through_every_engine():
    """One graph with hubs joined to each other whose degrees exceed every LDS table (device-memory path; the two-hop engine
    declines such graphs and the node-centric one takes over)."""
    import fuzz_parity
    graphs, values, runs = fuzz_parity.run(seed=3, seconds=240.0, graphs=1, engines=('nc', 'edge', 'h2'), hub_prob=1.0, verbose=False)
    assert graphs == 1 and values > 0


def test_engines_agree_on_medium_graphs():
    """tests/fuzz_engines.py with a small budget: graphs of 5 k - 150 k nodes through the two-hop kernels, the node-centric class
    kernels and the edge-by-edge kernels (round 5) — the same bits — and an incremental pass behind three edits at the hubs."""
    import fuzz_engines
    assert fuzz_engines.run(seed=4, seconds=12.0) >= 5


def test_soak_200_iterations_incremental_equals_full():
    import check_soak
    assert check_soak.run(iters=200, samples=4000) == 200


def test_sdrf_on_two_adjacent_hubs():
    import check_hub_sdrf
    assert check_hub_sdrf.run(cases=((float('inf'), 2),))


def test_fuzz_bfc_cuda_small_budget():
    import fuzz_bfc_cuda
    graphs, values, runs = fuzz_bfc_cuda.run(seed=7, seconds=120.0, graphs=40)
    assert graphs == 40 and values > 0
