"""CPU tests of the experiment layer: splits, LCC extraction, the Planetoid raw
reader, hyper-parameter / seed tables — against outputs of the reference's own helpers (tests/golden/
experiment_helpers.json, written by tools/make_golden_experiment.py)."""
import json
import os
import pickle

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from dcr.data import Data

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'experiment_helpers.json')


@pytest.fixture(scope='module')
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _idx(mask):
    return torch.nonzero(mask).flatten().tolist()


def test_hyperparams_and_seeds_match_reference(golden):
    from utils.hyperparams import hyperparams
    from utils import seeds
    assert hyperparams == golden['hyperparams']
    assert seeds.development_seed == golden['development_seed']
    assert len(seeds.val_seeds) == golden['n_val_seeds'] and len(seeds.test_seeds) == golden['n_test_seeds']
    assert seeds.val_seeds[:4] == golden['val_seeds_head'] and seeds.test_seeds[:4] == golden['test_seeds_head']
    assert sum(seeds.val_seeds) == golden['seed_checksums']['val']
    assert sum(seeds.test_seeds) == golden['seed_checksums']['test']


def test_splits_match_reference(golden):
    from experiment.data_splits import set_train_val_test_split, set_train_val_test_split_frac
    for case in golden['splits']:
        if case['kind'] == 'planetoid':
            d = set_train_val_test_split(case['seed'], Data(y=torch.tensor(case['y'])))
        else:
            d = set_train_val_test_split_frac(case['seed'], Data(y=torch.zeros(case['num_nodes'], dtype=torch.long)),
                                              val_frac=case['val_frac'], test_frac=case['test_frac'])
        assert _idx(d.train_mask) == case['train'] and _idx(d.val_mask) == case['val']
        assert _idx(d.test_mask) == case['test']
        assert not (d.train_mask & d.val_mask).any() and not (d.train_mask & d.test_mask).any()


def test_lcc_matches_reference(golden):
    from experiment.data_loader import get_largest_connected_component, restrict_to_nodes
    for case in golden['lcc']:
        ei = np.array(case['edge_index'], dtype=np.int64).reshape(2, -1)
        n = case['num_nodes']
        lcc = get_largest_connected_component(ei, n)
        assert lcc.tolist() == sorted(case['lcc'])
        # the reference relabels in the iteration order of its node set; fixtures record that order
        assert case['lcc'] == sorted(case['lcc'])
        x = np.arange(n, dtype=np.float32)[:, None]
        xs, ys, es = restrict_to_nodes(x, np.arange(n), ei, lcc)
        assert es.tolist() == case['remapped']
        assert xs[:, 0].tolist() == [float(t) for t in case['lcc']] and ys.tolist() == case['lcc']


def _write_planetoid(folder, name, x, y_onehot, adj, n_train, test_ids):
    """Write the eight ind.* files the way the Planetoid distribution lays them out."""
    os.makedirs(folder, exist_ok=True)
    n = x.shape[0]
    n_test = len(test_ids)
    n_all = n - n_test
    order_test = np.array(test_ids)
    parts = {'x': sp.csr_matrix(x[:n_train]), 'y': y_onehot[:n_train], 'allx': sp.csr_matrix(x[:n_all]),
             'ally': y_onehot[:n_all], 'tx': sp.csr_matrix(x[order_test]), 'ty': y_onehot[order_test],
             'graph': {u: list(vs) for u, vs in adj.items()}}
    for k, v in parts.items():
        with open(os.path.join(folder, f'ind.{name}.{k}'), 'wb') as f:
            pickle.dump(v, f)
    with open(os.path.join(folder, f'ind.{name}.test.index'), 'w') as f:
        f.write('\n'.join(str(t) for t in test_ids) + '\n')


def test_planetoid_reader_and_dataloader(tmp_path):
    from experiment.data_loader import DataLoader, read_planetoid
    rng = np.random.Generator(np.random.PCG64(3))
    n, n_feat, n_cls, n_test = 40, 7, 3, 10
    x = (rng.random((n, n_feat)) < 0.3).astype(np.float32)
    y = rng.integers(0, n_cls, n)
    y1 = np.eye(n_cls, dtype=np.int32)[y]
    adj = {u: set() for u in range(n)}
    for _ in range(60):
        a, b = (int(t) for t in rng.integers(0, 34, 2))   # nodes 34..39 stay isolated or in a small component
        adj[a].add(b)
    adj[36].add(37); adj[3].add(3)                         # a two-node component and a self-loop
    test_ids = list(rng.permutation(np.arange(n - n_test, n)))
    folder = tmp_path / 'Cora' / 'raw'
    # the test rows of the distribution are stored in the order of test.index and put back by it
    _write_planetoid(str(folder), 'cora', x, y1, adj, 12, [int(t) for t in test_ids])
    xr, yr, ei = read_planetoid(str(folder), 'Cora')
    assert np.array_equal(xr, x) and np.array_equal(yr, y)
    want = set()
    for u, vs in adj.items():
        for v in vs:
            if u != v:
                want.add((u, v)); want.add((v, u))
    assert set(map(tuple, ei.T.tolist())) == want and ei.shape[1] == len(want)
    assert np.all(np.diff(ei[0] * n + ei[1]) > 0)          # coalesced, sorted
    ds = DataLoader('Cora', use_lcc=True, undirected=True, data_dir=str(tmp_path))
    assert str(ds) == 'Cora_undirected_lcc=True'
    k = ds.data.num_nodes
    assert k < n and ds.data.x.shape == (k, n_feat) and ds.num_classes == len(np.unique(ds.data.y.numpy()))
    assert int(ds.data.edge_index.max()) == k - 1 and not ds.data.train_mask.any()
    with pytest.raises(FileNotFoundError):
        DataLoader('Pubmed', data_dir=str(tmp_path))


def test_npz_and_synthetic_sources(tmp_path):
    from experiment.data_loader import DataLoader
    ei = np.array([[0, 1, 1, 2, 4, 5], [1, 0, 2, 1, 5, 4]])
    np.savez(tmp_path / 'Tiny.npz', x=np.eye(6, dtype=np.float32), y=np.array([5, 5, 9, 9, 7, 7]), edge_index=ei)
    ds = DataLoader('Tiny', use_lcc=True, undirected=True, data_dir=str(tmp_path))
    assert ds.data.num_nodes == 3 and ds.data.y.tolist() == [0, 0, 1] and ds.num_classes == 2
    assert ds.data.edge_index.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]]
    base = DataLoader('Tiny', use_lcc=False, undirected=False, data_dir=str(tmp_path))
    assert base.data.num_nodes == 6 and base.data.edge_attr.shape[0] == 6 and base.num_classes == 3
    syn = DataLoader('synthetic:300:3:8:4', use_lcc=True, undirected=True)
    assert syn.data.x.shape == (300, 8) and syn.num_classes == 4


def test_training_loop_reproduces_the_reference_run():
    """the per-epoch (loss, val_acc) sequence, the stopping epoch and the returned weights of the
    reference's own training_loop (experiment/training_loop.py:22-37), recorded in the build container."""
    import training_loop_fixture as fx
    torch.set_num_threads(1)
    for case in fx.cases():
        model, data, losses, accs = fx.run_recorded(case)
        fx.check(case, model, data, losses, accs)
