"""GPU end-to-end test of the experiment layer: rewire -> split -> GCN -> training_loop
-> pickles -> evaluation, the flow of the reference's experiment/save_models.py and experiment/test_performance.py."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def oracle():
    from oracle import c_oracle
    return c_oracle


def _tiny_dataset(folder, n=400, n_feat=24, n_cls=4):
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(n, 3, seed=21)
    rng = np.random.Generator(np.random.PCG64(5))
    y = rng.integers(0, n_cls, n)
    x = rng.standard_normal((n, n_feat)).astype(np.float32)
    x[np.arange(n), y] += 3.0                        # learnable: the label shows in the features
    np.savez(os.path.join(folder, 'Tiny.npz'), x=x, y=y, edge_index=ei)
    return ei, n


def test_save_models_then_test_performance(tmp_path, oracle):
    from experiment.save_models import save_models, _dump
    from experiment.test_performance import test_performance as evaluate_saved, summary_cell
    from utils.hyperparams import hyperparams
    from utils.seeds import val_seeds
    tmp = str(tmp_path)
    ei, n = _tiny_dataset(tmp)
    hp = hyperparams['Cora']                         # names outside the table train with Cora's values
    np.random.seed(3)
    torch.manual_seed(3)
    sds = save_models('Tiny', 'bfc', patience=5, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:2], epochs=40,
                      verbose=False)
    assert len(sds) == 2
    assert sorted(sds[0]) == ['layers.0.bias', 'layers.0.lin.weight', 'layers.1.bias', 'layers.1.lin.weight']
    assert sds[0]['layers.0.lin.weight'].shape == (hp['hidden_dim'], 24) and not sds[0]['layers.0.bias'].is_cuda
    with open(os.path.join(tmp, 'edge_indices', 'Tiny', 'edge_index_bfc.pk'), 'rb') as f:
        rewired = pickle.load(f)
    assert rewired.dtype == torch.int64 and not rewired.is_cuda and rewired.shape[0] == 2
    # the pickled edge list is exactly what the CPU oracle's SDRF gives for the same numpy stream
    np.random.seed(3)
    want = oracle.sdrf(ei, n, 'bfc', hp['max_iterations'], True, hp['removal_bound'], hp['tau'], nthreads=8)
    assert np.array_equal(rewired.numpy(), want)
    _dump(sds, os.path.join(tmp, 'state_dicts', 'Tiny', 'state_dicts_bfc.pk'))
    accs, mean, std = evaluate_saved('Tiny', 'bfc', test=False, data_dir=tmp, out_dir=tmp)
    assert len(accs) == 2 and all(0.5 < a <= 1.0 for a in accs), accs      # 4 classes, separable features
    test_accs, _, _ = evaluate_saved('Tiny', 'bfc', test=True, data_dir=tmp, out_dir=tmp)
    assert len(test_accs) == 2 and all(0.0 <= a <= 1.0 for a in test_accs)
    assert '+-' in summary_cell(mean, std)
    # curvature None: no rewiring, the stored edge list is the input's
    sd0 = save_models('Tiny', None, patience=3, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:1], epochs=5, verbose=False)
    with open(os.path.join(tmp, 'edge_indices', 'Tiny', 'edge_index_None.pk'), 'rb') as f:
        assert np.array_equal(pickle.load(f).numpy(), ei)
    assert len(sd0) == 1


def test_redo_rewiring_writes_one_edge_list_per_seed(tmp_path):
    from experiment.save_models import save_models
    from utils.seeds import val_seeds
    tmp = str(tmp_path)
    _tiny_dataset(tmp, n=200)
    np.random.seed(0)
    save_models('Tiny', 'augmented', patience=2, redo_rewiring=True, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:2],
                epochs=3, verbose=False)
    folder = os.path.join(tmp, 'edge_indices', 'Tiny_redo_rewiring', 'augmented')
    assert sorted(os.listdir(folder)) == ['edge_index_augmented_00.pk', 'edge_index_augmented_01.pk']
