"""Evaluate saved models — call surface of the reference's experiment/test_performance.py:16-107 (an evaluation script,
not a unit test; pytest does not collect it: ``__test__`` is False).

``test_performance(dname, curv_type, test=True, redo_rewiring=False)`` loads the pickles written by
experiment/save_models.py (this build's or the reference's: same paths, same ``state_dict`` keys), rebuilds the seeded
split per seed, and returns ``(accuracies, mean, std)`` over the test seeds (or the validation seeds when ``test`` is
false).  The reference then reports ``mean*100 +- 1.96*std/sqrt(100)`` (test_performance.py:98); ``summary_cell``
formats that cell.
"""
import os
import pickle
import random

import numpy as np

from experiment.data_loader import DataLoader
from experiment.save_models import hyperparams_for, split_for
from experiment.training_loop import evaluate
from models.gcn import GCN
from utils.seeds import test_seeds, val_seeds

__test__ = False


def _load(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def test_performance(dname, curv_type, test=True, redo_rewiring=False, data_dir='dt', out_dir='.', device='cuda:0'):
    hp = hyperparams_for(dname)
    state_dicts = _load(os.path.join(out_dir, 'state_dicts', dname, f'state_dicts_{curv_type}.pk'))
    dataset = DataLoader(dname, undirected=True, data_dir=data_dir)
    if not redo_rewiring:
        dataset.data.edge_index = _load(os.path.join(out_dir, 'edge_indices', dname, f'edge_index_{curv_type}.pk'))

    accs = []
    seeds = test_seeds if test else val_seeds
    for i, (seed, state_dict) in enumerate(zip(seeds, state_dicts)):
        random.seed(seed)
        if redo_rewiring and curv_type is not None:
            dataset.data.edge_index = _load(os.path.join(out_dir, 'edge_indices', f'{dname}_redo_rewiring',
                                                         str(curv_type), f'edge_index_{curv_type}_{i:02d}.pk'))
        dataset.data = split_for(dname, seed, dataset.data.to('cpu')).to(device)
        model = GCN(dataset=dataset, hidden=[hp['hidden_dim']] * hp['hidden_depth'], dropout=hp['dropout']).to(device)
        model.load_state_dict(state_dict)
        ed = evaluate(model, dataset.data, test=test)
        accs.append(ed['test_acc'] if test else ed['val_acc'])
    return accs, np.mean(accs), np.std(accs)


def summary_cell(mean, std):
    """'mean*100 +- half-width of the 95 % interval over 100 seeds' (test_performance.py:98)."""
    return f'{round(mean * 100, 2)} +- {round(std * 100 * 0.196, 2)}'


def run(datasets, curvatures, **kwargs):
    """The reference's ``__main__`` table (test_performance.py:89-107), returned as {dataset: [cell per curvature]}
    and written as CSV (the reference writes results.xlsx through pandas/openpyxl)."""
    table = {dname: [] for dname in datasets}
    for dname in datasets:
        for curv in curvatures:
            try:
                _, mean, std = test_performance(dname, curv, **kwargs)
                entry = summary_cell(mean, std)
            except Exception:  # noqa: BLE001 - a missing pickle is recorded as the reference records it
                entry = 'ERROR'
            table[dname].append(entry)
            print(dname, curv, entry)
    result = table
    with open(os.path.join(kwargs.get('out_dir', '.'), 'results.csv'), 'w') as f:
        f.write('curvature,' + ','.join(datasets) + '\n')
        for j, c in enumerate(curvatures):
            f.write(str(c) + ',' + ','.join(result[d][j] for d in datasets) + '\n')
    return result
