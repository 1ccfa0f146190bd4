"""Seeded train / validation / test masks — call surface of the reference's experiment/data_splits.py:15-69.

Same functions, same arguments, same index sets for a given seed (pinned against the reference's own output in
tests/golden/experiment_helpers.json).  The random draws are kept exactly where the reference makes them
(``np.random.RandomState.choice`` and ``random.shuffle`` consume their streams identically); the membership filters the
reference writes as O(n^2) list scans (data_splits.py:29,48,56) are array set differences with the same result.
"""
import random
from math import ceil

import numpy as np
import torch

from utils.seeds import development_seed


def get_mask(idx, num_nodes):
    """Boolean node mask with ``idx`` set (data_splits.py:66-69)."""
    mask = torch.zeros(num_nodes, dtype=torch.bool)
    mask[torch.as_tensor(np.asarray(idx, dtype=np.int64))] = True
    return mask


def _shuffled(items, seed):
    """``items`` permuted by ``random.shuffle`` under ``random.seed(seed)`` (the generator the reference uses here)."""
    order = list(items)
    random.seed(seed)
    random.shuffle(order)
    return order


def _install_masks(data, num_nodes, train_idx, val_idx, test_idx):
    for name, idx in (('train_mask', train_idx), ('val_mask', val_idx), ('test_mask', test_idx)):
        setattr(data, name, get_mask(idx, num_nodes))
    return data


def set_train_val_test_split(seed, data, num_development=1500, num_per_class=20):
    """Planetoid-style split (data_splits.py:45-63): a development set of ``num_development`` nodes drawn with the
    fixed development seed, everything else is test; per class up to ``num_per_class`` development nodes (at most 70 %
    of the class) drawn with ``seed`` are train, the remaining development nodes are validation."""
    num_nodes = int(data.y.shape[0])
    y = data.y.cpu().numpy() if hasattr(data.y, 'cpu') else np.asarray(data.y)
    dev = np.random.RandomState(development_seed).choice(num_nodes, num_development, replace=False)
    in_dev = np.zeros(num_nodes, dtype=bool)
    in_dev[dev] = True
    test_idx = np.nonzero(~in_dev)[0]

    rnd = np.random.RandomState(seed)
    train_parts = []
    y_dev = y[dev]
    for c in range(int(y.max()) + 1):
        class_idx = dev[np.where(y_dev == c)[0]]
        train_parts.append(rnd.choice(class_idx, min(num_per_class, int(len(class_idx) * 0.7)), replace=False))
    train_idx = np.concatenate(train_parts) if train_parts else np.empty(0, dtype=np.int64)
    in_train = np.zeros(num_nodes, dtype=bool)
    in_train[train_idx] = True
    val_idx = dev[~in_train[dev]]

    return _install_masks(data, num_nodes, train_idx, val_idx, test_idx)


def set_train_val_test_split_frac(seed, data, val_frac, test_frac):
    """Fractional split (data_splits.py:15-42): the test nodes are the head of a shuffle with the development seed
    (the same test set for every seed); the rest, in the order that first shuffle left it, is shuffled with ``seed``
    and cut into train and validation."""
    num_nodes = int(data.y.shape[0])
    n_val, n_test = ceil(val_frac * num_nodes), ceil(test_frac * num_nodes)
    n_train = num_nodes - n_val - n_test

    first = _shuffled(range(num_nodes), development_seed)
    test_idx, rest = first[:n_test], first[n_test:]   # (the reference filters the test nodes out again: same list)
    second = _shuffled(rest, seed)
    train_idx, val_idx = second[:n_train], second[n_train:]
    if len(train_idx) + len(val_idx) + len(test_idx) != num_nodes:
        raise AssertionError('split sizes do not add up')
    return _install_masks(data, num_nodes, train_idx, val_idx, test_idx)
