#!/usr/bin/env python3
"""Generate tests/golden/experiment_*.json by RUNNING THE REFERENCE's experiment helpers (build container only).

Imports the reference's ``experiment/data_splits.py`` and ``experiment/data_loader.py`` unmodified; PyG is not
installed here, so stand-in ``torch_geometric.data`` / ``torch_geometric.datasets`` modules provide the names those
files import (a duck-typed ``Data``, an empty ``InMemoryDataset`` base, placeholder dataset classes that are never
instantiated).  Only numbers are written: inputs (labels, edge lists, seeds) and outputs (index lists).

Usage:  python tools/make_golden_experiment.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLDEN = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, os.path.join(REPO, 'discrete-curvature-rewiring_amd'))
from dcr.data import Data  # noqa: E402


class _InMemoryDataset:
    pass


def install_shims():
    tg = types.ModuleType('torch_geometric')
    tgd = types.ModuleType('torch_geometric.data')
    tgd.Data = Data
    tgd.InMemoryDataset = _InMemoryDataset
    tgs = types.ModuleType('torch_geometric.datasets')
    for name in ('Planetoid', 'Amazon', 'Coauthor', 'WebKB', 'WikipediaNetwork', 'Actor'):
        setattr(tgs, name, type(name, (), {}))
    tg.data, tg.datasets = tgd, tgs
    sys.modules.update({'torch_geometric': tg, 'torch_geometric.data': tgd, 'torch_geometric.datasets': tgs})
    sys.path.insert(0, REF)


install_shims()
for m in [k for k in sys.modules if k.split('.')[0] in ('experiment', 'utils')]:
    del sys.modules[m]
import experiment.data_loader as ref_loader  # noqa: E402  (reference)
import experiment.data_splits as ref_splits  # noqa: E402  (reference)
import utils.seeds as ref_seeds  # noqa: E402  (reference)

assert ref_splits.__file__.startswith(REF) and ref_loader.__file__.startswith(REF)


def idx(mask):
    return torch.nonzero(mask).flatten().tolist()


def split_cases():
    cases = []
    rng = np.random.Generator(np.random.PCG64(7))
    for n, c in ((1700, 6), (2485, 7)):
        y = rng.integers(0, c, n)
        for seed in ref_seeds.val_seeds[:2] + ref_seeds.test_seeds[:1]:
            d = ref_splits.set_train_val_test_split(seed, Data(y=torch.from_numpy(y)))
            cases.append({'kind': 'planetoid', 'seed': int(seed), 'y': y.tolist(), 'train': idx(d.train_mask),
                          'val': idx(d.val_mask), 'test': idx(d.test_mask)})
    for n in (183, 251, 2277):
        y = rng.integers(0, 5, n)
        for seed in ref_seeds.val_seeds[2:4]:
            d = ref_splits.set_train_val_test_split_frac(seed, Data(y=torch.from_numpy(y)), val_frac=0.2, test_frac=0.2)
            cases.append({'kind': 'frac', 'seed': int(seed), 'num_nodes': n, 'val_frac': 0.2, 'test_frac': 0.2,
                          'train': idx(d.train_mask), 'val': idx(d.val_mask), 'test': idx(d.test_mask)})
    return cases


def lcc_cases():
    cases = []
    rng = np.random.Generator(np.random.PCG64(11))
    for n, p, extra in ((60, 0.03, 0), (200, 0.008, 3), (400, 0.004, 0), (50, 0.0, 0)):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(iu.shape[0]) < p
        src = np.concatenate([iu[keep], ju[keep]])
        dst = np.concatenate([ju[keep], iu[keep]])
        if extra:  # a few one-directional edges and a self-loop, as raw files may hold
            a = rng.integers(0, n, extra)
            b = rng.integers(0, n, extra)
            src = np.concatenate([src, a, [5]])
            dst = np.concatenate([dst, b, [5]])
        order = rng.permutation(src.shape[0])
        ei = np.stack([src[order], dst[order]]).astype(np.int64)
        ds = types.SimpleNamespace(data=Data(x=torch.zeros(n, 1), edge_index=torch.from_numpy(ei)))
        lcc = ref_loader.get_largest_connected_component(ds)
        mapper = ref_loader.get_node_mapper(lcc)
        row, col = ei
        edges = [[int(i), int(j)] for i, j in zip(row, col) if i in lcc and j in lcc]
        remapped = ref_loader.remap_edges(edges, mapper) if edges else [[], []]
        cases.append({'num_nodes': n, 'edge_index': ei.tolist(), 'lcc': [int(t) for t in lcc],
                      'remapped': [[int(t) for t in remapped[0]], [int(t) for t in remapped[1]]]})
    return cases


def main():
    out = {'_about': 'outputs of the reference experiment/data_splits.py and experiment/data_loader.py helpers '
                     '(tools/make_golden_experiment.py)',
           'development_seed': int(ref_seeds.development_seed),
           'val_seeds_head': [int(s) for s in ref_seeds.val_seeds[:4]],
           'test_seeds_head': [int(s) for s in ref_seeds.test_seeds[:4]],
           'n_val_seeds': len(ref_seeds.val_seeds), 'n_test_seeds': len(ref_seeds.test_seeds),
           'seed_checksums': {'val': int(sum(ref_seeds.val_seeds)), 'test': int(sum(ref_seeds.test_seeds))},
           'splits': split_cases(), 'lcc': lcc_cases()}
    import utils.hyperparams as ref_hp  # (reference)
    out['hyperparams'] = ref_hp.hyperparams
    with open(os.path.join(GOLDEN, 'experiment_helpers.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('wrote experiment_helpers.json', os.path.getsize(os.path.join(GOLDEN, 'experiment_helpers.json')), 'bytes')


if __name__ == '__main__':
    main()
