"""Dataset loading without PyG — call surface of the reference's experiment/data_loader.py:33-180.

``DataLoader(name, use_lcc=True, undirected=False, data_dir=None)`` gives an object with ``.data`` (x, edge_index, y,
zeroed masks) and ``.num_classes``, restricted to the largest connected component with nodes relabelled in ascending
order of their original id and classes relabelled densely, exactly as data_loader.py:59-101 does.  The reference walks
components with Python sets in O(N*E) (data_loader.py:33-56); here it is one ``scipy.sparse.csgraph`` call plus array
indexing, pinned against the reference's own output in tests/golden/experiment_helpers.json.

There is no network in this image and PyG's downloaders are not available, so the raw files must already be on disk:

  * Planetoid (Cora, Citeseer, Pubmed): the eight ``ind.<name>.{x,tx,allx,y,ty,ally,graph,test.index}`` files under
    ``<data_dir>/<Name>/raw`` (or ``<data_dir>/<Name>``), read by ``read_planetoid`` below;
  * any name: ``<data_dir>/<Name>.npz`` with arrays ``x`` [N,F] float, ``edge_index`` [2,M] int, ``y`` [N] int;
  * ``synthetic:<N>:<m>:<F>:<C>``: a preferential-attachment graph with random features and labels (bench shapes).
"""
import os
import pickle

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse.csgraph import breadth_first_order, connected_components

from dcr.data import Data

DEFAULT_DATA_PATH = 'dt'
PLANETOID = ('Cora', 'Citeseer', 'Pubmed')


# ---- largest connected component -----------------------------------------------------------------------------------
def get_largest_connected_component(edge_index, num_nodes):
    """Node ids of the largest component, ascending (data_loader.py:33-56).  The reference discovers components from
    the smallest remaining node, following edges in their stored direction (row -> col), and keeps the first of the
    largest ones.  For a symmetric edge list (every real input: the datasets store both directions) that is the weak
    component holding the smallest id among the largest, found with one csgraph call; an asymmetric list falls back to
    the reference's own walk, one forward search per discovered component."""
    ei = np.asarray(edge_index)
    A = sp.coo_matrix((np.ones(ei.shape[1], dtype=np.int8), (ei[0], ei[1])), shape=(num_nodes, num_nodes)).tocsr()
    A.data[:] = 1
    if (A != A.T).nnz == 0:
        n_comp, label = connected_components(A, directed=False)
        sizes = np.bincount(label, minlength=n_comp)
        first_node = np.full(n_comp, num_nodes, dtype=np.int64)
        np.minimum.at(first_node, label, np.arange(num_nodes))
        order = np.argsort(first_node, kind='stable')      # discovery order of the reference's loop
        best = order[np.argmax(sizes[order])]              # first maximum in that order
        return np.nonzero(label == best)[0]
    remaining = np.ones(num_nodes, dtype=bool)
    best = np.empty(0, dtype=np.int64)
    while remaining.any():
        start = int(np.argmax(remaining))
        comp = breadth_first_order(A, start, directed=True, return_predecessors=False)
        if comp.shape[0] > best.shape[0]:
            best = comp
        remaining[comp] = False
    return np.sort(best).astype(np.int64)


def restrict_to_nodes(x, y, edge_index, nodes):
    """Keep the given nodes (ascending), relabel them 0..k-1, keep the edges among them in their stored order
    (data_loader.py:16-30, 78-84)."""
    ei = np.asarray(edge_index)
    new_id = np.full(int(x.shape[0]), -1, dtype=np.int64)
    new_id[nodes] = np.arange(nodes.shape[0])
    keep = (new_id[ei[0]] >= 0) & (new_id[ei[1]] >= 0)
    return x[nodes], y[nodes], np.stack([new_id[ei[0][keep]], new_id[ei[1][keep]]])


def densify_labels(y):
    """Relabel the classes 0..C-1 in ascending order of the original label (data_loader.py:96-99)."""
    classes, dense = np.unique(np.asarray(y), return_inverse=True)
    return dense.astype(np.int64), int(classes.shape[0])


# ---- raw file readers ----------------------------------------------------------------------------------------------
def _to_undirected_coalesced(src, dst, num_nodes):
    """Both directions of every pair, duplicates and self-loops dropped, sorted by (row, col): what PyG's Planetoid
    reader hands to the reference (to_undirected + coalesce)."""
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * num_nodes + dst, dst * num_nodes + src]))
    return np.stack([key // num_nodes, key % num_nodes]).astype(np.int64)


def read_planetoid(folder, name):
    """The Planetoid raw format (Kipf & Welling's ``ind.*`` pickles): labelled + unlabelled training rows (allx/ally),
    test rows (tx/ty) stored in the order of ``test.index``, adjacency as a dict of neighbour lists.  Citeseer has
    test ids without a row (isolated nodes): their rows are zero and their label is class 0, as in PyG."""
    prefix = os.path.join(folder, f'ind.{name.lower()}.')

    def load(part):
        with open(prefix + part, 'rb') as f:
            return pickle.load(f, encoding='latin1')

    allx, ally, tx, ty, graph = load('allx'), load('ally'), load('tx'), load('ty'), load('graph')
    with open(prefix + 'test.index') as f:
        test_index = np.array([int(line) for line in f.read().split()], dtype=np.int64)
    sorted_test = np.sort(test_index)
    allx = sp.csr_matrix(allx).toarray().astype(np.float32)
    tx = sp.csr_matrix(tx).toarray().astype(np.float32)
    ally, ty = np.asarray(ally), np.asarray(ty)
    if name.lower() == 'citeseer':  # pad the gaps in the test id range with empty rows
        span = int(sorted_test[-1] - sorted_test[0] + 1)
        tx_full = np.zeros((span, tx.shape[1]), dtype=np.float32)
        ty_full = np.zeros((span, ty.shape[1]), dtype=ty.dtype)
        tx_full[sorted_test - sorted_test[0]] = tx
        ty_full[sorted_test - sorted_test[0]] = ty
        tx, ty = tx_full, ty_full
    x = np.concatenate([allx, tx], 0)
    y = np.concatenate([ally, ty], 0).argmax(1).astype(np.int64)
    x[test_index] = x[sorted_test]
    y[test_index] = y[sorted_test]
    n = x.shape[0]
    src = np.fromiter((u for u, nb in graph.items() for _ in nb), dtype=np.int64)
    dst = np.fromiter((v for nb in graph.values() for v in nb), dtype=np.int64)
    return x, y, _to_undirected_coalesced(src, dst, n)


def read_npz(path):
    z = np.load(path)
    ei = np.asarray(z['edge_index'], dtype=np.int64)
    return np.asarray(z['x'], dtype=np.float32), np.asarray(z['y'], dtype=np.int64), ei


def synthetic_dataset(n, m, n_feat, n_classes, seed=0):
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(n, m, seed=12345)
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((n, n_feat), dtype=np.float32), rng.integers(0, n_classes, n), ei


def load_raw(name, data_dir):
    if name.startswith('synthetic:'):
        n, m, f, c = (int(t) for t in name.split(':')[1:5])
        return synthetic_dataset(n, m, f, c)
    npz = os.path.join(data_dir, f'{name}.npz')
    if os.path.exists(npz):
        return read_npz(npz)
    if name in PLANETOID:
        for folder in (os.path.join(data_dir, name, 'raw'), os.path.join(data_dir, name), data_dir):
            if os.path.exists(os.path.join(folder, f'ind.{name.lower()}.graph')):
                return read_planetoid(folder, name)
    raise FileNotFoundError(
        f'no raw files for dataset {name!r} under {data_dir!r}: expected {name}.npz or the Planetoid ind.* files '
        f'(this image has no network and no PyG downloaders)')


# ---- the reference's entry points ----------------------------------------------------------------------------------
def get_dataset(name, use_lcc=True, data_dir=DEFAULT_DATA_PATH):
    """x / y / edge_index of the named dataset, LCC-restricted and densely labelled (data_loader.py:59-101)."""
    x, y, ei = load_raw(name, data_dir)
    if use_lcc:
        x, y, ei = restrict_to_nodes(x, y, ei, get_largest_connected_component(ei, x.shape[0]))
    y, num_classes = densify_labels(y)
    n = x.shape[0]
    zeros = lambda: torch.zeros(n, dtype=torch.bool)  # noqa: E731
    data = Data(x=torch.from_numpy(np.ascontiguousarray(x)), edge_index=torch.from_numpy(np.ascontiguousarray(ei)),
                y=torch.from_numpy(y), num_nodes=n, train_mask=zeros(), val_mask=zeros(), test_mask=zeros())
    return data, num_classes


class DataLoader:
    """``DataLoader(name, use_lcc, undirected, data_dir)`` (data_loader.py:104-180): ``.data``, ``.num_classes``,
    ``.num_features``; ``str()`` names the processed variant the way the reference's cache file is named."""

    def __init__(self, name='Cora', use_lcc=True, undirected=False, data_dir=None):
        self.name = name
        self.use_lcc = use_lcc
        self.undirected = undirected
        self.data_dir = DEFAULT_DATA_PATH if data_dir is None else data_dir
        data, self.num_classes = get_dataset(name, use_lcc, self.data_dir)
        if not undirected:
            # data_loader.py:155-166 rebuilds the edge list from a dense symmetrised adjacency; the same edge set
            # (row-major order, unit weights) without the N x N matrix
            ei = data.edge_index.numpy()
            data.edge_index = torch.from_numpy(_to_undirected_coalesced(ei[0], ei[1], data.num_nodes))
            data.edge_attr = torch.ones(data.edge_index.shape[1], dtype=torch.float32)
        self.data = data

    @property
    def num_features(self):
        return int(self.data.x.shape[1])

    def __str__(self):
        return f"{self.name}_{'base' if not self.undirected else 'undirected'}_lcc={self.use_lcc}"
