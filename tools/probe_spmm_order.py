"""Does a node order help the aggregation's gathers on the S1M bench graph?  (round-4 verdict, item 7: "renumber nodes at CSR build
— hubs first / degree-bucketed, optionally BFS inside buckets".)  Times dcr_spmm_csr_f32_dev at width 128 and 16 on the same
graph under four labellings: the generator's own (preferential attachment: a node's expected degree falls with its id, so the
ids ARE roughly hubs-first), exactly degree-sorted, BFS from the largest hub, and a random permutation."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import numpy as np
import torch
from dcr import synthetic
from models.gcn import gcn_norm_csr, _spmm_hip
dev = torch.device('cuda', 0)
ei_np, n = synthetic.powerlaw_graph(int(os.environ.get('N', 1000000)), 10, seed=12345)
deg = np.bincount(ei_np[0], minlength=n)


def bfs_order():
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order
    A = sp.csr_matrix((np.ones(ei_np.shape[1], dtype=np.int8), (ei_np[0], ei_np[1])), shape=(n, n))
    order, _ = breadth_first_order(A, int(np.argmax(deg)), directed=False)
    rest = np.setdiff1d(np.arange(n), order)
    return np.concatenate([order, rest])


def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


orders = {'generator ids': np.arange(n), 'degree-sorted (hubs first)': np.argsort(-deg, kind='stable'), 'BFS from the largest hub': bfs_order(),
          'random permutation': np.random.Generator(np.random.PCG64(1)).permutation(n)}
for name, order in orders.items():
    new_id = np.empty(n, dtype=np.int64)
    new_id[order] = np.arange(n)
    ei = torch.from_numpy(new_id[ei_np]).to(dev)
    csr = gcn_norm_csr(ei, None, n)
    out = []
    for w in (128, 16):
        z = torch.randn(n, w, device=dev)
        out.append(f'width {w}: {t(lambda: _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)):8.1f} us')
    print(f'{name:30s} ' + '   '.join(out), flush=True)
