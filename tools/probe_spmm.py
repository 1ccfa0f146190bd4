"""Timing probe of the SpMM kernel on the GCN bench graph (not part of the bench contract)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from models.gcn import gcn_norm_csr, _spmm_hip, _spmm_torch
n = int(os.environ.get('N', 1000000))
ei_np, n = synthetic.powerlaw_graph(n, 10, seed=12345)
dev = torch.device('cuda', 0)
ei = torch.from_numpy(ei_np).to(dev)
csr = gcn_norm_csr(ei, None, n)
nnz = int(csr.col.shape[0])
for F in (16, 64, 128):
    z = torch.randn(n, F, device=dev)
    for _ in range(3):
        out = _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if n <= 200000:
        ref = _spmm_torch(csr.rowptr, csr.col, csr.val, z, csr.n_rows)
        err = (out - ref).abs().max().item()
    else:
        err = float('nan')
    print(f'F={F:4d}  {ms:8.3f} ms   gathered {nnz * (F * 4 + 8) / ms / 1e6:8.1f} GB/s   max|err| vs torch {err:.2e}', flush=True)
