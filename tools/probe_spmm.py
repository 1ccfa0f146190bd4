"""Times the aggregation kernel on the S1M graph: all rows at widths 16 / 128, and the row-selected calls of an epoch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from models.gcn import gcn_norm_csr, _spmm_hip, spmm_rows, RowSelection, spmm
dev = torch.device('cuda', 0)
ei_np, n = synthetic.powerlaw_graph(1000000, 10, seed=12345)
csr = gcn_norm_csr(torch.from_numpy(ei_np).to(dev), None, n)
g = torch.Generator(device=dev).manual_seed(0)
r = torch.rand(n, device=dev, generator=g)
sel = RowSelection(csr, r < 0.1)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for w in (16, 128):
    z = torch.randn(n, w, device=dev)
    print(f'all rows, width {w}: {t(lambda: _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)):.1f} us', flush=True)
z2 = torch.randn(n, 32, device=dev)
print(f'100k selected rows, width 16 (half of a 32-wide buffer): {t(lambda: spmm_rows(csr, sel, z2[:, :16])):.1f} us')
rp, ci, va = sel.transposed()
gc = torch.randn(sel.n, 16, device=dev)
print(f'backward through the column-restricted transpose: {t(lambda: spmm(rp, ci, va, gc, csr.n_cols)):.1f} us')
