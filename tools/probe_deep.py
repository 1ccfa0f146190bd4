"""Captured epoch of a deeper GCN (Pubmed's hidden_depth 3): which configuration survives the HIP graph capture."""
import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for head in ('1', '0'):
        for drop in ('0.0', '0.5'):
            for pre in ('0', '1'):
                env = dict(os.environ, DCR_FUSED_HEAD=head, HID='64', DROP=drop, PRE=pre)
                r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
                print(f'head={head} dropout={drop} eager forward_head + forward_pair first={pre}: rc={r.returncode} {r.stdout.strip()[-100:]} {r.stderr.strip()[-200:] if r.returncode else ""}', flush=True)
    sys.exit(0)
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data, Dataset
from models.gcn import GCN
from experiment.training_loop import make_epoch
dev = torch.device('cuda', 0)
ei_np, n = synthetic.powerlaw_graph(3000, 4, seed=5)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, 96, device=dev, generator=g)
y = torch.randint(0, 5, (n,), device=dev, generator=g)
r = torch.rand(n, device=dev, generator=g)
data = Data(x=x, edge_index=torch.from_numpy(ei_np).to(dev), y=y, num_nodes=n, train_mask=r < 0.3, val_mask=(r >= 0.3) & (r < 0.6))
H = int(os.environ['HID'])
model = GCN(Dataset(data, 5), hidden=[H, H, H], dropout=float(os.environ.get('DROP', '0.5'))).to(dev)
if os.environ.get('PRE') == '1':
    tr_idx, ev_idx = data.train_mask.nonzero().squeeze(1), data.val_mask.nonzero().squeeze(1)
    y_tr, y_ev = data.y[tr_idx].contiguous(), data.y[ev_idx].contiguous()
    model.train()
    out = model.forward_head(data, rows_train=tr_idx, y_train=y_tr, rows_eval=ev_idx, y_eval=y_ev)
    if out is not None:
        out[0].backward()
    model.zero_grad()
    lp_tr, lp_ev = model.forward_pair(data, rows_train=tr_idx, rows_eval=ev_idx)
    torch.nn.functional.nll_loss(lp_tr, y_tr).backward()
opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
epoch = make_epoch(model, opt, data, lagged=True)
accs = [epoch() for _ in range(8)]
torch.cuda.synchronize()
print(type(epoch).__name__, 'ok', accs[-1])
