"""Does a captured epoch survive eager training steps of the same model before it?  (round 5: a test that ran an eager backward
on the default stream and then captured an epoch of the same model died in capture_end.)"""
import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for mode in ('none', 'train', 'train_zero', 'manual', 'manual_sync'):
        for hid in ('1', '3'):
            r = subprocess.run([sys.executable, __file__, mode, hid], capture_output=True, text=True)
            print(f'eager before capture: {mode:12s} hidden layers {hid}: rc={r.returncode} {r.stdout.strip()[-80:]} {r.stderr.strip()[-300:] if r.returncode else ""}', flush=True)
    sys.exit(0)
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data, Dataset
from models.gcn import GCN
from experiment.training_loop import make_epoch, train, evaluate
mode, depth = sys.argv[1], int(sys.argv[2])
dev = torch.device('cuda', 0)
ei_np, n = synthetic.powerlaw_graph(3000, 4, seed=5)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, 96, device=dev, generator=g)
y = torch.randint(0, 5, (n,), device=dev, generator=g)
r = torch.rand(n, device=dev, generator=g)
data = Data(x=x, edge_index=torch.from_numpy(ei_np).to(dev), y=y, num_nodes=n, train_mask=r < 0.3, val_mask=(r >= 0.3) & (r < 0.6))
model = GCN(Dataset(data, 5), hidden=[64] * depth, dropout=0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
if mode in ('train', 'train_zero'):
    for _ in range(2):
        train(model, opt, data)
        evaluate(model, data, test=False)
    if mode == 'train_zero':
        opt.zero_grad(set_to_none=True)
elif mode in ('manual', 'manual_sync'):
    model.train()
    lp = model(data)
    torch.nn.functional.nll_loss(lp[data.train_mask], data.y[data.train_mask]).backward()
    if mode == 'manual_sync':
        torch.cuda.synchronize()
        model.zero_grad(set_to_none=True)
epoch = make_epoch(model, opt, data, lagged=True)
accs = [epoch() for _ in range(8)]
torch.cuda.synchronize()
print(type(epoch).__name__, 'ok', round(accs[-1], 3))
