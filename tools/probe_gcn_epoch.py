"""The GCN epoch of bench.py's GCN leg alone (S1M shape, one GPU, the one-graph lagged epoch), for per-kernel profiling
(tools/prof_gcn.sh).  Not part of the bench contract.  EPOCHS, EAGER=1 (no HIP graph: kernels keep their names under rocprofv3)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic
from dcr.data import Data, Dataset
from models.gcn import GCN
from experiment.training_loop import make_epoch
dev = torch.device('cuda', 0)
n, F, H, C = int(os.environ.get('N', 1000000)), int(os.environ.get('F', 256)), int(os.environ.get('H', 128)), int(os.environ.get('C', 16))
ei_np, n = synthetic.powerlaw_graph(n, int(os.environ.get('M', 10)), seed=12345)
ei = torch.from_numpy(ei_np).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, F, device=dev, generator=g)
if os.environ.get('SPARSEX') == '1':   # a row-normalised bag of words of Planetoid's density
    x = (torch.rand(n, F, device=dev, generator=g) < 0.009).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
y = torch.randint(0, C, (n,), device=dev, generator=g)
r = torch.rand(n, device=dev, generator=g)
data = Data(x=x, edge_index=ei, y=y, num_nodes=n, train_mask=r < 0.1, val_mask=(r >= 0.1) & (r < 0.2))
torch.manual_seed(0)
model = GCN(Dataset(data, C), hidden=[H], dropout=0.5).to(dev)
from experiment.save_models import make_adam
mode = os.environ.get('DCR_FUSED_ADAM', '2')     # 2: experiment/adam.py (one launch), 1: torch fused, 0: torch stock
os.environ['DCR_FUSED_ADAM'] = mode
opt = make_adam([{'params': model.non_reg_params, 'weight_decay': 0},
                 {'params': model.reg_params, 'weight_decay': 5e-4}], 0.01, dev, fused=None if mode == '2' else mode == '1')
epoch = make_epoch(model, opt, data, lagged=True)
for _ in range(6):
    epoch()
torch.cuda.synchronize()
E = int(os.environ.get('EPOCHS', 20))
t = time.perf_counter()
for _ in range(E):
    epoch()
torch.cuda.synchronize()
print(f'epoch ms {(time.perf_counter() - t) / E * 1e3:.3f} driver {type(epoch).__name__} adam {type(opt).__name__} head {os.environ.get("DCR_FUSED_HEAD", "1")}', flush=True)
from models import gcn as _g
for key, ahead in getattr(_g, '_AHEAD', {}).items():   # decisions drawn ahead: the stamp's offset is the counter's value when the draw hit
    print('dropout ahead', key, 'stamp', ahead.words[-8:-4].tolist(), 'next', int(ahead.words[-4].item()), 'counter', int(_g._dropout_counter(dev).item()), flush=True)
