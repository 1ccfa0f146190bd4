import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/discrete-curvature-rewiring_amd']
import torch
from models import gcn
from models.gcn import _ActLinearFn, _ReluDropoutFn
dev = torch.device('cuda', 0)
n, H, C = 1000000, 128, 16
x = torch.randn(n, H, device=dev); w = torch.randn(C, H, device=dev) * 0.1
def tm(f, name):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 10 * 1e3, 'ms', flush=True)
tm(lambda: _ActLinearFn.apply(x, w, 0.5, True, True), 'pair')
tm(lambda: _ActLinearFn.apply(x, w, 0.5, True, False), 'train only')
tm(lambda: _ActLinearFn.apply(x, w, 0.0, False, True), 'eval only')
tm(lambda: _ReluDropoutFn.apply(x, 0.5), 'relu_dropout alone')
tm(lambda: torch.nn.functional.linear(x, w), 'library linear')
