"""Timing of the one-kernel first layer (dcr_first_layer_fwd_f32_dev) against the route it replaces (GEMM library +
dcr_act_linear_fwd_f32_dev) at the bench shape (1M x 256 -> 128 -> 16).  DCR_LIB picks a variant build."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from models import gcn
from models.gcn import _ActLinearFn, _FirstLayerFn
dev = torch.device('cuda', 0)
n, F, H, C = int(os.environ.get('N', 1000000)), int(os.environ.get('F', 256)), int(os.environ.get('H', 128)), int(os.environ.get('C', 16))
g = torch.Generator(device=dev).manual_seed(0)
F16 = (F + 15) // 16 * 16
ax = torch.zeros(n, F16, device=dev)
ax[:, :F] = torch.randn(n, F, device=dev, generator=g)
w1 = torch.randn(H, F, device=dev, generator=g) * F ** -0.5
b1 = torch.randn(H, device=dev, generator=g) * 0.1
w2 = torch.randn(C, H, device=dev, generator=g) * 0.1


def tm(f, name):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    reps = int(os.environ.get('REPS', 10))
    for _ in range(reps):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    print(f'{name}: {ev[0].elapsed_time(ev[1]) / reps * 1e3:.1f} us', flush=True)


with torch.no_grad():
    tm(lambda: _FirstLayerFn.apply(ax, w1, b1, w2, 0.5, True, True), 'one kernel, pair')
    tm(lambda: _FirstLayerFn.apply(ax, w1, b1, w2, 0.5, True, False), 'one kernel, train only')
    tm(lambda: _FirstLayerFn.apply(ax, w1, b1, w2, 0.0, False, True), 'one kernel, eval only (no pre written)')
    axc = ax[:, :F].contiguous()
    pre = torch.nn.functional.linear(axc, w1, b1)
    tm(lambda: torch.nn.functional.linear(axc, w1, b1), 'library linear')
    tm(lambda: _ActLinearFn.apply(pre, w2, 0.5, True, True), 'act_linear pair')
