"""Timing of the one-kernel backward of the first layer (dcr_first_layer_bwd_f32_dev) against the route it replaces
(dcr_act_linear_bwd_fused_f32_dev + dcr_atb_f32_dev) at the bench shape (1M x 256 -> 128 -> 16).  DCR_LIB picks a variant build."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from models.gcn import _FirstLayerFn
dev = torch.device('cuda', 0)
n, F, H, C = int(os.environ.get('N', 1000000)), 256, 128, 16
g = torch.Generator(device=dev).manual_seed(0)
ax = torch.randn(n, F, device=dev, generator=g)
w1 = (torch.randn(H, F, device=dev, generator=g) * F ** -0.5).requires_grad_(True)
b1 = (torch.randn(H, device=dev, generator=g) * 0.1).requires_grad_(True)
w2 = (torch.randn(C, H, device=dev, generator=g) * 0.1).requires_grad_(True)
gz = torch.randn(n, C, device=dev, generator=g)
z, _ = _FirstLayerFn.apply(ax, w1, b1, w2, 0.5, True, True)
for mode in ('1', '0', '1', '0'):
    os.environ['DCR_FIRST_BWD_FUSED'] = mode
    for _ in range(3):
        z.backward(gz, retain_graph=True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        z.backward(gz, retain_graph=True)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"backward, {'one kernel' if mode == '1' else 'separate kernels'}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms", flush=True)
