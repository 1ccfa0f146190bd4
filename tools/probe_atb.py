"""Times dcr_atb_f32_dev at the two weight-gradient shapes of the S1M GCN epoch (not part of the bench contract)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from models.gcn import atb_hip
dev = torch.device('cuda', 0)
K = 1000000
for M, N in ((128, 256), (16, 128)):
    a = torch.randn(K, M, device=dev)
    b = torch.randn(K, N, device=dev)
    ref = (a[:20000].double().t() @ b[:20000].double())
    got = atb_hip(a[:20000].contiguous(), b[:20000].contiguous()).double()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3):
        atb_hip(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        atb_hip(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'atb {M}x{N}xK={K}: {ms * 1e3:.1f} us  {2 * K * M * N / ms / 1e9:.1f} TFLOP/s  rel err {err:.2e}', flush=True)
