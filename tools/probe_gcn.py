"""Timing probe of the GCN epoch and the weight-gradient kernel (not part of the bench contract)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from models.gcn import atb_hip
dev = torch.device('cuda', 0)
for K, M, N in ((1000000, 128, 256), (1000000, 16, 128), (100000, 64, 3703)):
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev)
    for fn, name in ((atb_hip, 'mfma kernel'), (lambda a, b: a.t() @ b, 'GEMM library')):
        for _ in range(3): fn(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn(a, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'AtB K={K} M={M} N={N}  {name:13s} {ms:7.3f} ms  {2.0 * K * M * N / ms / 1e9:7.1f} TFLOP/s  {(K * (M + N) * 4) / ms / 1e6:7.0f} GB/s', flush=True)
