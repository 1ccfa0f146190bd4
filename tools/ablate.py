import os, sys, time, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(_lib.CSRC, sys.argv[1])
from dcr import synthetic
from dcr.graph import DcrGraph
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
G = DcrGraph(ei, n)
G.curvature_pass('bfc'); G.profile_reset()
for _ in range(10): G.curvature_pass('bfc')
ms, cnt = G.profile_read()
print(sys.argv[1:] or 'full', 'pass ms', round(ms / cnt, 3), flush=True)
