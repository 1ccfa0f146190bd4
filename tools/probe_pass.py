"""Timing probe of the curvature pass alone on the north-star graph (not part of the bench contract)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
from dcr.graph import DcrGraph
ei, n = synthetic.powerlaw_graph(int(os.environ.get('N', 100000)), int(os.environ.get('M', 10)), seed=12345)
G = DcrGraph(ei, n)
G.curvature_pass('bfc')
G.curvature_pass('bfc')
G.profile_reset()
reps = int(os.environ.get('REPS', 5))
for _ in range(reps):
    G.curvature_pass('bfc')
ms, cnt = G.profile_read()
print('pass ms', ms / cnt, 'engine', G.pass_engine(), flush=True)
