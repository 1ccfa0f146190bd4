import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
from dcr.graph import DcrGraph
from oracle import c_oracle
ei, nn = synthetic.erdos_renyi_graph(300, 0.9, seed=5)
G = DcrGraph(ei, nn); C = c_oracle.CGraph(ei, nn)
eu, ev = G.edges()
bad = 0
for e in range(0, 1500):
    a = G.bfc_ingredients(int(eu[e]), int(ev[e])).tolist(); b = C.ingredients(int(eu[e]), int(ev[e])).tolist()
    if a != b:
        bad += 1
        if bad < 5: print(e, int(eu[e]), int(ev[e]), a, b, flush=True)
    if e % 250 == 0: print('at', e, 'bad', bad, flush=True)
print('done bad', bad)
