"""Host-side cost of library calls (ctypes + wrapper), to separate it from GPU time.  Not part of the bench contract."""
import os, sys, time, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import synthetic, _lib
from dcr.graph import DcrGraph
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
G = DcrGraph(ei, n)
x, y, _ = G.curvature_pass_argmin('bfc')
def t(f, reps=200):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
print('number_of_edges      %8.1f us' % t(G.number_of_edges))
print('improvements(view)   %8.1f us' % t(lambda: G.improvements(x, y, 'bfc')))
print('improvements_count   %8.1f us' % t(lambda: G.improvements_count(x, y, 'bfc')))
nn = ctypes.c_int64(); pi = _lib._f64p()
L = _lib.lib()
print('raw ctypes call      %8.1f us' % t(lambda: L.dcr_improvements(G._h, x, y, 0, 0, ctypes.byref(nn), ctypes.byref(pi), None, None)))
imp, _, _ = G.improvements(x, y, 'bfc')
print('n =', imp.shape[0])
print('copy of the view     %8.1f us' % t(lambda: np.array(imp)))
print('view * tau           %8.1f us' % t(lambda: imp * 163.0))
c = np.array(imp)
print('copy * tau           %8.1f us' % t(lambda: c * 163.0))
print('exp(copy*tau)        %8.1f us' % t(lambda: np.exp(c * 163.0)))
e = np.exp(c * 163.0)
print('sum                  %8.1f us' % t(lambda: e.sum()))
from rewiring.sdrf_no_cuda import draw_index
np.random.seed(0)
print('draw_index           %8.1f us' % t(lambda: draw_index(imp, 163.0)))
