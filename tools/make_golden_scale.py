#!/usr/bin/env python3
"""End-to-end SDRF fixtures at BASELINE.json's two synthetic sizes (configs[2]: 100k nodes / 1M edges; configs[4]: 1M nodes /
10M edges), recorded ONCE in the build container with the pinned C oracle (oracle/dcr_oracle.c: bit-identical to every
fixture the reference itself produced, tests/test_oracle_golden.py — the Python reference needs 2 s per bfc_edge call at this
size, 24 days per pass, so it cannot record these itself).

Per iteration of rewiring/sdrf_no_cuda.py:22-66 (tau = 163, removal_bound = 0.95: the Cora hyper-parameters SURVEY §8(d)
fixes for the synthetic runs; np.random.seed(0)): arg-min edge and its curvature (float64 hex), number of candidates,
SHA-256 of the candidate list (int32 pairs in the reference's nested-loop order, duplicates kept) and of the improvement
vector (float64 bytes), drawn index, added and removed edge; then the SHA-256 of the final edge_index (int64 [2, 2E], the
reference's from_networkx order) and curvatures of the initial and of the rewired graph on sampled + heaviest edges.

    python tools/make_golden_scale.py s100k [iterations]     -> tests/golden/sdrf_s100k_oracle.json   (configs[2] as written: 500)
    python tools/make_golden_scale.py s1m   [iterations]     -> tests/golden/sdrf_s1m_oracle.json
    python tools/make_golden_scale.py s100k_removal          -> tests/golden/sdrf_s100k_removal_oracle.json
    python tools/make_golden_scale.py s1m_removal            -> tests/golden/sdrf_s1m_removal_oracle.json

The two `_removal` cases exercise sdrf_no_cuda.py:57-63 (stale arg-max, exclusion of the added edge, conditional removal) at
full size.  Every Balanced Forman curvature of these preferential-attachment graphs is below -1.07 (minimum degree 10:
2/d1 + 2/d2 - 2 <= -1.6 before the triangle terms), so none of the reference's removal bounds (utils/hyperparams.py: 0 ... 14.43)
ever fires on them; the cases keep Citeseer's tau = 180 (hyperparams.py:19) and put the bound at -1.19, which 20 edges of the
initial S100k graph exceed: the run removes for its first iterations and stops removing once the stale maximum falls under the
bound, so both outcomes of `:62` are in the trace (S1M, whose largest curvature is -1.238: bound -1.26, 17 edges above it, one
iteration).  The generator asserts the number of removals.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'discrete-curvature-rewiring_amd'))

from oracle import c_oracle  # noqa: E402
from dcr.synthetic import powerlaw_graph  # noqa: E402

CASES = {
    's100k': dict(n=100_000, m=10, seed=12345, iterations=500, sampled=5000, heaviest=300),
    's1m': dict(n=1_000_000, m=10, seed=12345, iterations=3, sampled=5000, heaviest=300),
    's100k_removal': dict(n=100_000, m=10, seed=12345, iterations=40, sampled=5000, heaviest=300, tau=180.0, bound=-1.19,
                          min_removals=5, min_kept=5),
    's1m_removal': dict(n=1_000_000, m=10, seed=12345, iterations=1, sampled=2000, heaviest=300, tau=180.0, bound=-1.26,
                        min_removals=1, min_kept=0),
}
TAU, BOUND, NP_SEED = 163.0, 0.95, 0


def sample_edges(edge_index, n, sampled, heaviest, seed):
    """`sampled` edges drawn without replacement (PCG64(seed)) + the `heaviest` edges by deg(u) + deg(v) (ties: edge
    order), as (u, v) with u < v."""
    ei = np.asarray(edge_index)
    und = ei[:, ei[0] < ei[1]]
    deg = np.bincount(ei[0], minlength=n)
    rng = np.random.Generator(np.random.PCG64(seed))
    pick = np.sort(rng.choice(und.shape[1], size=min(sampled, und.shape[1]), replace=False))
    weight = deg[und[0]] + deg[und[1]]
    heavy = np.argsort(-weight, kind='stable')[:heaviest]
    idx = np.concatenate([pick, heavy])
    return und[0, idx].astype(np.int32), und[1, idx].astype(np.int32)


def curvature_block(edge_index, n, sampled, heaviest, seed, nthreads):
    eu, ev = sample_edges(edge_index, n, sampled, heaviest, seed)
    G = c_oracle.CGraph(edge_index, n)
    vals = G.curv_edges(eu, ev, 'bfc', nthreads)
    return {'u': eu.tolist(), 'v': ev.tolist(), 'bfc_hex': [float(x).hex() for x in vals], 'sample_seed': seed,
            'sampled': int(min(sampled, (np.asarray(edge_index)[0] < np.asarray(edge_index)[1]).sum())), 'heaviest': heaviest}


def main():
    name = sys.argv[1]
    case = dict(CASES[name])
    if len(sys.argv) > 2:
        case['iterations'] = int(sys.argv[2])
    tau, bound = case.get('tau', TAU), case.get('bound', BOUND)
    nthreads = int(os.environ.get('ORACLE_THREADS', str(max(1, (os.cpu_count() or 2) - 1))))
    t0 = time.time()
    ei, n = powerlaw_graph(case['n'], case['m'], seed=case['seed'])
    print(f'[{name}] graph: {n} nodes, {ei.shape[1] // 2} edges ({time.time() - t0:.0f} s)', flush=True)
    out = {'generator': 'tools/make_golden_scale.py (C oracle, oracle/dcr_oracle.c)', 'graph': {'n': n, 'm': case['m'], 'seed': case['seed'],
           'edges': int(ei.shape[1] // 2), 'edge_index_sha256': hashlib.sha256(np.ascontiguousarray(ei).tobytes()).hexdigest()},
           'tau': tau, 'removal_bound': bound, 'numpy_seed': NP_SEED, 'remove_edges': True}
    out['initial_curvature'] = curvature_block(ei, n, case['sampled'], case['heaviest'], 1, nthreads)
    print(f'[{name}] initial curvatures sampled ({time.time() - t0:.0f} s)', flush=True)
    trace = []
    np.random.seed(NP_SEED)

    def progress(i, rec):
        print(f'[{name}] iteration {i}: argmin {rec["argmin"]} candidates {rec.get("n_candidates")} added {rec.get("added")} '
              f'removed {rec.get("removed")} ({time.time() - t0:.0f} s)', flush=True)

    final = c_oracle.sdrf(ei, n, 'bfc', case['iterations'], True, bound, tau, trace=trace, nthreads=nthreads, compact=True,
                          progress=progress)
    out['iterations'] = trace
    removals = sum(1 for r in trace if r.get('removed') is not None)
    out['removals'] = removals
    if 'min_removals' in case:    # the removal branch (sdrf_no_cuda.py:57-63) must be in the fixture, both ways
        assert removals >= case['min_removals'], (removals, 'removals: the bound does not fire often enough')
        assert len(trace) - removals >= case['min_kept'], (removals, len(trace), 'every iteration removed: raise the bound')
    out['numpy_next_uniform_hex'] = float(np.random.random_sample()).hex()   # the stream's position after the run
    final = np.ascontiguousarray(final, dtype=np.int64)
    out['final'] = {'edges': int(final.shape[1] // 2), 'edge_index_sha256': hashlib.sha256(final.tobytes()).hexdigest()}
    out['rewired_curvature'] = curvature_block(final, n, case['sampled'], case['heaviest'], 2, nthreads)
    out['recorded'] = {'seconds': round(time.time() - t0), 'threads': nthreads}
    path = os.path.join(ROOT, 'tests', 'golden', f'sdrf_{name}_oracle.json')
    with open(path, 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print(f'[{name}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KB, {time.time() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    main()
