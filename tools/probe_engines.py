"""Pass time of the node-centric and the two-hop engine over graph sizes (where automatic selection should switch)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n, m in [(2485, 2), (2500, 10), (5000, 10), (10000, 10), (30000, 10), (100000, 10), (300000, 10)]:
    out = []
    for eng in ('nc', 'h2'):
        env = dict(os.environ, DCR_PASS=eng, N=str(n), M=str(m), REPS='20')
        r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'probe_pass.py')], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if 'pass ms' in l]
        out.append(line[0].split()[2] if line else 'failed')
    print(f'n {n} m {m}: node-centric {out[0]} ms, two-hop {out[1]} ms', flush=True)
