#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-gcn --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d['bfc_pass_ms'], d['outside_pass_ms'])
        print(json.dumps(d.get('sdrf_cora_shape')))
        print(json.dumps(d.get('incremental_mode')))
"
