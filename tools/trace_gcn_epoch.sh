cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg && EPOCHS=6 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_gcn_epoch.py > /dev/null 2>&1; python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/pg/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last epoch: from the last k_first_layer_fwd back to the one before
idx = [i for i, r in enumerate(rows) if 'k_first_layer_fwd' in r['Kernel_Name'] or 'k_first_layer_wide' in r['Kernel_Name']]
if len(idx) < 2:   # the sparse-input route: an epoch starts with the snapshot's multi-tensor copy
    idx = [i for i, r in enumerate(rows) if 'k_adam_multi' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {r['Kernel_Name'][:110]}")
PY
