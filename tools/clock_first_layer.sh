#!/bin/bash
# Shader clock under each kernel of tools/probe_first_layer.py: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk -o p -- python3 $GRAFT_REPO_ROOT/tools/${PROBE:-probe_first_layer.py} > /dev/null 2>&1
ls /tmp/clk
python3 - <<'PY'
import csv, collections
dur = {}
for r in csv.DictReader(open('/tmp/clk/p_kernel_trace.csv')):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/clk/p_counter_collection.csv')):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    name, ns = dur[r['Dispatch_Id']]
    if 'k_first_layer' not in name and 'k_act_linear' not in name and 'k_atb' not in name and 'Cijk' not in name: continue
    acc[name.split('(')[0].replace('void dcr::', '')[:44]].append((ns, float(r['Counter_Value'])))
for k, v in sorted(acc.items()):
    ns = sum(a for a, _ in v) / len(v); cyc = sum(b for _, b in v) / len(v)
    print(f'{k:46s} {ns / 1e3:8.1f} us  {cyc / 8 / 1e6:7.3f} Mcycles/XCD  clock {cyc / 8 / ns:5.2f} GHz  ({len(v)} launches)')
PY
