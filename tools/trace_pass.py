"""Timeline of one curvature pass from a rocprofv3 --kernel-trace csv: tools/trace_pass.py <kernel_trace.csv> [pass index from the end]
(start / end of every kernel relative to the first kernel of the pass, microseconds)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a pass starts at k_h2_clear / k_nc_clear
starts = [i for i, n in enumerate(names) if 'k_h2_clear' in n or 'k_nc_clear' in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0 = starts[-which]
i1 = starts[-which + 1] if which > 1 else len(rows)
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  {r['Kernel_Name'].split('(')[0][:70]}")
