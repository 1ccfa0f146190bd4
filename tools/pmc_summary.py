"""Per-kernel averages of a rocprofv3 --pmc run (csv output): tools/pmc_summary.py <counter_collection.csv> [filter]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row['Kernel_Name'].split('(')[0][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    n[k].add(row['Dispatch_Id'])
for k in acc:
    print(k, 'dispatches', len(n[k]))
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {v / len(n[k]):16.1f}')
